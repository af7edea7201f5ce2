"""ctypes binding of librtm3d_hip.so (C ABI in include/rtm3d_hip.h).

The product path has no CPU fallback: if the library is missing or cannot be loaded, every
entry point raises ``RuntimeError`` telling the user to run ``python -c "import __graft_entry__ as g; g.build()"``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_C', 'librtm3d_hip.so')
ABI_VERSION = 9
MAX_GROUPS, MAX_TAPS = 4, 80

c_int, c_void_p, c_float, c_size_t = ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_size_t
c_double = ctypes.c_double


class ConvDesc(ctypes.Structure):
    """Mirror of struct rtm3d_conv_desc."""
    _fields_ = [
        ('in_tensor', c_int), ('out_tensor', c_int), ('res_tensor', c_int),
        ('Hm', c_int), ('Wm', c_int),
        ('in_stride', c_int), ('out_scale', c_int),
        ('cin', c_int), ('cout', c_int),
        ('groups', c_int), ('ntaps', c_int),
        ('in_coff', c_int * MAX_GROUPS), ('out_coff', c_int * MAX_GROUPS), ('res_coff', c_int * MAX_GROUPS),
        ('out_oy', c_int * MAX_GROUPS), ('out_ox', c_int * MAX_GROUPS),
        ('tap_dy', (c_int * MAX_TAPS) * MAX_GROUPS), ('tap_dx', (c_int * MAX_TAPS) * MAX_GROUPS),
        ('tap_dc', (c_int * MAX_TAPS) * MAX_GROUPS),
        ('s2d_tensor', c_int), ('s2d_coff', c_int), ('in_s2d', c_int),
        ('relu', c_int),
        ('w_blob', c_int), ('bias_blob', c_int),
        ('kernel', c_int), ('bn_tile', c_int),
        ('out_nchw_f32', c_int), ('out_H', c_int), ('out_W', c_int),
        ('softmax_stat_slot', c_int),
    ]


class VTensor(ctypes.Structure):
    """Mirror of struct rtm3d_vtensor (fp32 verification executor)."""
    _fields_ = [('d', c_void_p), ('Hp', c_int), ('Wp', c_int), ('C', c_int), ('P', c_int), ('coff', c_int)]


class VConvDesc(ctypes.Structure):
    """Mirror of struct rtm3d_vconv_desc."""
    _fields_ = [
        ('inp', VTensor), ('out', VTensor), ('res', VTensor),
        ('d_w', c_void_p), ('d_bias', c_void_p),
        ('B', c_int), ('Hm', c_int), ('Wm', c_int), ('in_stride', c_int), ('out_scale', c_int), ('out_oy', c_int), ('out_ox', c_int),
        ('cin', c_int), ('cout', c_int), ('ntaps', c_int), ('relu', c_int),
        ('out_nchw_f32', c_int), ('out_H', c_int), ('out_W', c_int),
        ('tap_dy', c_int * MAX_TAPS), ('tap_dx', c_int * MAX_TAPS),
    ]


# name -> (restype, argtypes); also the list of symbols include/rtm3d_hip.h declares
SIGNATURES = {
    'rtm3d_last_error': (ctypes.c_char_p, []),
    'rtm3d_abi_version': (c_int, []),
    'rtm3d_ctx_create': (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    'rtm3d_ctx_destroy': (None, [c_void_p]),
    'rtm3d_tensor_create': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int)]),
    'rtm3d_tensor_download': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    'rtm3d_tensor_upload': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    'rtm3d_blob_create': (c_int, [c_void_p, c_void_p, c_size_t, ctypes.POINTER(c_int)]),
    'rtm3d_tensor_info': (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p)] + [ctypes.POINTER(c_int)] * 5),
    'rtm3d_blob_address': (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t)]),
    'rtm3d_op_input_nhwc4': (c_int, [c_void_p, c_int]),
    'rtm3d_op_conv': (c_int, [c_void_p, ctypes.POINTER(ConvDesc)]),
    'rtm3d_op_stem_fused': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'rtm3d_op_conv32s2_fused': (c_int, [c_void_p] + [c_int] * 10),
    'rtm3d_op_conv64_root': (c_int, [c_void_p] + [c_int] * 16),
    'rtm3d_op_headout': (c_int, [c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int)]),
    'rtm3d_op_patch_mask': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int]),
    'rtm3d_gather_peak_patches': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_int, c_size_t]),
    'rtm3d_decode2d_finish': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    'rtm3d_op_maxpool': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'rtm3d_op_maxpool_s2d': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int]),
    'rtm3d_op_softmax_fuse': (c_int, [c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_int)]),
    'rtm3d_forward': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p)]),
    'rtm3d_ctx_set_graph': (c_int, [c_void_p, c_int]),
    'rtm3d_ctx_debug_memset_in_replay': (c_int, [c_void_p, c_int]),
    'rtm3d_ctx_debug_read_words': (c_int, [c_void_p, c_int, c_int, c_void_p]),
    'rtm3d_ctx_graph_stats': (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'rtm3d_forward_timed': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_float), c_int, ctypes.POINTER(c_int)]),
    'rtm3d_forward_marks': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_float)]),
    'rtm3d_op_info': (c_int, [c_void_p, c_int, ctypes.POINTER(c_double), ctypes.POINTER(c_double), ctypes.POINTER(ctypes.c_char_p)]),
    'rtm3d_probe_set': (c_int, [c_void_p, c_int]),
    'rtm3d_probe_read': (c_int, [c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_int)]),
    'rtm3d_decode2d_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'rtm3d_decode2d': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_float,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rtm3d_decode3d': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_int]),
    'rtm3d_pack_records': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_double, c_void_p]),
    'rtm3d_project_boxes': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rtm3d_decode_smoke': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                                   c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rtm3d_preprocess': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'rtm3d_preprocess_batch': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                       c_void_p, c_void_p]),
    'rtm3d_input_tensor': (c_int, [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                   ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'rtm3d_stream_create_cumask': (c_int, [c_int, c_int, ctypes.POINTER(c_void_p)]),
    'rtm3d_stream_destroy': (c_int, [c_void_p]),
    'rtm3d_decode3d_reference_form': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p]),
    'rtm3d_decode3d_scalar': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    'rtm3d_verify_conv_f32': (c_int, [c_void_p, ctypes.POINTER(VConvDesc)]),
    'rtm3d_verify_maxpool_f32': (c_int, [c_void_p, ctypes.POINTER(VTensor), ctypes.POINTER(VTensor), c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'rtm3d_verify_softmax_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'rtm3d_verify_softmax_fuse_f32': (c_int, [c_void_p, ctypes.POINTER(VTensor), ctypes.POINTER(VTensor), c_int, ctypes.POINTER(VTensor),
                                              c_int, c_int, c_int, c_int, c_void_p]),
    'rtm3d_decode3d_slots': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
}

_lib = None


def load():
    """Load (once) and return the ctypes library with typed entry points.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('rtm3d_amd: HIP library %s is missing - build it with '
                           '`python -c "import __graft_entry__ as g; g.build()"` (or `make -C rtm3d_amd/csrc`). '
                           'There is no CPU fallback.' % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError('rtm3d_amd: cannot load %s: %s (ROCm runtime present? there is no CPU fallback)' % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = library/ABI mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.rtm3d_abi_version() != ABI_VERSION:
        raise RuntimeError('rtm3d_amd: ABI version mismatch (library %d, binding %d)' % (lib.rtm3d_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().rtm3d_last_error()
        raise RuntimeError('rtm3d_hip %s failed: %s' % (what, msg.decode() if msg else 'unknown error'))
