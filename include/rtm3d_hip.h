/*
 * rtm3d_hip.h - C ABI of librtm3d_hip.so, the MI355X (gfx950) implementation of the RTM3D
 * inference hot path.
 *
 * The reference (hitfeelee/rtm3d) has no native code and no FFI: its "operator API" for this path
 * is the Python surface  Model.forward / Model.inference / optim_decode_bbox3d.  Each entry point
 * below names the reference interface (file:line under /root/reference) whose arithmetic it
 * replaces; the Python binding a maintainer would add is shown in INTEGRATION.md and is what
 * rtm3d_amd/_lib.py does with ctypes.
 *
 * Conventions
 *   - plain C types only; every pointer named d_* is a DEVICE pointer owned by the caller
 *     (e.g. torch.Tensor.data_ptr()), every h_* pointer is HOST memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are stream-ordered
 *     and never synchronise unless stated;
 *   - every function returns 0 on success, non-zero on error; rtm3d_last_error() then returns a
 *     thread-local description.  "No detections" is not an error (n_out[b] == 0), mirroring the
 *     `None` list entries of models/model.py:33-44;
 *   - a context is not re-entrant: one caller thread at a time, one context per process-GPU (the reference is
 *     single-threaded per process, train_multi_gpu.py:243).  Replays of one context never overlap on the device: calls on
 *     one stream are ordered by the stream, a call on a different stream is ordered behind the previous replay by an event.
 */
#ifndef RTM3D_HIP_H
#define RTM3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTM3D_ABI_VERSION 9
#define RTM3D_MAX_GROUPS 4
#define RTM3D_MAX_TAPS 80

typedef struct rtm3d_ctx rtm3d_ctx;

const char* rtm3d_last_error(void);
int rtm3d_abi_version(void);

/* ------------------------------------------------------------------ context / plan building
 * A context owns the activation workspace (padded NHWC fp16 tensors), the packed + BN-folded
 * weights and an ordered list of kernel launches ("the plan") for ONE network at ONE input shape.
 * The host (Python, rtm3d_amd/plan.py) records the plan once; rtm3d_forward replays it.
 * Replaces: the nn.Module graph walked by Model.forward, models/model.py:20-27.                  */
int rtm3d_ctx_create(int device, rtm3d_ctx** out);
void rtm3d_ctx_destroy(rtm3d_ctx* ctx);

/* Activation tensor: [B][H+2*pad][W+2*pad][C] fp16, zero border (written once at creation and
 * never touched by kernels, which is what implements the convolutions' zero padding).           */
int rtm3d_tensor_create(rtm3d_ctx* ctx, int B, int H, int W, int C, int pad, int* id);
/* Debug / parity helpers: copy channels [c0, c0+C) of the interior to/from fp32 NCHW host memory
 * (synchronous).                                                                                 */
int rtm3d_tensor_download(rtm3d_ctx* ctx, int id, int c0, int C, float* h_nchw);
int rtm3d_tensor_upload(rtm3d_ctx* ctx, int id, int c0, int C, const float* h_nchw);

/* Device blob (packed weights / folded biases).  Copies `bytes` from host; returns blob id.      */
int rtm3d_blob_create(rtm3d_ctx* ctx, const void* h_data, size_t bytes, int* id);
/* Device address and geometry of a plan tensor / device address of a blob: for kernels that fill or read plan storage from
 * outside the replay (rtm3d_gather_peak_patches writes the patch plan's input tensor and its (y, x) blob).  Any output
 * pointer of rtm3d_tensor_info may be NULL.  d_base = padded element [0][0][0][0] (NHWC fp16, border `border`).         */
int rtm3d_tensor_info(rtm3d_ctx* ctx, int id, void** d_base, int* B, int* H, int* W, int* C, int* border);
int rtm3d_blob_address(rtm3d_ctx* ctx, int id, void** d_ptr, size_t* bytes);

/* Copy the caller's fp32 NCHW (B,3,H,W) image into a 4-channel padded NHWC fp16 tensor (4th channel
 * zero, border >= 4): operand layout of the register-direct MFMA stem (kernel = 3 with cin = 4).      */
int rtm3d_op_input_nhwc4(rtm3d_ctx* ctx, int out_tensor);

/* Generic convolution descriptor (one launch; `groups` independent sub-problems on grid.z).
 * Covers conv KxK (any stride / dilation), 1x1 over channel slices of wider tensors (the DLA root
 * "concat" never materialises: producers write slices), grouped head convs, and the four
 * sub-pixel phases of ConvTranspose2d(k4,s2,p1) (models/nets/module.py:7-15).
 * Iteration domain: m in [0, B*Hm*Wm) -> (n, y, x).
 *   input  pixel (n, y*in_stride + tap_dy[g][t], x*in_stride + tap_dx[g][t])   (unpadded coords),
 *          channels [in_coff[g] + tap_dc[g][t], ... + cin) of the input tensor (tap_dc = 0 for an ordinary convolution;
 *          a tap may name another channel slice of the same tensor: a DLA block's `project` 1x1 on the pooled map
 *          (models/nets/dla.py:175-198) becomes extra K-steps of the block's second conv instead of a residual, with the
 *          conv's own taps split into 64-channel pseudo-taps so that every tap carries cin = 64 channels)
 *   output pixel (n, y*out_scale + out_oy[g],    x*out_scale + out_ox[g])
 * Epilogue: + bias[cout] (BN folded) [+ residual at the output pixel] [ReLU] -> fp16 NHWC,
 * or (out_nchw_f32 != 0) fp32 NCHW into the caller buffer given to rtm3d_forward.                */
typedef struct rtm3d_conv_desc {
    int in_tensor, out_tensor, res_tensor; /* res_tensor < 0: none; out_tensor < 0 with out_nchw_f32 */
    int Hm, Wm;                            /* iteration domain per image */
    int in_stride, out_scale;
    int cin, cout;                         /* per group (cout = real output channels) */
    int groups, ntaps;
    int in_coff[RTM3D_MAX_GROUPS], out_coff[RTM3D_MAX_GROUPS], res_coff[RTM3D_MAX_GROUPS];
    int out_oy[RTM3D_MAX_GROUPS], out_ox[RTM3D_MAX_GROUPS];
    int tap_dy[RTM3D_MAX_GROUPS][RTM3D_MAX_TAPS], tap_dx[RTM3D_MAX_GROUPS][RTM3D_MAX_TAPS];
    int tap_dc[RTM3D_MAX_GROUPS][RTM3D_MAX_TAPS]; /* channel offset of the tap relative to in_coff (multiple of 8; kernels 0 and 2 only) */
    int s2d_tensor, s2d_coff;              /* s2d_tensor = tensor id + 1, 0 = none (ABI 8; like out_nchw_f32, so that a zero-initialised descriptor asks
                                              for nothing).  Set (kernel 0 or 5, groups 1, out_scale 1, even output height / width): the output is
                                              written a SECOND time in space-to-depth layout - pixel (y, x) to pixel (y >> 1, x >> 1), channels
                                              s2d_coff + ((y & 1) * 2 + (x & 1)) * cout + c of the half-resolution s2d_tensor - so that the neck
                                              can read the feature map from the grid of its transposed conv's input (plan.py: _neck_up_folds).
                                              With out_tensor < 0 (kernel 0) ONLY this copy is written: its readers are rtm3d_op_maxpool_s2d, a
                                              stride-2 conv restated on the copy (stride-1 taps with tap_dc, or kernel 7 with in_s2d) and tap_dc taps. */
    int in_s2d;                            /* kernel 7 only: in_tensor holds the SPACE-TO-DEPTH copy (half resolution, 4 x cin channels at in_coff) of the
                                              map the conv is stated on (Hm, Wm, taps, stride as for the ordinary map) */
    int relu;
    int w_blob, bias_blob;                 /* packed fp16 weights (layout depends on `kernel`), fp32 bias [groups][cout_pad] */
    int kernel;                            /* 0 = MFMA implicit GEMM 128-px tile (cin % 64 == 0), 2 = MFMA 256x256 tile
                                              (cout % 256 == 0), 3 = register-direct MFMA (cin 4/16/32), 5 = 64->64 3x3 halo kernel,
                                              filter bank in registers (W % 32 == 0, H % 8 == 0), 6 = 3x3 halo kernel for cin, cout % 128 == 0, weights
                                              in the bn_tile = 128 packing of kernel 0 (W % 32 == 0, H % 8 == 0), 7 = 64->128 3x3 STRIDE-2 halo kernel, filter bank in
                                              registers (output W % 32 == 0, H % 4 == 0; weights fp16 [9 taps][2 k halves][8 tiles][64 lanes][8]); 1 is retired */
    int bn_tile;                           /* MFMA: cout tile the weights were packed for (16/32/64/128) */
    int out_nchw_f32;                      /* 0, or 1..4 = index+1 into rtm3d_forward's out_logits[] */
    int out_H, out_W;                      /* only for out_nchw_f32 */
    int softmax_stat_slot;                 /* -1, or 0..2: the output map is operand u[slot] of the NEXT rtm3d_op_softmax_fuse and this
                                              launch also emits its spatial-softmax partials (per channel max / sum exp over each
                                              128-pixel run) from the epilogue, so the fusion does not re-read the map to reduce it
                                              (keypoint_fpn_fusion.py:67).  Only for kernel = 2 launches that take the halo-tile kernel
                                              (taps within +-1 pixel, stride 1, Hm % 8 == 0, Wm % 32 == 0) with cout = 256 written at
                                              channel offset 0 of a 256-channel tensor; anything else is refused. */
} rtm3d_conv_desc;
int rtm3d_op_conv(rtm3d_ctx* ctx, const rtm3d_conv_desc* desc);

/* DLA-34 stem, fused (models/nets/dla.py:259-279): base_layer 7x7 3->16 + BN + ReLU, level0 3x3 16->16 + BN + ReLU and - when
 * w_l1_blob >= 0 - level1 3x3 stride 2 16->32 + BN + ReLU in ONE launch; the 16-channel full-resolution intermediate maps stay
 * in LDS.  x4_tensor: the NHWC4 image tensor (border >= 4, H % 16 == 0, W % 32 == 0); out_tensor: level0's output (full
 * resolution, 16 channels at out_coff) or level1's (half resolution, 32 channels).  Weights packed as for rtm3d_op_conv
 * kernel = 3 (cin = 4: [7][64][8]; cin = 16: [cout/16][5][64][8]), fp32 biases with BN folded.  Same result as the
 * rtm3d_op_conv launches it replaces up to fp32 summation order.                                                        */
int rtm3d_op_stem_fused(rtm3d_ctx* ctx, int x4_tensor, int out_tensor, int out_coff, int w_base_blob, int b_base_blob,
                        int w_l0_blob, int b_l0_blob, int w_l1_blob, int b_l1_blob);

/* Entry of a DLA level-1 tree with stride 2 on a 32-channel map (DLA-34 level2, models/nets/dla.py:186-206), three reference ops in
 * one launch reading the input once: bottom = max_pool2d(x, 2, 2) (never materialised), proj_tensor = BN(conv1x1(bottom)) (the
 * `project` branch, no ReLU), conv_tensor = ReLU(BN(conv3x3 stride 2 (x))) (tree1.conv1); 32 -> 64 channels each.  Input: 32
 * channels at in_coff of a tensor with border >= 1, H % 16 == 0, W % 64 == 0.  Weights fp16 [9 taps][4][64 lanes][8] and
 * [4][64][8] (MFMA A fragments, K = 32), fp32 biases [64] with BN folded.                                                      */
int rtm3d_op_conv32s2_fused(rtm3d_ctx* ctx, int in_tensor, int in_coff, int conv_tensor, int conv_coff, int proj_tensor, int proj_coff,
                            int w_conv_blob, int b_conv_blob, int w_proj_blob, int b_proj_blob);

/* Tail of a DLA level-1 tree on 64 channels (DLA-34 level2) in one launch, three reference ops reading x1 once and never
 * writing x2: x2 = ReLU(BN(conv3x3(in)) + x1) (tree2's BasicBlock.conv2 + residual, models/nets/dla.py:92-99), out =
 * ReLU(BN(conv1x1(cat[x2, x1]))) (Root.forward, dla.py:233-241) and - when pool_tensor >= 0 - pool = max_pool2d(out, 2, 2)
 * (the next level's `downsample`, dla.py:170-172,190).  in / x1 (res) / out: 64 channels at *_coff of tensors of equal
 * shape, H % 8 == 0, W % 32 == 0, `in` with a border >= 1; pool: 64 channels at half resolution.  w_conv_blob: the
 * kernel = 5 packing of rtm3d_op_conv ([9][2][4][64 lanes][8]); w_root_blob: fp16 [4 output tiles][4 K-steps][64 lanes][8]
 * with lane = fk * 16 + row, element j = root weight [tile * 16 + row][s * 32 + (j >> 2) * 16 + fk * 4 + (j & 3)] over the
 * concatenated input [x2 | x1] (the K order in which the conv's accumulator fragments are handed to the root's MFMAs);
 * fp32 biases [64] with BN folded.  Same result as the rtm3d_op_conv / rtm3d_op_maxpool launches it replaces up to fp32
 * summation order.  out_tensor < 0: the ordinary copy of `out` is not written (needs s2d_tensor).  s2d_tensor >= 0 (a plain tensor id here, -1 = none): `out` is written a second time in space-to-depth layout - pixel (y, x) to half-resolution
 * pixel (y >> 1, x >> 1), channels s2d_coff + ((y & 1) * 2 + (x & 1)) * 64 + c of s2d_tensor - which lets the neck read the
 * feature map at the resolution of the transposed conv's INPUT grid (rtm3d_amd/plan.py: RealizedPlan._neck_up_folds).        */
int rtm3d_op_conv64_root(rtm3d_ctx* ctx, int in_tensor, int in_coff, int res_tensor, int res_coff, int conv_relu,
                         int w_conv_blob, int b_conv_blob, int w_root_blob, int b_root_blob,
                         int out_tensor, int out_coff, int root_relu, int pool_tensor, int pool_coff,
                         int s2d_tensor, int s2d_coff);

/* The four final 3x3 convolutions of the heads in one launch (models/nets/header.py:17,27,32,37):
 * input = the nheads x 256-channel tensor written by the grouped head conv (nheads = 4, or 2 for the
 * "smoke" head table), outputs = rtm3d_forward's first nheads fp32 NCHW logit buffers, cout4[i] channels.  Weights: fp16 [head][tap][8 k-blocks][64 lanes][8]
 * (MFMA fragment order, 16 zero-padded rows per head), bias fp32 [head][16].                          */
int rtm3d_op_headout(rtm3d_ctx* ctx, int in_tensor, int w_blob, int bias_blob, int nheads, const int* cout4);
/* Patch plans (peaks-only regression heads, csrc/sparse_heads.hip): `tensor` holds one S x S window per "image" (= detection
 * slot, no border); window position (i, j) of slot s lies at map pixel (y_s + i - origin, x_s + j - origin) with (y_s, x_s)
 * from `yx_blob` ([slots][2] int32, -1 = empty slot; written per batch by rtm3d_gather_peak_patches).  Positions outside
 * the img_H x img_W map are zeroed: they are the zero padding of the next convolution (models/nets/header.py:24-37).  */
int rtm3d_op_patch_mask(rtm3d_ctx* ctx, int tensor, int yx_blob, int img_H, int img_W, int origin);

/* Max pooling k x k / stride / pad over channel slice, NHWC fp16 (models/nets/dla.py:170-172,
 * models/nets/resnet.py:128).  Inputs are post-ReLU (>= 0) so the zero border equals -inf padding. */
int rtm3d_op_maxpool(rtm3d_ctx* ctx, int in_tensor, int in_coff, int out_tensor, int out_coff,
                     int channels, int ksize, int stride, int pad);

/* z_out = z + sum_i u_i * softmax_{H*W}(u_i)  per (image, channel)
 * (models/nets/keypoint_fpn_fusion.py:60-69).  n_u <= 3.                                         */
int rtm3d_op_softmax_fuse(rtm3d_ctx* ctx, int z_in, int z_out, int n_u, const int* u_tensors);

/* 2x2 / stride 2 max-pool (a DLA level's `downsample`, models/nets/dla.py:170-172) of a map that exists only as its space-to-depth
 * copy (rtm3d_conv_desc.s2d_tensor with out_tensor < 0): the window = the four channel slices of one pixel of in_tensor
 * (`channels` each, from in_coff); output: `channels` at out_coff of out_tensor, same resolution as in_tensor.                */
int rtm3d_op_maxpool_s2d(rtm3d_ctx* ctx, int in_tensor, int in_coff, int out_tensor, int out_coff, int channels);

/* Replay the plan.  d_in: fp32 NCHW (B,3,H,W) normalised image batch (detect.py:53), or NULL when the input tensor was
 * filled by rtm3d_preprocess_batch (out_mode 1);
 * d_out_logits[4]: fp32 NCHW (B,3|16|2|2,H/4,W/4) = pred_logits of models/model.py:22-27.        */
int rtm3d_forward(rtm3d_ctx* ctx, void* stream, const float* d_in, float* const d_out_logits[4]);

/* enable != 0: rtm3d_forward replays the plan as ONE hipGraph launch instead of ~60 kernel launches (small batches are
 * launch-gap bound; the reference's detect.py runs bs = 1).  The graph is captured on first use per distinct
 * (d_in, d_out_logits[0..3]) pointer tuple - kernel arguments are baked into it - and up to 8 tuples are cached (LRU).
 * Results are bit-identical to the eager replay.  The live probe (rtm3d_probe_set) forces the eager path.
 * RULE - ALL-KERNEL GRAPHS ONLY: nothing that runs during a replay may be a hipMemset*Async / hipMemcpy*Async (they become
 * memset / memcpy graph NODES).  With three or more graph execs of different node counts alive in one context, re-launching an
 * older exec ran the kernel node behind its memset node with stale arguments (round 3: a conversion pass silently had no effect,
 * a convolution took a memory-access fault; all-kernel graphs alternate correctly), which is why the ticket counters are zeroed
 * by a kernel.  The rule is enforced: after every capture the node types are enumerated (hipGraphGetNodes /
 * hipGraphNodeGetType); a graph with any non-kernel node is destroyed, the context leaves graph mode for good, the call is
 * served by the eager replay and rtm3d_last_error() says why (rtm3d_ctx_graph_stats reports enabled = 0).               */
int rtm3d_ctx_set_graph(rtm3d_ctx* ctx, int enable);
/* TEST HOOK for the rule above: enable != 0 puts one hipMemsetAsync in front of every replay, so that a capture holds a memset
 * node and must be refused (tests/test_gpu_kernels.py::test_graph_with_memset_node_is_refused).  Never set by the product.  */
int rtm3d_ctx_debug_memset_in_replay(rtm3d_ctx* ctx, int enable);
/* DIAGNOSTICS, unsupported: copies n 32-bit words, starting at word `offset`, of the context's debug area to the host (after a
 * device synchronisation).  Only diagnostic builds of the library write there (-DC256_STAMPS: in-kernel s_memtime stamps of the
 * persistent 256 x 256 convolution, tools/gpu_c256_stamps.sh); in the product build the words read zero.                   */
int rtm3d_ctx_debug_read_words(rtm3d_ctx* ctx, int offset, int n, unsigned int* h_out);
/* Graph bookkeeping of a context: captures made, replays served from the cache, whether graph mode is still on (a caller that
 * hands over fresh buffers on every call makes every call a capture; after 32 captures without as many hits the context gives
 * up on graphs - Model.forward_logits(out=...) keeps the addresses stable).  Any pointer may be NULL.                      */
int rtm3d_ctx_graph_stats(rtm3d_ctx* ctx, int* captures, int* hits, int* enabled);

/* Wall time of STAGES of one eager replay: events on the caller's stream in front of the ops mark_ops[0..n_marks) (ascending op
 * indices; n_marks <= 8) and behind the last op; h_ms[i] = mark i -> mark i + 1 (the last: -> end).  (ABI 9: the side-lane replay
 * schedule of ABI 8 - rtm3d_op_schedule / rtm3d_ctx_set_lanes - is gone: measured as no gain for the forward and a slower
 * two-stream pipeline, profiles/r05_neck_lanes.txt.)                                                                        */
int rtm3d_forward_marks(rtm3d_ctx* ctx, void* stream, const float* d_in, float* const d_out_logits[4],
                        int n_marks, const int* mark_ops, float* h_ms);

/* Per-op timing of one replay with hipEvents (synchronous; for profiling/bench):
 * h_ms[i] = elapsed ms of op i; returns number of ops through *n_ops (h_ms may be NULL).          */
int rtm3d_forward_timed(rtm3d_ctx* ctx, void* stream, const float* d_in, float* const d_out_logits[4],
                        float* h_ms, int cap, int* n_ops);
/* Algorithmic work of op i: flops (2*MAC) and minimum bytes moved; name is a static string.       */
int rtm3d_op_info(rtm3d_ctx* ctx, int i, double* flops, double* bytes, const char** name);

/* Live probe for roofline accounting: record a hipEvent pair around op `op_index` on the caller's
 * stream in every rtm3d_forward (op_index < 0 disables); rtm3d_probe_read synchronises on the
 * recorded events and returns the average duration of the (up to 64) most recent launches.          */
int rtm3d_probe_set(rtm3d_ctx* ctx, int op_index);
int rtm3d_probe_read(rtm3d_ctx* ctx, double* avg_ms, int* count);

/* ------------------------------------------------------------------ 2D decode
 * sigmoid -> 3x3 equality NMS -> top-k -> threshold -> gather/regress 8 vertices + 2D box.
 * Replaces Model.inference, models/model.py:29-98,117-132 and nms_hm, utils/model_utils.py:17-26.
 * Inputs fp32 NCHW logits.  Outputs (all device, caller-owned), image b at row b:
 *   d_n[B] int32 count; d_cls[B*topk] int64; d_score[B*topk]; d_mproj[B*topk*2];
 *   d_verts[B*topk*16] (8 x (x,y)); d_bbox[B*topk*4] (x1,y1,x2,y2); rows >= n are untouched.
 * Order within an image: score descending, ties by ascending flat index (class-major).
 * d_workspace: at least rtm3d_decode2d_workspace_bytes(B, ncls, H, W) bytes.
 * Peaks-only mode (d_offset_fr_main == d_main_offset == NULL, used by the "smoke" head table): only
 * d_n, d_cls, d_score and d_mproj = integer key point (x, y) are written.                            */
size_t rtm3d_decode2d_workspace_bytes(int B, int ncls, int H, int W);
int rtm3d_decode2d(void* stream, const float* d_main_kf, const float* d_offset_fr_main,
                   const float* d_main_offset, int B, int ncls, int H, int W, float score_thresh,
                   int topk, float down_sample, void* d_workspace, int32_t* d_n, int64_t* d_cls,
                   float* d_score, float* d_mproj, float* d_verts, float* d_bbox);

/* Peaks-only regression heads: what Model.inference reads of the regression maps is their value at the <= topk peaks
 * (models/model.py:47-50,124-128: offset_fr_main and main_offset through two gathers; vertex_offset is never read).
 * rtm3d_gather_peak_patches: after rtm3d_decode2d in its peaks-only mode, copy for every live slot the samples of the fused
 * map z (padded NHWC fp16, 256 channels, border z_pad) that the three head convolutions of a peak depend on into a
 * 15 x 15 x 256 patch (layout: csrc/sparse_heads.hip) and the peak's (y, x) into d_yx; empty slots get (-1, -1).
 * The caller states the CAPACITY of what is written: d_patch holds patch_slots windows of patch_S x patch_S x 256 halves, d_yx
 * yx_bytes bytes; the call is refused unless patch_S == 15, patch_slots >= B * topk and yx_bytes >= 8 * B * topk.
 * rtm3d_decode2d_finish: sub-pixel key point, 8 vertices and the 2D box of every live slot from the regression logits
 * evaluated at its peak ([B*topk][16] and [B*topk][2] fp32) - the second half of rtm3d_decode2d, same fp32 operation order;
 * d_mproj holds the integer key points on entry and the sub-pixel ones (x down_sample) on return.                      */
int rtm3d_gather_peak_patches(void* stream, const void* d_z, int z_H, int z_W, int z_C, int z_pad, int B, int topk,
                              const int32_t* d_n, const float* d_peak_xy, void* d_patch, int32_t* d_yx,
                              int patch_slots, int patch_S, size_t yx_bytes);
int rtm3d_decode2d_finish(void* stream, int B, int topk, const int32_t* d_n, const float* d_reg_offset_fr_main,
                          const float* d_reg_main_offset, float down_sample, float* d_mproj, float* d_verts, float* d_bbox);

/* ------------------------------------------------------------------ 3D decode
 * Per object: minimise the 8-corner reprojection error over [sin,cos,l,h,w,X,Y,Z] with an fp64
 * L-BFGS-B (m=10, factr=1e7, pgtol=1e-5, maxls=20, unbounded) started at
 * [0,1,l_ref,h_ref,w_ref,ref_loc].  Replaces optim_decode_bbox3d, utils/model_utils.py:264-312
 * (objective :155-177, gradient :206-234) and scipy.optimize.minimize(method='L-BFGS-B').
 * Iteration driver, More'-Thuente line search, BFGS skip rule and stopping tests follow L-BFGS-B 3.0 step by step.
 * `form` selects how the search direction -B^-1 g of the unbounded case is computed (one wavefront per object either way):
 *   RTM3D_SOLVER_PUBLISHED  L-BFGS-B 3.0's published subspace step (formk / subsm / formt) - the arithmetic SciPy runs behind
 *                           utils/model_utils.py:295-296, operation for operation.  THE PRODUCT'S DEFAULT since ABI 9 (the
 *                           Python facade passes it unless told otherwise): on every reference-run fixture all kept boxes
 *                           are within 1e-4 of SciPy's.  Bit-identical to rtm3d_decode3d_reference_form.
 *   RTM3D_SOLVER_DIRECT     the two-loop recursion over the stored pairs - the same vector in exact arithmetic, a third of the
 *                           dependent fp64 operations per iteration (rtm3d_amd/csrc/lbfgsb.h); keep / reject identical, but one
 *                           kept object in ~1000 stops an iteration apart from SciPy (99.1-100 % within 1e-4 per fixture;
 *                           DESIGN.md section 4).  Opt-in.  Bit-identical to rtm3d_decode3d_scalar.
 *   d_cls[N] int64, d_verts[N*16] fp32, d_K[N*9] fp64 (row-major 3x3 per object),
 *   d_dim_ref[ncls*3] fp64 (h,w,l), d_ref_loc[3] fp64.
 * Outputs: d_x[N*8] fp64 final iterate, d_fun[N] fp64, d_nit[N] int32, d_status[N] int32
 * (0 converged, 1 max iterations, 2 abnormal line search, 3 non-finite objective at the start point:
 * NaN / Inf key points give x = x0, fun = NaN, nit = 0 like SciPy does).  The caller applies `fun < 0.1`. */
#define RTM3D_SOLVER_DIRECT 0
#define RTM3D_SOLVER_PUBLISHED 1
int rtm3d_decode3d(void* stream, int N, const int64_t* d_cls, const float* d_verts, const double* d_K,
                   const double* d_dim_ref, int ncls, const double* d_ref_loc, double* d_x,
                   double* d_fun, int32_t* d_nit, int32_t* d_status, int form);

/* Same solver over the fixed-size slots written by rtm3d_decode2d, without a host round trip:
 * slot i = (image i / topk, rank i % topk) is solved iff rank < d_n[image]; other slots get
 * status -1 and are otherwise untouched.  d_K_per_image[B*9].  Outputs have B*topk rows.
 * form: as for rtm3d_decode3d.                                                                                              */
int rtm3d_decode3d_slots(void* stream, int B, int topk, const int32_t* d_n, const int64_t* d_cls,
                         const float* d_verts, const double* d_K_per_image, const double* d_dim_ref, int ncls,
                         const double* d_ref_loc, double* d_x, double* d_fun, int32_t* d_nit, int32_t* d_status, int form);

/* Fixed-size detection records for collecting the results of a sharded batch (new functionality: the reference's
 * inference is single-GPU, detect.py:18): slot (image, rank) -> 32 fp32 =
 *   [0] class  [1] score  [2:4] main key point  [4:20] 8 vertices (x, y)  [20:24] 2D box
 *   [24:27] dimension (h, w, l)  [27:30] location  [30] Ry = atan2(x0, x1)    (the ParamList fields of
 *   utils/model_utils.py:300-303, rounded to fp32)   [31] flag: 0 empty, 1 2D only, 2 3D kept (fun < fun_accept, :298).
 * Slots with rank >= d_n[image] are written as 32 zeros.  d_x/d_fun/d_status (solver outputs of
 * rtm3d_decode3d_slots) may all be NULL: fields 24..30 are then zero and the flag is 0/1.  d_rec: B*topk*32 floats. */
int rtm3d_pack_records(void* stream, int B, int topk, const int32_t* d_n, const int64_t* d_cls, const float* d_score,
                       const float* d_mproj, const float* d_verts, const float* d_bbox, const double* d_x,
                       const double* d_fun, const int32_t* d_status, double fun_accept, float* d_rec);

/* Box post-processing on the device: for every solved slot (d_status >= 0) the eight corners + the centre of the
 * box x = [sin, cos, l, h, w, X, Y, Z] projected through K - calc_proj_corners / create_corners / rotation_matrix,
 * utils/model_utils.py:66-152 - and the bounding rectangle of the eight corners (the 2D box a KITTI label line carries).
 * d_K: topk > 0 -> one K (9 doubles) per image, slot i belongs to image i / topk; topk == 0 -> one K per slot.
 * Outputs: d_proj[N][9][2], d_rect[N][4] = (x1, y1, x2, y2) fp64; unsolved slots get zeros.                                     */
int rtm3d_project_boxes(void* stream, int N, int topk, const double* d_x, const int32_t* d_status, const double* d_K,
                        double* d_proj, double* d_rect);

/* "smoke" head-table variant (its source is not in the reference snapshot: PARITY
 * UNPINNED, published SMOKE formulation): closed-form box from the 8 regression channels at each key
 * point of the peaks-only decode.  Outputs use the solver's layout x = [sin ry, cos ry, l, h, w, X, Y, Z]. */
int rtm3d_decode_smoke(void* stream, int B, int topk, const int32_t* d_n, const int64_t* d_cls, const float* d_peak_xy,
                       const float* d_reg, int H, int W, float down_sample, const double* d_K_per_image,
                       const double* d_dim_ref, int ncls, double* d_x, double* d_fun, int32_t* d_nit, int32_t* d_status);

/* Input pipeline step in front of the path: letterbox an already resized uint8 HWC
 * image (h x w x 3, channel order as loaded) into the H x W canvas, centred, border = the image's mean
 * colour truncated to uint8 (datasets/dataset_reader.py:175-195), then Normalize/ToTensor/ToNCHW
 * (preprocess/transforms.py:110-120,312-322) -> fp32 CHW at d_out_chw.  d_lut: fp32 [3][256] =
 * float32((v/255. - mean[c]) / std[c]) computed in float64 like the reference; d_sums3: 3 x uint64 scratch. */
int rtm3d_preprocess(void* stream, const uint8_t* d_img_hwc, int h, int w, float* d_out_chw, int H, int W,
                     const float* d_lut, unsigned long long* d_sums3);

/* The same step for a whole batch of ragged images in TWO launches (interior + per-image channel sums, then borders),
 * with the bilinear Resize of the reference's TestTransform in front (preprocess/transforms.py:480-495,
 * cv2.resize INTER_LINEAR: OpenCV's published 11-bit fixed-point algorithm restated - OpenCV is an un-vendored
 * dependency absent from this image, so the resize itself is PARITY UNPINNED; equal source and target sizes are the
 * identity and then the result is bit-identical to rtm3d_preprocess).
 *   h_imgs[B]: HOST array of DEVICE pointers to uint8 HWC images; h_hw[2B] = (h, w) per image;
 *   h_resized_hw[2B] = (h', w') after Resize, or NULL for "already resized";
 *   out_mode 0: d_out = fp32 NCHW (B,3,H,W), the reference's `imgs` (detect.py:53);
 *   out_mode 1: d_out = the network's own operand, fp16 NHWC4 [B][H+2*out_border][W+2*out_border][4] (4th channel 0;
 *               the border itself is not written) - see rtm3d_input_tensor / rtm3d_forward with d_in == NULL;
 *   d_lut fp32 [3][256] as above, d_lut16 the same table rounded to fp16 (mode 1), d_sums: B*3 uint64 scratch. */
int rtm3d_preprocess_batch(void* stream, int B, const uint8_t* const* h_imgs, const int* h_hw, const int* h_resized_hw,
                           void* d_out, int out_mode, int H, int W, int out_border, const float* d_lut, const void* d_lut16,
                           unsigned long long* d_sums);

/* Device address and geometry of the plan's 4-channel fp16 input tensor (written by rtm3d_op_input_nhwc4 from the caller's
 * fp32 batch, or directly by rtm3d_preprocess_batch in out_mode 1, after which rtm3d_forward is called with d_in == NULL
 * and skips the conversion).  Fails if the plan has no such tensor.                                                  */
int rtm3d_input_tensor(rtm3d_ctx* ctx, void** d_base, int* B, int* H, int* W, int* border);

/* Stream restricted to `n_cus` compute units (hipExtStreamCreateWithCUMask) for the latency-bound
 * 3D decode of the two-stream pipeline; destroy with rtm3d_stream_destroy.                          */
int rtm3d_stream_create_cumask(int device, int n_cus, void** stream);
int rtm3d_stream_destroy(void* stream);

/* Cross-check entries, the arguments of rtm3d_decode3d without `form`, one lane per object (slow; parity tests only):
 *   rtm3d_decode3d_scalar          the direct form: results bit-identical to rtm3d_decode3d(form = RTM3D_SOLVER_DIRECT);
 *   rtm3d_decode3d_reference_form  L-BFGS-B 3.0 with its published subspace step (formk / subsm / formt), the form
 *                                  scipy.optimize.minimize(method='L-BFGS-B') runs (utils/model_utils.py:295-296): bit-identical
 *                                  to rtm3d_decode3d(form = RTM3D_SOLVER_PUBLISHED).                                            */
int rtm3d_decode3d_scalar(void* stream, int N, const int64_t* d_cls, const float* d_verts, const double* d_K,
                          const double* d_dim_ref, int ncls, const double* d_ref_loc, double* d_x,
                          double* d_fun, int32_t* d_nit, int32_t* d_status);
int rtm3d_decode3d_reference_form(void* stream, int N, const int64_t* d_cls, const float* d_verts, const double* d_K,
                                  const double* d_dim_ref, int ncls, const double* d_ref_loc, double* d_x,
                                  double* d_fun, int32_t* d_nit, int32_t* d_status);

/* ------------------------------------------------------------------ fp32 verification executor (regime ii)
 * The product path stores activations and weights in fp16; BASELINE's "3D-box L-inf vs CPU ref" through fp16 logits is
 * bounded by that storage.  These three stateless entry points run the SAME recorded plan
 * (rtm3d_amd/plan.py: tap tables, channel slices, sub-pixel phases, folded BN, composed 1x1 pairs) on padded NHWC *fp32*
 * tensors with fp32 weights and fp64 accumulation, so that Model.forward_logits_fp32 -> rtm3d_decode2d ->
 * rtm3d_decode3d_slots can be compared with the reference's fp32 CPU path (models/model.py:20-27, :29-75,
 * utils/model_utils.py:264-312) to fp32 round-off.  Verification only: simple kernels (no MFMA, no LDS), ~100x slower than
 * rtm3d_forward, never selected by Model.forward.
 * rtm3d_vtensor: channel slice [coff, coff + c) of a padded NHWC fp32 buffer [B][Hp][Wp][C] with border P (zeros), all
 * device memory owned by the caller.                                                                                    */
typedef struct rtm3d_vtensor {
    float* d;
    int Hp, Wp, C, P, coff;
} rtm3d_vtensor;

/* One group of rtm3d_conv_desc (same iteration domain, tap and output-pixel semantics) in fp32.
 * d_w: fp32 [ntaps][cin][cout]; d_bias: fp32 [cout] (BN folded).  cin, in.C and in.coff must be multiples of 4.
 * out_nchw_f32 != 0: out.d is an fp32 NCHW (B, cout, out_H, out_W) buffer (the logits) and out's geometry is ignored.
 * res.d == NULL: no residual.                                                                                           */
typedef struct rtm3d_vconv_desc {
    rtm3d_vtensor in, out, res;
    const float* d_w;
    const float* d_bias;
    int B, Hm, Wm, in_stride, out_scale, out_oy, out_ox;
    int cin, cout, ntaps, relu;
    int out_nchw_f32, out_H, out_W;
    int tap_dy[RTM3D_MAX_TAPS], tap_dx[RTM3D_MAX_TAPS];
} rtm3d_vconv_desc;
int rtm3d_verify_conv_f32(void* stream, const rtm3d_vconv_desc* desc);

/* rtm3d_op_maxpool in fp32 (same zero-border = -inf convention: pooled maps are post-ReLU).                             */
int rtm3d_verify_maxpool_f32(void* stream, const rtm3d_vtensor* in, const rtm3d_vtensor* out, int B, int Ho, int Wo,
                             int channels, int ksize, int stride, int pad);

/* rtm3d_op_softmax_fuse in fp32: z_out = z_in + sum_i u_i * softmax_{H*W}(u_i), operands added in the given order
 * (models/nets/keypoint_fpn_fusion.py:60-69); exp in fp32, the sums in fp64.  u: array of n_u <= 3 tensors;
 * d_workspace: rtm3d_verify_softmax_workspace_bytes(B, C, n_u) bytes.                                                   */
size_t rtm3d_verify_softmax_workspace_bytes(int B, int C, int n_u);
int rtm3d_verify_softmax_fuse_f32(void* stream, const rtm3d_vtensor* z_in, const rtm3d_vtensor* z_out, int n_u,
                                  const rtm3d_vtensor* u, int B, int H, int W, int C, void* d_workspace);

#ifdef __cplusplus
}
#endif
#endif /* RTM3D_HIP_H */
