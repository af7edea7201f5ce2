"""Guards on the compiled device code that need no GPU: the ISA of every source in rtm3d_amd/csrc,
compiled with the library's own flags (`make isa`), is checked for

  * no VGPR spills and no private (scratch) memory in any product kernel: a register-budget regression shows up here
    instead of as a slow kernel on the GPU box;
  * who touches `m0`: the LDS-DMA of the pipelined conv kernels is issued from inline asm (`s_mov_b32 m0, <lds base>` +
    `global_load_lds_dwordx4` in ONE asm statement, so the value never has to survive outside it).  `m0` is a reserved
    register the compiler does not promise to preserve around such a statement; the idiom is sound as long as the
    compiler's OWN code in those kernels never uses m0.  So: every textual mention of m0 must lie inside an inline-asm
    block, must be the `s_mov_b32 m0, sN` of the DMA macro, and must be followed by the LDS-DMA load within the block.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'rtm3d_amd', 'csrc')
ISA = os.path.join(ROOT, 'rtm3d_amd', '_C', 'obj', 'isa')

# kernels that are allowed private memory: none of the product path.  Exempt: the scalar one-thread-per-object L-BFGS-B
# kernel (lbfgsb.h in thread-private arrays) behind rtm3d_decode3d_scalar / rtm3d_decode3d_reference_form, which exists
# only as the bit-exact cross-check of the wave-cooperative product kernel (decode3d_wave_kernel).
SCRATCH_EXEMPT_PREFIXES = ('_Z15decode3d_kernelILi',)


@pytest.fixture(scope='module')
def isa_files():
    import shutil
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        # a FAILURE, not a skip: this guard is what stands in for the m0 clobber the compiler does not honour (common.h, RT_DMA16);
        # a box that cannot run it must not report the suite green
        pytest.fail('no hipcc (%s): the ISA guard needs the compiler (set HIPCC)' % hipcc)
    r = subprocess.run(['make', '-C', CSRC, '-j8', 'isa'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        pytest.fail('`make isa` failed (rc %d):\n%s' % (r.returncode, '\n'.join(r.stdout.splitlines()[-60:])))
    # the build is warning-free since round 5 (-Wno-inline-asm for the LDS-DMA translation units only, Makefile DMA_TUS): any
    # diagnostic is news
    warns = [ln for ln in r.stdout.splitlines() if 'warning:' in ln]
    assert not warns, 'compiler warnings in `make isa`:\n%s' % '\n'.join(warns[:20])
    files = sorted(f for f in os.listdir(ISA) if f.endswith('.s'))
    srcs = sorted(f[:-4] + '.s' for f in os.listdir(CSRC) if f.endswith('.hip'))
    assert files == srcs, (files, srcs)
    return {f: open(os.path.join(ISA, f)).read() for f in files}


def kernels_metadata(text):
    """[(name, {field: int})] from the amdhsa.kernels metadata of one .s file."""
    out = []
    meta = text[text.find('amdhsa.kernels:'):]
    for blk in re.split(r'\n  - ', meta)[1:]:
        name = re.search(r'\.name:\s+(\S+)', blk)
        if not name:
            continue
        f = {k: int(v) for k, v in re.findall(r'\.(vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|vgpr_count|'
                                              r'agpr_count|group_segment_fixed_size):\s+(\d+)', blk)}
        out.append((name.group(1), f))
    return out


def test_no_spills_no_scratch(isa_files):
    seen = 0
    for fname, text in isa_files.items():
        for name, f in kernels_metadata(text):
            seen += 1
            if name.startswith(SCRATCH_EXEMPT_PREFIXES):
                continue
            assert f['vgpr_spill_count'] == 0, '%s: %s spills %d VGPRs' % (fname, name, f['vgpr_spill_count'])
            assert f['private_segment_fixed_size'] == 0, '%s: %s uses %d B of scratch' % (fname, name, f['private_segment_fixed_size'])
            assert f['group_segment_fixed_size'] <= 160 * 1024
    assert seen >= 40, seen           # the library has ~60 kernel instantiations; an empty parse must not pass


def test_m0_only_inside_the_dma_macro(isa_files):
    """Per kernel function: if inline asm writes m0 (the DMA macro), then NOTHING else in that function may mention m0 -
    neither compiler-generated code (e.g. the m0 the compiler sets up for __builtin_amdgcn_global_load_lds, s_movrel,
    v_readlane ..., m0) nor another asm form.  Kernels that leave m0 to the compiler (the generic conv_mfma_kernel: the builtin) are free."""
    asm_kernels = 0
    for fname, text in isa_files.items():
        body = text.split('.amdgpu_metadata')[0]
        # function bodies: "<symbol>:   ; @<symbol>" ... ".Lfunc_end"
        for m in re.finditer(r'^(\S+):\s*; @\1\n(.*?)^\.Lfunc_end', body, re.S | re.M):
            func, code_txt = m.group(1), m.group(2)
            in_asm, pending = False, False
            asm_movs, outside = 0, []
            for ln in code_txt.splitlines():
                s = ln.strip()
                if s.startswith(';;#ASMSTART'):
                    in_asm, pending = True, False
                    continue
                if s.startswith(';;#ASMEND'):
                    assert not pending, '%s %s: s_mov_b32 m0 without its LDS-DMA load in the same asm block' % (fname, func)
                    in_asm = False
                    continue
                code = s.split(';')[0]
                if in_asm and pending and re.match(r'global_load_lds_', code):
                    pending = False
                if re.search(r'\bm0\b', code):
                    if in_asm:
                        assert re.match(r's_mov_b32\s+m0,\s*(s\d+|vcc_lo|vcc_hi|ttmp\d+)$', code), '%s %s: unexpected use of m0 in inline asm: %r' % (fname, func, s)
                        pending = True
                        asm_movs += 1
                    else:
                        outside.append(s)
            if asm_movs:
                asm_kernels += 1
                assert not outside, '%s %s: inline asm clobbers m0 AND compiler-generated code uses it: %r' % (fname, func, outside[:3])
    assert asm_kernels >= 8, asm_kernels       # conv_mfma (deep), conv_mfma256 (+halo), conv128 / conv64 halo, conv32s2


# SGPR spills of the persistent 256-pixel kernels: the compiler parks scalars in VGPR lanes (v_writelane /
# v_readlane).  That is harmless per TILE (a few dozen lane reads among ~2000 MFMA cycles) and expensive inside the steady-state K
# loop, where every vector instruction between MFMAs costs issue slots of the binding pipe.  Caps = what the committed code has
# (profiles/r04_sgpr_spills.txt); the inner loop must stay free of lane traffic.
# (round 5: the halo kernel carries the row-slice offsets and the next K-tile's fragment addresses: 60 / 53 scalars parked, none in the K loop)
SGPR_SPILL_CAPS = {'conv_mfma256_persistent_kernel': 32, 'conv_mfma256_halo_kernel': 64}


def test_persistent_kernels_keep_sgpr_spills_out_of_the_k_loop(isa_files):
    checked = 0
    for fname in ('conv_mfma256.s', 'conv_mfma256_halo.s'):
        text = isa_files[fname]
        meta = dict(kernels_metadata(text))
        body = text.split('.amdgpu_metadata')[0]
        for m in re.finditer(r'^(\S+):\s*; @\1\n(.*?)^\.Lfunc_end', body, re.S | re.M):
            func, code = m.group(1), m.group(2)
            cap = next((c for k, c in SGPR_SPILL_CAPS.items() if k in func), None)
            if cap is None:
                continue
            assert meta[func]['sgpr_spill_count'] <= cap, '%s spills %d SGPRs (cap %d)' % (func, meta[func]['sgpr_spill_count'], cap)
            lines = code.splitlines()
            # innermost loops: a label whose header comment says "Inner Loop Header: Depth=2"; body = label .. last branch back to it
            for i, ln in enumerate(lines):
                lab = re.match(r'^(\.LBB\d+_\d+):', ln)
                if not lab or 'Depth=2' not in ' '.join(lines[i:i + 3]) or 'Inner Loop Header' not in ' '.join(lines[i:i + 3]):
                    continue
                back = [j for j in range(i + 1, len(lines)) if re.match(r'\s*s_c?branch\S*\s+%s\b' % re.escape(lab.group(1)), lines[j])]
                assert back, (func, lab.group(1))
                inner = [l.strip() for l in lines[i:back[-1] + 1]]
                n_mfma = sum(l.startswith('v_mfma') for l in inner)
                lane = [l for l in inner if l.startswith(('v_readlane', 'v_writelane'))]
                assert n_mfma >= 64, (func, lab.group(1), n_mfma)          # it IS the K loop (four phases of 16-32 MFMAs)
                assert not lane, '%s: SGPR spill traffic inside the K loop %s: %r' % (func, lab.group(1), lane[:4])
                checked += 1
    assert checked >= 4, checked        # three persistent instantiations + two halo ones
