// Shared declarations between the runtime (runtime.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// m / d for 0 <= m < 2^31 and a quotient below 2^21 (image or row index): float estimate, +-1 fix-up.
// Eight VALU instructions instead of the ~30 of the generic 32-bit division sequence.
__device__ __forceinline__ int div_small_q(int m, int d, float rcp_d) {
    const int q = (int)((float)m * rcp_d);
    const int r = m - q * d;
    return q + (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
}


// ------------------------------------------------------------------------------------------------
// LDS-DMA issued from inline asm: the ONE definition every pipelined conv kernel uses (m0 = LDS byte address of the wave's
// 1 KB run, one 16-byte piece per lane).  Why asm: the compiler must not know these write LDS - its waitcnt pass treats every
// visible ds_read as possibly aliasing a pending LDS-DMA and drains vmcnt(0) in front of it; ordering against the DMA is each
// kernel's counted `s_waitcnt vmcnt(N)` + barrier.
// m0 is a RESERVED register: the compiler does not promise to honour it in a clobber list (it says so: -Winline-asm "clobber
// list contains reserved registers", silenced for exactly the translation units that use these macros: Makefile, DMA_TUS).
// The idiom is sound because m0 is written and consumed inside ONE asm statement, so its value never has to survive outside
// it, AND because the compiler's own code in those kernels never touches m0: tests/test_isa_guard.py::
// test_m0_only_inside_the_dma_macro reads the generated ISA of every kernel and FAILS if a kernel that uses these macros
// contains any other mention of m0 (compiler-emitted or hand-written).  A toolchain upgrade that changes this shows up there.
#define RT_DMA16(gptr, lds_byte_addr) \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_byte_addr) : "memory", "m0")
// non-temporal form (single-read streams)
#define RT_DMA16_NT(gptr, lds_byte_addr) \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(gptr), "s"(lds_byte_addr) : "memory", "m0")
// wave-uniform 64-bit base in SGPRs + a 32-bit byte offset per lane (no 64-bit address registers).  The base goes through
// rt_uniform_ptr: where the compiler already holds it in SGPRs that is no instruction at all; where its divergence analysis lost
// track (seen in the -DC256_STAMPS build) it becomes two v_readfirstlane instead of a VGPR pair in an "s" operand, which the assembler
// rejects.
static __device__ __forceinline__ const void* rt_uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
#define RT_DMA16_SBASE(voff_bytes, sbase, lds_byte_addr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff_bytes), "s"(rt_uniform_ptr(sbase)), "s"(lds_byte_addr) : "memory", "m0")

// the same with the wave's active lanes cut down to `lanemask` (a 64-bit SGPR value) for this one instruction: no branch around it
#define RT_DMA16_SBASE_LANES(voff_bytes, sbase, lds_byte_addr, lanemask) \
    { unsigned long long rt_exec_; \
      asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0" \
                   : "=&s"(rt_exec_) : "v"(voff_bytes), "s"(rt_uniform_ptr(sbase)), "s"(lds_byte_addr), "s"(lanemask) : "memory", "m0", "scc"); }
#define RT_DMA16_SBASE_NT(voff_bytes, sbase, lds_byte_addr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(voff_bytes), "s"(rt_uniform_ptr(sbase)), "s"(lds_byte_addr) : "memory", "m0")

#define RT_MAX_GROUPS 4
#define RT_MAX_TAPS 80

// Geometry of one padded NHWC fp16 activation tensor as the kernels see it.
struct TensorView {
    f16* base;          // element [n=0][yp=0][xp=0][c=0] of the padded buffer
    int Hp, Wp, C, P;   // padded height/width, channel pitch, border width
};

struct ConvGroupArgs {
    int in_coff, out_coff, res_coff;
    int out_oy, out_ox;
    uint32_t w_off;          // element offset of this group's packed weights
    int bias_off;            // element offset into bias
    int tap_off[RT_MAX_TAPS];  // (dy*in_Wp + dx)*in_C + tap_dc, elements (may be negative)
};

struct ConvKArgs {
    const f16* in;
    const f16* wgt;
    const float* bias;
    const f16* res;
    void* out;
    int M, HmWm, Wm;
    int in_Hp, in_Wp, in_C, in_stride, in_P;
    int out_Hp, out_Wp, out_C, out_scale, out_P;
    int res_Hp, res_Wp, res_C, res_P;
    int cin, cout, ksteps, cpt;   // cpt = 64-channel chunks per tap (MFMA path)
    int ntaps;
    int relu, MT, NT;
    int out_H, out_W;             // NCHW fp32 epilogue only
    float* slab;                  // split-K (conv_mfma_deep_kernel): fp32 partial tiles [split][group][tile][128 x BN], else null
    int ksplit;                   // number of K ranges (grid.z); 0 / 1 = off
    int deep;                     // 1: small launch, runs on conv_mfma_deep_kernel (4-slot LDS ring, optional split-K)
    // second copy of the output in SPACE-TO-DEPTH layout (conv_mfma.hip epilogue only), or null: pixel (y, x) -> half-resolution
    // pixel (y >> 1, x >> 1), channels s_coff + ((y & 1) * 2 + (x & 1)) * cout + c
    f16* s2d;
    int s_Hp, s_Wp, s_C, s_P, s_coff;
    int in_s2d;                   // conv64s2_halo.hip: `in` is the space-to-depth copy (half resolution, 4 x 64 channels at in_coff) of the 64-channel input map
    ConvGroupArgs g[RT_MAX_GROUPS];
};

struct StemKArgs {
    const float* in;      // fp32 NCHW (B,3,H,W)
    const float* wgt;     // fp32 [ky][kx][ci][co]
    const float* bias;    // fp32 [co]
    f16* out;
    int B, H, W, Ho, Wo, stride, pad;
    int out_Hp, out_Wp, out_C, out_P;
};

struct PoolKArgs {
    const f16* in; f16* out;
    int B, Ho, Wo, C8;            // C8 = channels/8
    int in_Hp, in_Wp, in_C, in_P, in_coff;
    int out_Hp, out_Wp, out_C, out_P, out_coff;
    int ksize, stride, pad;
};

struct SoftmaxKArgs {
    const f16* z_in; f16* z_out;
    const f16* u[3];
    int n_u;
    int B, H, W, C;                // C == 256
    int z_Hp, z_Wp, z_C, z_P;      // z_out geometry
    int zi_Hp, zi_Wp, zi_C, zi_P;  // z_in geometry
    int u_Hp[3], u_Wp[3], u_C[3], u_P[3];
    float* partial;                // [n_u][B][chunks][C][2] (max, sumexp)
    float* stats;                  // [n_u][B][C][2] (max, 1/sum)
    int chunks, rows_per_chunk;    // reduce pass: one workgroup per (u, image, chunk of rows)
    int apply_chunks, apply_rows, xsplit, seg_w;   // apply pass: one workgroup per (image, chunk of rows, column segment)
    int partial_chunks;            // > 0: `partial` was written by the producers' epilogues with this many chunks per image
};

// Workgroup b of an n-workgroup launch runs on XCD b % 8 (round-robin dispatch).  Index of the work item it should take so that
// every XCD walks a CONTIGUOUS eighth of the items (a bijection on [0, n) for any n): items that share input land in one L2.
__device__ __forceinline__ int xcd_contiguous_index(int b, int n) {
#ifdef RT_TIMING_NO_XCD_ORDER
    return b;
#endif
    const int q = n >> 3, r = n & 7, xcd = b & 7, k = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + k;
}

struct HeadOutArgs {
    const f16* in;          // h2: padded NHWC, 4 x 256 channels
    const f16* wgt;         // [head][tap][chunk*2+kk][lane][8]  (16 output rows, zero padded)
    const float* bias;      // [head][16]
    float* out[4];          // fp32 NCHW logits per head
    int cout[4];
    int nheads;
    int B, H, W;
    int in_Hp, in_Wp, in_C, in_P;
    int tiles_x, tiles_y, tile_rows;   // tile_rows: 8 or 16
};

struct StemFusedArgs {
    const f16* x4;          // NHWC4 fp16 image tensor (padded, border >= 4)
    const float* x_nchw;    // or (per launch) the caller's fp32 NCHW batch: read directly, x4 unused
    f16* out;               // level0 output tensor (two-layer form) / level1 output tensor, half resolution (three-layer form)
    const f16* w_base;      // [7 k-steps][64 lanes][8]
    const f16* w_l0;        // [5 k-steps][64 lanes][8]
    const f16* w_l1;        // [2 channel tiles][5 k-steps][64 lanes][8], or null: two-layer form
    const float* b_base;    // [16]
    const float* b_l0;      // [16]
    const float* b_l1;      // [32]
    int B, H, W;
    int x_Hp, x_Wp, x_P;
    int o_Hp, o_Wp, o_C, o_P, o_coff;
    int tiles_x, tiles_y;
};

struct Conv32S2Args {
    const f16* in;          // 32-channel slice of a padded NHWC tensor (border >= 1), full resolution of this op
    f16* out_conv;          // ReLU(BN(conv3x3 stride 2)), 64 channels, half resolution
    f16* out_proj;          // BN(conv1x1(maxpool2x2)), 64 channels, half resolution
    const f16* w_conv;      // [9 taps][4 channel tiles][64 lanes][8]
    const f16* w_proj;      // [4 channel tiles][64 lanes][8]
    const float* b_conv;    // [64]
    const float* b_proj;    // [64]
    int B, Ho, Wo;
    int in_Hp, in_Wp, in_C, in_P, in_coff;
    int oc_Hp, oc_Wp, oc_C, oc_P, oc_coff;
    int op_Hp, op_Wp, op_C, op_P, op_coff;
};

struct RootKArgs {
    const f16* w;           // [4 output tiles][4 K-steps][64 lanes][8] MFMA A fragments of the 128 -> 64 root, K order permuted (conv64_root.hip)
    const float* bias;      // [64]
    f16* out;               // root output slice: 64 channels at o_coff
    int o_Hp, o_Wp, o_C, o_P, o_coff;
    f16* pool;              // 2x2 / stride 2 max-pool of the root output (64 channels at p_coff, half resolution), or null
    int p_Hp, p_Wp, p_C, p_P, p_coff;
    f16* s2d;               // second copy of the root output in SPACE-TO-DEPTH layout, or null: pixel (y, x) -> half-resolution pixel
    int s_Hp, s_Wp, s_C, s_P, s_coff;      // (y >> 1, x >> 1), channels s_coff + ((y & 1) * 2 + (x & 1)) * 64 + c
    int relu;
};

struct PatchMaskArgs {
    f16* base;              // [n_slots][S][S][C] patch tensor (no border)
    const int32_t* yx;      // [n_slots][2] peak (y, x) in map pixels, -1 = empty slot (written by rtm3d_gather_peak_patches)
    int n_slots, S, C, img_H, img_W, origin;
};

// kernel launchers (each returns hipGetLastError())
hipError_t launch_patch_mask(const PatchMaskArgs& a, hipStream_t s);
hipError_t launch_conv32s2_fused(const Conv32S2Args& a, int cu_count, unsigned int* ticket_ctr, hipStream_t s);
hipError_t launch_stem_fused(const StemFusedArgs& a, hipStream_t s);
hipError_t launch_conv_headout(const HeadOutArgs& a, hipStream_t s);
hipError_t launch_conv_mfma(const ConvKArgs& a, int bn_tile, int groups, int epi_nchw, hipStream_t s);
hipError_t launch_conv_mfma_deep(const ConvKArgs& a, int bn_tile, int groups, unsigned int* tile_ctr, hipStream_t s);
hipError_t launch_conv_mfma256(const ConvKArgs& a, int groups, unsigned int* tile_ctr, float* stat_out, hipStream_t s);
bool conv_mfma256_uses_halo(const ConvKArgs& a, int groups);
bool conv_mfma256_uses_lattice(const ConvKArgs& a, int groups);
bool conv64_halo_supported(const ConvKArgs& a, int groups);
hipError_t launch_conv64_halo(const ConvKArgs& a, int cu_count, unsigned int* ticket_ctr, hipStream_t s);
hipError_t launch_conv64_root(const ConvKArgs& a, const RootKArgs& r, int cu_count, unsigned int* ticket_ctr, hipStream_t s);
bool conv64s2_halo_supported(const ConvKArgs& a, int groups);
hipError_t launch_conv64s2_halo(const ConvKArgs& a, int cu_count, unsigned int* ticket_ctr, hipStream_t s);
bool conv128_halo_supported(const ConvKArgs& a, int groups);
hipError_t launch_conv128_halo(const ConvKArgs& a, int cu_count, unsigned int* ticket_ctr, hipStream_t s);
hipError_t launch_conv_smallc(const ConvKArgs& a, hipStream_t s);
bool conv_smallc_supported(int cin, int cout, int ntaps);
hipError_t launch_nchw_to_nhwc4(const float* in, f16* out, int B, int H, int W, int Hp, int Wp, int P, hipStream_t s);
hipError_t launch_maxpool(const PoolKArgs& a, hipStream_t s);
hipError_t launch_softmax_fuse(const SoftmaxKArgs& a, hipStream_t s);
