// Implicit-GEMM convolution, large-tile variant for the FLOP-heavy layers (head 3x3 convs, the
// 256->256 transposed-conv phases): 256 pixels x 256 output channels per workgroup, 8 waves,
// v_mfma_f32_16x16x32_f16, two K-tile LDS buffers (2 x 64 KB) that are re-staged half-tile by
// half-tile while the other buffer is being multiplied.
//
// Same math, operand roles, LDS swizzle and epilogue as conv_mfma.hip; what differs is the schedule:
//   * each 64-deep K-tile = four half-tiles of 16 KB: XA/XB = pixel rows 0-127/128-255,
//     WA/WB = channel rows 0-127/128-255.  A wave owns 64 pixels of XA + 64 of XB and 32 channels
//     of WA + 32 of WB, so each of its four output quadrants needs exactly one X half and one W half;
//   * a K-tile is four phases  Q(A,A) Q(A,B) Q(B,B) Q(B,A)  of 16 MFMAs; every phase also
//     issues the global_load_lds of ONE half-tile that lies 1-2 K-tiles ahead
//       P1: WB(t+1)  P2: XB(t+1)  P3: XA(t+2)  P4: WA(t+2)
//     (each target half was last read >= 2 phases earlier), then `s_waitcnt vmcnt(8)`:
//     four half-tiles stay in flight across the barriers, the oldest is retired one phase before
//     its first ds_read;
//   * raw s_barrier twice per phase (after the load segment, after the MFMA segment); waves 4-7
//     run one barrier behind waves 0-3, so on every SIMD one wave issues MFMAs while its partner
//     issues LDS reads / DMA / waits (the two co-resident waves of a SIMD share one matrix pipe).
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

#define HALF_ELEMS (128 * 64)            // one half-tile: 128 rows x 64 halves = 16 KB
#define BUF_ELEMS (4 * HALF_ELEMS)       // XA XB WA WB
#define SLOT_XA 0
#define SLOT_XB 1
#define SLOT_WA 2
#define SLOT_WB 3

template <int EPI>
__global__ __launch_bounds__(512) void conv_mfma256_kernel(const ConvKArgs a) {
    __shared__ __attribute__((aligned(16))) f16 lds[2 * BUF_ELEMS + HALF_ELEMS];   // + dummy slot for tail stages
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wc = wave >> 1;         // 2 pixel groups x 4 channel groups

    const int bid = blockIdx.x;
    // XCD-aware order: workgroups b, b+8, b+16.. share an XCD (L2).  Each XCD walks a CONTIGUOUS run of
    // pixel tiles (neighbouring tiles share halo rows) and, per pixel tile, all NT channel tiles.
    const int xcd = bid & 7, jb = bid >> 3;
    const int chunk = (a.MT + 7) >> 3;
    const int ntile = jb % a.NT;
    const int mtile = xcd * chunk + jb / a.NT;
    if (mtile >= a.MT) return;
    const ConvGroupArgs& g = a.g[blockIdx.y];
    const int T = a.ksteps;

    // source offsets of the pixel rows this thread stages: [half][i] -> tile row half*128 + i*64 + tid/8
    uint32_t xoff[2][2];
    {
        const int rr = tid >> 3, cs = tid & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int m = mtile * 256 + h * 128 + i * 64 + rr;
                m = m < a.M ? m : a.M - 1;
                const int n = m / a.HmWm, rem = m - n * a.HmWm;
                const int y = rem / a.Wm, x = rem - y * a.Wm;
                const uint32_t pix = (uint32_t)((n * a.in_Hp + y * a.in_stride + a.in_P) * a.in_Wp + x * a.in_stride + a.in_P);
                xoff[h][i] = pix * (uint32_t)a.in_C + (uint32_t)g.in_coff + (uint32_t)((cs ^ (rr & 7)) * 8);
            }
    }
    const f16* wbase = a.wgt + g.w_off + (size_t)ntile * T * (256 * 64);
    f16* const dummy = lds + 2 * BUF_ELEMS;

    // stage half-tile `slot` of K-tile kt (clamped; tiles past the end go to the dummy slot so the
    // number of outstanding DMA instructions per phase stays constant)
    auto stage = [&](int slot, int kt) {
        const bool live = kt < T;
        const int k = live ? kt : T - 1;
        f16* dst = live ? (lds + (k & 1) * BUF_ELEMS + slot * HALF_ELEMS) : dummy;
        if (slot < 2) {
            const int tap = k / a.cpt, q = k - tap * a.cpt;
            const int koff = g.tap_off[tap] + q * 64;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f16* src = a.in + (size_t)xoff[slot][i] + (ptrdiff_t)koff;
                __builtin_amdgcn_global_load_lds((const GLB_AS void*)src, (LDS_AS void*)(dst + (i * 512 + wave * 64) * 8), 16, 0, 0);
            }
        } else {
            const f16* ws = wbase + (size_t)k * (256 * 64) + (slot - 2) * HALF_ELEMS;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const GLB_AS void*)(ws + (i * 512 + tid) * 8),
                                                 (LDS_AS void*)(dst + (i * 512 + wave * 64) * 8), 16, 0, 0);
        }
    };

    f32x4 acc[2][2][2][4];     // [X half][W half][channel tile][pixel tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[i][j][c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = lane >> 4;
    const int sw0 = ((0 * 4 + fk) ^ (frow & 7)) * 8, sw1 = ((1 * 4 + fk) ^ (frow & 7)) * 8;
    const int xrow = (wp * 64 + frow) * 64;      // + p*16*64 within an X half
    const int wrow = (wc * 32 + frow) * 64;      // + c*16*64 within a W half

    // ---- prologue: tile 0 complete, XA(1), WA(1) in flight
    stage(SLOT_XA, 0); stage(SLOT_WA, 0); stage(SLOT_WB, 0); stage(SLOT_XB, 0);
    stage(SLOT_XA, 1); stage(SLOT_WA, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) __builtin_amdgcn_s_barrier();          // waves 4-7 run one barrier behind

    f16x8 xf[4][2], wa[2][2], wb[2][2];

#define LOAD_X(slot_base)                                                                   \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                         \
        xf[p][0] = *(const f16x8*)((slot_base) + xrow + p * 1024 + sw0);                    \
        xf[p][1] = *(const f16x8*)((slot_base) + xrow + p * 1024 + sw1);                    \
    }
#define LOAD_W(dstf, slot_base)                                                             \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                         \
        dstf[c][0] = *(const f16x8*)((slot_base) + wrow + c * 1024 + sw0);                  \
        dstf[c][1] = *(const f16x8*)((slot_base) + wrow + c * 1024 + sw1);                  \
    }
#define SEG_SYNC()                                                                          \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                        \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    __builtin_amdgcn_s_barrier();                                                           \
    __builtin_amdgcn_sched_barrier(0);
#define MMA(i, j, wfrag)                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                          \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                        \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                       \
            _Pragma("unroll") for (int p = 0; p < 4; ++p)                                   \
                acc[i][j][c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag[c][kk], xf[p][kk], acc[i][j][c][p], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    __builtin_amdgcn_s_barrier();                                                           \
    __builtin_amdgcn_sched_barrier(0);

    for (int t = 0; t < T; ++t) {
        const f16* buf = lds + (t & 1) * BUF_ELEMS;
        // P1: Q(A,A)
        LOAD_X(buf + SLOT_XA * HALF_ELEMS)
        LOAD_W(wa, buf + SLOT_WA * HALF_ELEMS)
        stage(SLOT_WB, t + 1);
        SEG_SYNC()
        MMA(0, 0, wa)
        // P2: Q(A,B)
        LOAD_W(wb, buf + SLOT_WB * HALF_ELEMS)
        stage(SLOT_XB, t + 1);
        SEG_SYNC()
        MMA(0, 1, wb)
        // P3: Q(B,B)
        LOAD_X(buf + SLOT_XB * HALF_ELEMS)
        stage(SLOT_XA, t + 2);
        SEG_SYNC()
        MMA(1, 1, wb)
        // P4: Q(B,A)
        stage(SLOT_WA, t + 2);
        SEG_SYNC()
        MMA(1, 0, wa)
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();           // pair the extra barrier of waves 4-7

    // ---------------------------------------------------------------- epilogue
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = mtile * 256 + i * 128 + wp * 64 + p * 16 + frow;
            if (m >= a.M) continue;
            const int n = m / a.HmWm, rem = m - n * a.HmWm;
            const int y = rem / a.Wm, x = rem - y * a.Wm;
            const int oy = y * a.out_scale + g.out_oy, ox = x * a.out_scale + g.out_ox;
            const size_t opix = ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + g.out_coff;
            const size_t rpix = a.res ? ((size_t)(n * a.res_Hp + oy + a.res_P) * a.res_Wp + ox + a.res_P) * a.res_C + g.res_coff : 0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int c0 = ntile * 256 + j * 128 + wc * 32 + c * 16 + fk * 4;
                    f32x4 v = acc[i][j][c][p];
                    const f32x4 b = *(const f32x4*)(a.bias + g.bias_off + c0);
                    v += b;
                    if (a.res) {
                        const f16x4 r = *(const f16x4*)(a.res + rpix + c0);
                        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
                    }
                    if (a.relu) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                    }
                    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    *(f16x4*)((f16*)a.out + opix + c0) = h;
                }
        }
}

hipError_t launch_conv_mfma256(const ConvKArgs& a, int groups, hipStream_t s) {
    const int mt8 = (a.MT + 7) / 8 * 8;
    dim3 grid(mt8 * a.NT, groups, 1), block(512, 1, 1);
    hipLaunchKernelGGL((conv_mfma256_kernel<0>), grid, block, 0, s, a);
    return hipGetLastError();
}
