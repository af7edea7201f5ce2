// Implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_16x16x32_f16).
//
// GEMM view:  D[cout][pixel] = sum_k  W[cout][k] * X[pixel][k],   k = (tap, 64-channel chunk)
//   - activations are padded NHWC fp16, so a tap is a constant element offset from the pixel's
//     base address and the zero border implements the convolution padding: no bounds checks;
//   - one k-step stages a BM x 64 pixel tile and a BN x 64 weight tile (128-byte rows) into LDS
//     with global_load_lds_dwordx4 (16 B/lane, direct to LDS).  The LDS image is lane-linear, so
//     the bank swizzle (16-B chunk ^= row & 7) is applied on the per-lane SOURCE address for
//     pixels and baked into the host-side weight packing; reads apply the same XOR -> conflict
//     free ds_read_b128 for the 16x16x32 operand map (lane -> row = lane & 15, k-chunk = lane >> 4);
//   - weights are the MFMA "A" operand, pixels the "B" operand, so each lane ends up with four
//     consecutive output channels of one pixel: 8-byte NHWC stores, fused bias/residual/ReLU.
//   - blockIdx is remapped so the NT channel tiles of one pixel tile run on one XCD (shared L2).
//
// Replaces the stock ATen conv2d / conv_transpose2d calls of models/nets/{dla,resnet,header,
// keypoint_fpn_fusion,module}.py.
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

// Epilogue shared by the kernels of this file: + bias (+ residual) (ReLU) -> fp16 NHWC.  Loads (bias, residual) first, then one
// run of independent stores: a load between two stores makes the compiler wait vmcnt(0), i.e. for the previous store's
// acknowledgement too.
template <int BM, int BN, int WGM, int WGN, int RES>
__device__ __forceinline__ void store_nhwc_tile(const ConvKArgs& a, const ConvGroupArgs& g, f32x4 (&acc)[BN / WGN / 16][BM / WGM / 16],
                                                const int mtile, const int ntile, const int wp, const int wc, const int frow, const int fk) {
    constexpr int TP = BM / WGM / 16, TC = BN / WGN / 16;
    const float rcp_hw = 1.0f / (float)a.HmWm, rcp_w = 1.0f / (float)a.Wm;
    constexpr bool WIDE = (TC % 2) == 0;       // pair two 16-channel tiles -> 16-byte stores
    const int cw = ntile * BN + wc * (BN / WGN);
    f32x4 bv[TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) bv[c] = *(const f32x4*)(a.bias + g.bias_off + cw + c * 16 + fk * 4);
    const f16 lo = a.relu ? (f16)0.f : (f16)(-__builtin_inff());
    const f16x4 lo4 = {lo, lo, lo, lo};
    size_t opix[TP], spix[TP];
    f16x4 rv[TP][TC];
#pragma unroll
    for (int p = 0; p < TP; ++p) {
        // rows past M were staged from pixel M-1 and hold its result: a same-value write
        int m = mtile * BM + wp * (BM / WGM) + p * 16 + frow;
        m = m < a.M ? m : a.M - 1;
        const int n = div_small_q(m, a.HmWm, rcp_hw), rem = m - n * a.HmWm;
        const int y = div_small_q(rem, a.Wm, rcp_w), x = rem - y * a.Wm;
        const int oy = y * a.out_scale + g.out_oy, ox = x * a.out_scale + g.out_ox;
        opix[p] = ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + g.out_coff + cw;
        spix[p] = 0;
        if (a.s2d)      // space-to-depth copy (the neck reads the feature at the resolution of its transposed conv's input grid)
            spix[p] = ((size_t)(n * a.s_Hp + (oy >> 1) + a.s_P) * a.s_Wp + (ox >> 1) + a.s_P) * a.s_C + a.s_coff + ((oy & 1) * 2 + (ox & 1)) * a.cout + cw;
        if (RES) {
            const f16* rp = a.res + ((size_t)(n * a.res_Hp + oy + a.res_P) * a.res_Wp + ox + a.res_P) * a.res_C + g.res_coff + cw + fk * 4;
#pragma unroll
            for (int c = 0; c < TC; ++c) rv[p][c] = *(const f16x4*)(rp + c * 16);
        }
    }
#pragma unroll
    for (int p = 0; p < TP; ++p) {
        f16x4 h[TC];
#pragma unroll
        for (int c = 0; c < TC; ++c) {
            f32x4 v = acc[c][p] + bv[c];
            if (RES) {
                const f16x4 r = rv[p][c];
                v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
            }
            const f16x4 t = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            h[c] = __builtin_elementwise_max(t, lo4);
        }
        if (WIDE) {
            // v_permlane16_swap: rows (16-lane groups) 1,3 of the even tile's registers <-> rows 0,2 of
            // the odd tile's, after which a lane owns 8 consecutive channels of the 32-channel pair
            const int so = (fk & 1) * 16 + (fk >> 1) * 8;
#pragma unroll
            for (int c = 0; c < TC; c += 2) {
                uint32_t u0[2], u1[2];
                __builtin_memcpy(u0, &h[c], 8);
                __builtin_memcpy(u1, &h[c + 1 < TC ? c + 1 : c], 8);
                const auto s0 = __builtin_amdgcn_permlane16_swap(u0[0], u1[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(u0[1], u1[1], false, false);
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                if (cw + c * 16 + so < a.cout) {
                    if (a.out) *(u32x4*)((f16*)a.out + opix[p] + c * 16 + so) = o;       // (null: only the space-to-depth copy exists)
                    if (a.s2d) *(u32x4*)(a.s2d + spix[p] + c * 16 + so) = o;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < TC; ++c)
                if (cw + c * 16 + fk * 4 < a.cout) {
                    if (a.out) *(f16x4*)((f16*)a.out + opix[p] + c * 16 + fk * 4) = h[c];
                    if (a.s2d) *(f16x4*)(a.s2d + spix[p] + c * 16 + fk * 4) = h[c];
                }
        }
    }
}

template <int BM, int BN, int WGM, int WGN, int EPI, int RES>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvKArgs a) {
    constexpr int XP = BM * 8;                 // 16-byte pieces of the pixel tile per k-step
    constexpr int WP = BN * 8;                 // pieces of the weight tile
    constexpr int XI = XP / 256;
    constexpr int WI = (WP + 255) / 256;
    constexpr int TP = BM / WGM / 16;          // 16-pixel sub-tiles per wave
    constexpr int TC = BN / WGN / 16;          // 16-channel sub-tiles per wave
    constexpr int STAGE = (BM + BN) * 64;      // halves per LDS stage
    __shared__ __attribute__((aligned(16))) f16 lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave / WGN, wc = wave % WGN;

    // XCD-aware tile mapping: blocks b, b+8, b+16.. share an XCD -> give them the same pixel tile.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int chunk = (a.MT + 7) >> 3;           // each XCD walks a contiguous run of pixel tiles (shared halo rows)
    const int ntile = j % a.NT;
    const int mtile = xcd * chunk + j / a.NT;
    if (mtile >= a.MT) return;
    const ConvGroupArgs& g = a.g[blockIdx.y];

    // per-lane source offsets of the pixel rows this lane stages (fixed for the whole K loop)
    uint32_t xoff[XI];
    {
        const int rr = tid >> 3, cs = tid & 7;
        const float rcp_hw0 = 1.0f / (float)a.HmWm, rcp_w0 = 1.0f / (float)a.Wm;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int m = mtile * BM + i * 32 + rr;
            m = m < a.M ? m : a.M - 1;
            const int n = div_small_q(m, a.HmWm, rcp_hw0), rem = m - n * a.HmWm;
            const int y = div_small_q(rem, a.Wm, rcp_w0), x = rem - y * a.Wm;
            const uint32_t pix = (uint32_t)((n * a.in_Hp + y * a.in_stride + a.in_P) * a.in_Wp + x * a.in_stride + a.in_P);
            xoff[i] = pix * (uint32_t)a.in_C + (uint32_t)g.in_coff + (uint32_t)((cs ^ (rr & 7)) * 8);
        }
    }
    const f16* wbase = a.wgt + g.w_off + (size_t)ntile * a.ksteps * (BN * 64);
    // a 1x1 conv with ONE channel tile reads every input byte once: non-temporal DMA (aux = 2) for the pixel operand, so the
    // stream does not push the weights and the neighbours' lines out of L2 / the Infinity Cache (same-box A/B, profiles/r03_ab_nt.txt:
    // level3 roots 0.044 -> 0.041 ms, the neck's 320 -> 256 1x1 0.249 -> 0.226; with several channel tiles or taps the operand is
    // re-read through L2 and nt costs: the logit convs' halo staging 0.512 -> 0.553 ms)
    const bool x_once = a.ntaps == 1 && a.NT == 1 && a.in_stride == 1;

    auto stage = [&](int buf, int ks) {
        const int tap = ks / a.cpt, q = ks - tap * a.cpt;
        const int koff = g.tap_off[tap] + q * 64;
        f16* xl = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const f16* src = a.in + (size_t)xoff[i] + (ptrdiff_t)koff;
            if (x_once) __builtin_amdgcn_global_load_lds((const GLB_AS void*)src, (LDS_AS void*)(xl + (i * 256 + wave * 64) * 8), 16, 0, 2);
            else __builtin_amdgcn_global_load_lds((const GLB_AS void*)src, (LDS_AS void*)(xl + (i * 256 + wave * 64) * 8), 16, 0, 0);
        }
        const f16* ws = wbase + (size_t)ks * (BN * 64);
        f16* wl = xl + BM * 64;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            if (WP >= 256 || wave * 64 < WP) {   // wave-uniform guard for the 16-channel tile
                __builtin_amdgcn_global_load_lds((const GLB_AS void*)(ws + (i * 256 + tid) * 8),
                                                 (LDS_AS void*)(wl + (i * 256 + wave * 64) * 8), 16, 0, 0);
            }
        }
    };

    f32x4 acc[TC][TP];
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
        for (int p = 0; p < TP; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = lane >> 4;
    stage(0, 0);
    __syncthreads();
    for (int ks = 0; ks < a.ksteps; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < a.ksteps) stage(cur ^ 1, ks + 1);
        const f16* xl = lds + cur * STAGE;
        const f16* wl = xl + BM * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = ((kk * 4 + fk) ^ (frow & 7)) * 8;
            f16x8 xf[TP], wf[TC];
#pragma unroll
            for (int p = 0; p < TP; ++p)
                xf[p] = *(const f16x8*)(xl + (wp * (BM / WGM) + p * 16 + frow) * 64 + sw);
#pragma unroll
            for (int c = 0; c < TC; ++c)
                wf[c] = *(const f16x8*)(wl + (wc * (BN / WGN) + c * 16 + frow) * 64 + sw);
#pragma unroll
            for (int c = 0; c < TC; ++c)
#pragma unroll
                for (int p = 0; p < TP; ++p)
                    acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], xf[p], acc[c][p], 0, 0, 0);
        }
        __syncthreads();   // drains the in-flight global_load_lds (vmcnt(0)) and orders LDS reuse
    }

    // ---------------------------------------------------------------- epilogue
    const float rcp_hw = 1.0f / (float)a.HmWm, rcp_w = 1.0f / (float)a.Wm;
    if (EPI == 0) {
        store_nhwc_tile<BM, BN, WGM, WGN, RES>(a, g, acc, mtile, ntile, wp, wc, frow, fk);
    } else {
#pragma unroll
        for (int p = 0; p < TP; ++p) {
            const int m = mtile * BM + wp * (BM / WGM) + p * 16 + frow;
            if (m >= a.M) continue;
            const int n = div_small_q(m, a.HmWm, rcp_hw), rem = m - n * a.HmWm;
            const int y = div_small_q(rem, a.Wm, rcp_w), x = rem - y * a.Wm;
            const int oy = y * a.out_scale + g.out_oy, ox = x * a.out_scale + g.out_ox;
#pragma unroll
            for (int c = 0; c < TC; ++c) {
                const int c0 = ntile * BN + wc * (BN / WGN) + c * 16 + fk * 4;
                if (c0 >= a.cout) continue;
                const f32x4 v = acc[c][p];
                const float* bp = a.bias + g.bias_off + c0;
                float* o = (float*)a.out;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cc = c0 + e;
                    if (cc < a.cout) {
                        float r = v[e] + bp[e];
                        if (a.relu) r = fmaxf(r, 0.f);
                        o[((size_t)(n * a.out_C + g.out_coff + cc) * a.out_H + oy) * a.out_W + ox] = r;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small launches (no more workgroups than CUs: small batches, the deep levels).  One workgroup per CU cannot hide the
// global->LDS latency behind other workgroups, so the K loop of the kernel above (one stage in flight, drained every
// k-step) pays a full memory round trip per k-step: ~1 us x up to 144 k-steps, 12-70 us for layers that hold ~2 us of
// MFMA work.  This variant
//   - keeps DEEP_NS - 1 = 3 stages in flight in a 4-slot LDS ring (LDS is free: nobody else is on the CU), waiting with
//     counted s_waitcnt vmcnt so only the oldest stage must have landed (the DMA is issued from inline asm: the
//     compiler would otherwise drain vmcnt(0) before every LDS read);
//   - splits K over blockIdx.z (a.ksplit ranges) when the launch would leave most CUs idle.  Partial tiles go to an fp32
//     slab; the LAST workgroup of a tile to arrive (device-scope fence + one atomic ticket per tile, self-resetting)
//     sums the partials IN SPLIT ORDER - the result does not depend on which workgroup arrives last - and runs the
//     usual epilogue.  No second launch.
// Same operand maps, weight packing and epilogue as conv_mfma_kernel: bit-identical results when ksplit == 1.
#define DEEP_NS 4
#define DEEP_DMA16 RT_DMA16                         // common.h: the one LDS-DMA definition

template <int BN, int RES>
__global__ __launch_bounds__(256) void conv_mfma_deep_kernel(const ConvKArgs a, unsigned int* tile_ctr) {
    constexpr int BM = 128, WGM = 2, WGN = 2;
    constexpr int XI = BM * 8 / 256;           // 4 pixel-tile pieces per lane and stage
    constexpr int WI = BN * 8 / 256;           // 2 / 4 weight-tile pieces
    constexpr int LPS = XI + WI;               // DMA instructions per wave and stage
    constexpr int TP = BM / WGM / 16, TC = BN / WGN / 16;
    constexpr int STAGE = (BM + BN) * 64;      // halves per LDS stage
    extern __shared__ __attribute__((aligned(128))) f16 lds[];
    __shared__ int arrived;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave / WGN, wc = wave % WGN;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int chunk = (a.MT + 7) >> 3;
    const int ntile = j % a.NT;
    const int mtile = xcd * chunk + j / a.NT;
    if (mtile >= a.MT) return;
    const ConvGroupArgs& g = a.g[blockIdx.y];

    uint32_t xoff[XI];
    {
        const int rr = tid >> 3, cs = tid & 7;
        const float rcp_hw0 = 1.0f / (float)a.HmWm, rcp_w0 = 1.0f / (float)a.Wm;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int m = mtile * BM + i * 32 + rr;
            m = m < a.M ? m : a.M - 1;
            const int n = div_small_q(m, a.HmWm, rcp_hw0), rem = m - n * a.HmWm;
            const int y = div_small_q(rem, a.Wm, rcp_w0), x = rem - y * a.Wm;
            const uint32_t pix = (uint32_t)((n * a.in_Hp + y * a.in_stride + a.in_P) * a.in_Wp + x * a.in_stride + a.in_P);
            xoff[i] = pix * (uint32_t)a.in_C + (uint32_t)g.in_coff + (uint32_t)((cs ^ (rr & 7)) * 8);
        }
    }
    const int ks0 = (int)(((long long)blockIdx.z * a.ksteps) / a.ksplit);
    const int ks1 = (int)(((long long)(blockIdx.z + 1) * a.ksteps) / a.ksplit);
    const int nk = ks1 - ks0;
    const f16* wnext = a.wgt + g.w_off + ((size_t)ntile * a.ksteps + ks0) * (BN * 64) + tid * 8;   // this lane's piece of the next stage
    int tap = ks0 / a.cpt, q = ks0 - tap * a.cpt;                                                   // (tap, chunk) of the next stage
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;

    auto stage = [&](int slot) {
        const int koff = g.tap_off[tap] + q * 64;
        const uint32_t xl = lds_base + (uint32_t)(slot * STAGE * 2);
#pragma unroll
        for (int i = 0; i < XI; ++i)
            DEEP_DMA16(a.in + (size_t)xoff[i] + (ptrdiff_t)koff, __builtin_amdgcn_readfirstlane(xl + (uint32_t)((i * 256 + wave * 64) * 16)));
#pragma unroll
        for (int i = 0; i < WI; ++i)
            DEEP_DMA16(wnext + i * 256 * 8, __builtin_amdgcn_readfirstlane(xl + (uint32_t)(BM * 128 + (i * 256 + wave * 64) * 16)));
        wnext += BN * 64;
        if (++q == a.cpt) { q = 0; ++tap; }
    };

    f32x4 acc[TC][TP];
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
        for (int p = 0; p < TP; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;

#pragma unroll
    for (int i = 0; i < DEEP_NS - 1; ++i)
        if (i < nk) stage(i);
    for (int i = 0; i < nk; ++i) {
        // stages issued so far: min(nk, i + NS - 1); stage i must have landed, the younger ones may stay in flight
        const int ahead = (nk < i + DEEP_NS - 1 ? nk : i + DEEP_NS - 1) - (i + 1);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * LPS) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();            // stage i is complete for every wave; slot (i - 1) % NS has been read by all
        __builtin_amdgcn_sched_barrier(0);
        if (i + DEEP_NS - 1 < nk) stage((i + DEEP_NS - 1) % DEEP_NS);
        const f16* xl = lds + (i % DEEP_NS) * STAGE;
        const f16* wl = xl + BM * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = ((kk * 4 + fk) ^ (frow & 7)) * 8;
            f16x8 xf[TP], wf[TC];
#pragma unroll
            for (int p = 0; p < TP; ++p) xf[p] = *(const f16x8*)(xl + (wp * (BM / WGM) + p * 16 + frow) * 64 + sw);
#pragma unroll
            for (int c = 0; c < TC; ++c) wf[c] = *(const f16x8*)(wl + (wc * (BN / WGN) + c * 16 + frow) * 64 + sw);
#pragma unroll
            for (int c = 0; c < TC; ++c)
#pragma unroll
                for (int p = 0; p < TP; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], xf[p], acc[c][p], 0, 0, 0);
        }
    }

    if (a.ksplit > 1) {
        // slab[split][group][tile][fragment half][thread]: lane-linear 8-byte stores / loads
        const int tile = mtile * a.NT + ntile;
        const size_t tile_floats = (size_t)BM * BN;
        const size_t split_stride = (size_t)gridDim.y * a.MT * a.NT * tile_floats;
        // The partial tiles cross XCDs (one L2 each).  A device-scope fence would do (release = write back the L2, acquire =
        // invalidate it) but every workgroup pays for the whole-cache operations: measured 45-80 us per layer.  Instead the
        // slab is only ever touched with agent-scope relaxed atomic 8-byte accesses (sc1: write-through / L2-coherent
        // reads), ordered against the ticket by the waves' own vmcnt(0) + the workgroup barrier: the hand-off form of
        // MI355X_MICROARCH.md 'Hand-offs measured with sc1 loads', first row (one lane per storing workgroup adds to one
        // counter after every wave's vmcnt(0) and a barrier; the workgroup whose add came last reads; one workgroup per CU).
        unsigned long long* mine = (unsigned long long*)(a.slab + (size_t)blockIdx.z * split_stride
                                                         + ((size_t)blockIdx.y * a.MT * a.NT + tile) * tile_floats) + tid;
#pragma unroll
        for (int c = 0; c < TC; ++c)
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                unsigned long long u[2];
                __builtin_memcpy(u, &acc[c][p], 16);
                __hip_atomic_store(mine + ((c * TP + p) * 2 + 0) * 256, u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + ((c * TP + p) * 2 + 1) * 256, u[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned int* ctr = tile_ctr + blockIdx.y * a.MT * a.NT + tile;
        if (tid == 0) arrived = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (arrived != a.ksplit - 1) return;
        if (tid == 0) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every split has arrived: re-arm for the next launch
        const unsigned long long* all = (const unsigned long long*)(a.slab + ((size_t)blockIdx.y * a.MT * a.NT + tile) * tile_floats) + tid;
#pragma unroll
        for (int c = 0; c < TC; ++c)
#pragma unroll
            for (int p = 0; p < TP; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // software-pipelined: the loads of split z + 1 are in flight while split z is added (the adds stay in split order)
        constexpr int NF = TC * TP * 2;
        unsigned long long buf[2][NF];
        auto fetch = [&](unsigned long long (&dst)[NF], int z) {
            const unsigned long long* pz = all + (size_t)z * (split_stride / 2);
#pragma unroll
            for (int f = 0; f < NF; ++f) dst[f] = __hip_atomic_load(pz + f * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto add = [&](const unsigned long long (&src)[NF]) {
#pragma unroll
            for (int c = 0; c < TC; ++c)
#pragma unroll
                for (int p = 0; p < TP; ++p) {
                    f32x4 v;
                    __builtin_memcpy(&v, &src[(c * TP + p) * 2], 16);
                    acc[c][p] += v;
                }
        };
        fetch(buf[0], 0);
        for (int z = 0; z < a.ksplit; z += 2) {
            if (z + 1 < a.ksplit) fetch(buf[1], z + 1);
            add(buf[0]);
            if (z + 1 < a.ksplit) {
                if (z + 2 < a.ksplit) fetch(buf[0], z + 2);
                add(buf[1]);
            }
        }
    }
    store_nhwc_tile<BM, BN, WGM, WGN, RES>(a, g, acc, mtile, ntile, wp, wc, frow, fk);
}

template <int BN>
static hipError_t launch_deep(const ConvKArgs& a, int groups, unsigned int* tile_ctr, hipStream_t s) {
    const int mt8 = (a.MT + 7) / 8 * 8;
    dim3 grid(mt8 * a.NT, groups, a.ksplit > 1 ? a.ksplit : 1), block(256, 1, 1);
    const size_t lds_bytes = (size_t)DEEP_NS * (128 + BN) * 64 * sizeof(f16);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_mfma_deep_kernel<BN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_mfma_deep_kernel<BN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (a.res) hipLaunchKernelGGL((conv_mfma_deep_kernel<BN, 1>), grid, block, lds_bytes, s, a, tile_ctr);
    else hipLaunchKernelGGL((conv_mfma_deep_kernel<BN, 0>), grid, block, lds_bytes, s, a, tile_ctr);
    return hipGetLastError();
}

hipError_t launch_conv_mfma_deep(const ConvKArgs& a, int bn_tile, int groups, unsigned int* tile_ctr, hipStream_t s) {
    if (a.ksplit > 1 && (!a.slab || !tile_ctr)) return hipErrorInvalidValue;
    ConvKArgs b = a;
    if (b.ksplit < 1) b.ksplit = 1;
    switch (bn_tile) {
        case 128: return launch_deep<128>(b, groups, tile_ctr, s);
        case 64: return launch_deep<64>(b, groups, tile_ctr, s);
        default: return hipErrorInvalidValue;
    }
}

template <int BM, int BN, int WGM, int WGN>
static hipError_t launch_t(const ConvKArgs& a, int groups, int epi, hipStream_t s) {
    const int mt8 = (a.MT + 7) / 8 * 8;
    dim3 grid(mt8 * a.NT, groups, 1), block(256, 1, 1);
    if (epi)
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WGM, WGN, 1, 0>), grid, block, 0, s, a);
    else if (a.res)
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WGM, WGN, 0, 1>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WGM, WGN, 0, 0>), grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_conv_mfma(const ConvKArgs& a, int bn_tile, int groups, int epi_nchw, hipStream_t s) {
    switch (bn_tile) {
        case 128: return launch_t<128, 128, 2, 2>(a, groups, epi_nchw, s);
        case 64: return launch_t<128, 64, 2, 2>(a, groups, epi_nchw, s);
        case 32: return launch_t<128, 32, 4, 1>(a, groups, epi_nchw, s);
        case 16: return launch_t<128, 16, 4, 1>(a, groups, epi_nchw, s);
        default: return hipErrorInvalidValue;
    }
}
