// Tail of a DLA level-1 tree on 64 channels (DLA-34 level2, models/nets/dla.py:186-206) in ONE persistent launch:
//   x2   = ReLU(BN(conv3x3(t)) + x1)                    tree2's second block conv + residual (dla.py:86-100)
//   out  = ReLU(BN(conv1x1(cat[x2, x1])))               the tree's root (dla.py:233-241, root_residual = False)
//   pool = max_pool2d(out, 2, 2)                        the NEXT level's `downsample` (dla.py:170-172,190), optional
//   s2d  = out in space-to-depth layout                 optional second copy for the neck (plan.py: _neck_up_folds)
// As three launches these are HBM-bound (conv 0.093 ms reading t and x1 and writing x2, root 0.081 ms reading x2 and x1
// again and writing out, pool 0.029 ms reading out again: 7.25 passes over a 126 MB map at bs=32).  Fused, x2 never
// leaves the registers, x1 is read once and the pooled map is written from the root's output registers: 3.25 passes.
//
// The conv part is conv64_halo.hip unchanged: 8 waves on an 8 x 32 pixel tile, (8+2) x (32+2) halo double-buffered by
// LDS-DMA, wave (wc, wp) = 32 output channels x 64 pixels (tile rows 2wp, 2wp+1) with its 36 filter fragments in registers,
// one ticket counter per op.  What follows the nine taps:
//   * x2 tile = ReLU(acc + x1) in fp16, still in the MFMA C layout: lane (frow, fk) holds channels wc*32 + c*16 + fk*4 + {0..3}
//     (c = 0, 1) of pixel frow.  A 16x16x32 MFMA takes as B operand, per lane, 8 values of K-slot fk*8 + j: the root's
//     weights are packed on the host with the K order PERMUTED to k(fk, j) = (j >> 2) * 16 + fk * 4 + (j & 3), so the lane's
//     two f16x4 C fragments, concatenated, ARE its B fragment - no transpose, no LDS (the flash-attention P.V chaining);
//     the x1 registers loaded for the residual add serve the same way as the root's second operand.
//   * a wave holds only its 32 of the 64 channels, its partner (wc ^ 1) the other 32 of the same pixels: the two waves swap
//     HALVES of their pixels through LDS (fp16 fragments, 4 KB per wave; fp32 partial sums would be twice that), after
//     which wave wc owns all 128 root input channels of columns wc*16 .. wc*16+15 of both its rows: 4 K-steps x 4 output
//     tiles x 2 pixel fragments = 32 MFMAs on top of the conv's 144.  Root weights (16 KB) and both biases sit in LDS.
//   * epilogue: bias (accumulator seed) + ReLU, v_permlane16_swap -> 16-byte stores of `out`; the 2x2 max is one packed
//     max between the wave's two rows (same lane) and one with the neighbouring lane (DPP quad_perm); even lanes store
//     channels 0-31 of the pooled pixel, odd lanes 32-63.
// Two barriers per tile instead of one (fragments written -> read); the exchange area is reused only behind the next
// tile's first barrier.  Sums: conv as conv64_halo; root in the order own x2 half, partner's x2 half, own x1 half, partner's
// (position-determined, bit-identical run to run).
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define C64_HALO_W 34
#define C64_HALO_PIECES (10 * C64_HALO_W * 8)      // 2720 16-byte pieces per halo
#define C64_BUF_PIECES 3072                        // 6 DMA instructions x 512 lanes (the overrun re-stages the last piece)
#define CR_W_PIECES 1024                           // root weights: [4 output tiles][4 K-steps][64 lanes] 16-byte fragments
#define CR_X_PIECES 2048                           // exchange area: [8 waves][4 fragments][64 lanes]
#define C64_DMA16 RT_DMA16                          // common.h: the one LDS-DMA definition
#define C64_LDS_F16X8(byte_addr) (*(const LDS_AS f16x8*)(uintptr_t)(byte_addr))
#define C64_LDS_F16X8_W(byte_addr) (*(LDS_AS f16x8*)(uintptr_t)(byte_addr))

__device__ __forceinline__ uint32_t cr_pkmax(uint32_t x, uint32_t y) {
    f16x2 a, b;
    __builtin_memcpy(&a, &x, 4); __builtin_memcpy(&b, &y, 4);
    a = __builtin_elementwise_max(a, b);
    uint32_t r; __builtin_memcpy(&r, &a, 4);
    return r;
}

// NORM = 0: the ordinary copy of `out` is not written (r.out null): every reader takes the space-to-depth copy (S2D = 1)
template <int POOL, int S2D, int NORM>
__global__ __launch_bounds__(512) void conv64_root_kernel(const ConvKArgs a, const RootKArgs r, unsigned int* ticket_ctr, const int single) {
    __shared__ __attribute__((aligned(128))) f16 lds[(2 * C64_BUF_PIECES + CR_W_PIECES + CR_X_PIECES) * 8];
    __shared__ __attribute__((aligned(16))) float sbias[128];      // [0, 64): conv bias, [64, 128): root bias
    __shared__ int tk[3];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wp = wave >> 1;                  // 32-channel half (conv) / 16-column half (root), pixel-row pair
    const int frow = lane & 15, fk = lane >> 4;
    const ConvGroupArgs& g = a.g[0];

    const int Hm = a.HmWm / a.Wm;
    const int tiles_x = a.Wm >> 5, tpi = tiles_x * (Hm >> 3);
    const int total = (a.M / a.HmWm) * tpi;

    // the filter bank of this wave's 32 output channels: [tap][k half][16-channel tile], MFMA A fragments
    f16x8 wreg[9][2][2];
    {
        const f16* wb = a.wgt + g.w_off;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    wreg[t][kk][c] = *(const f16x8*)(wb + ((size_t)(((t * 2 + kk) * 4 + wc * 2 + c) * 64 + lane)) * 8);
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;
    const uint32_t lds_rw = lds_base + (uint32_t)(2 * C64_BUF_PIECES * 16);
    const uint32_t lds_xc = lds_rw + (uint32_t)(CR_W_PIECES * 16);
    // root weights and the two biases -> LDS (once per launch)
    {
        const f16x8* src = (const f16x8*)r.w;
        C64_LDS_F16X8_W(lds_rw + (uint32_t)(tid * 16)) = src[tid];
        C64_LDS_F16X8_W(lds_rw + (uint32_t)((tid + 512) * 16)) = src[tid + 512];
        if (tid < 64) sbias[tid] = a.bias[g.bias_off + tid];
        else if (tid < 128) sbias[tid] = r.bias[tid - 64];
    }

    // per-thread source offsets (elements, relative to the halo origin) of its six DMA pieces
    uint32_t poff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int p = i * 512 + tid;
        p = p < C64_HALO_PIECES ? p : C64_HALO_PIECES - 1;
        const int hq = p >> 3, hcs = p & 7;
        const int hy = hq / C64_HALO_W, hx = hq - hy * C64_HALO_W;
        poff[i] = (uint32_t)((hy * a.in_Wp + hx) * a.in_C + ((hcs ^ ((hx ^ (hy << 2)) & 7)) * 8));
    }

    // tickets: as conv64_halo.hip (one draw of three up front, one more per tile by thread 0)
    int cur, nxt, nn;
    if (single) {
        cur = blockIdx.x; nxt = nn = total;
        __syncthreads();                            // root weights / biases visible
    } else {
        if (tid == 0) tk[0] = (int)atomicAdd(ticket_ctr, 3u);
        __syncthreads();
        const int tk0 = __builtin_amdgcn_readfirstlane(tk[0]);
        cur = tk0; nxt = tk0 + 1; nn = tk0 + 2;
    }
    if (cur >= total) return;
    __syncthreads();                                // tk[0..1] are reused as the per-tile slots below

    auto halo_origin = [&](int v) -> size_t {
        const int n = v / tpi, rr = v - n * tpi;
        const int ty = rr / tiles_x, tx = rr - ty * tiles_x;
        return ((size_t)(n * a.in_Hp + ty * 8 - 1 + a.in_P) * a.in_Wp + tx * 32 - 1 + a.in_P) * a.in_C + g.in_coff;
    };
    auto stage = [&](int v, int par) {
        const f16* src = a.in + halo_origin(v);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            C64_DMA16(src + poff[i], __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((par * C64_BUF_PIECES + i * 512 + wave * 64) * 16)));
    };

    const uint32_t lane_px = (uint32_t)(((2 * wp + 1) * C64_HALO_W + frow + 1) * 128);
    const uint32_t ck_m = (uint32_t)((((frow + 0) ^ fk) & 7) << 4), ck_0 = (uint32_t)((((frow + 1) ^ fk) & 7) << 4),
                   ck_p = (uint32_t)((((frow + 2) ^ fk) & 7) << 4);
    const f16 lo = a.relu ? (f16)0.f : (f16)(-__builtin_inff());
    const f16x4 lo4 = {lo, lo, lo, lo};
    const uint32_t lane_roff = (uint32_t)(frow * a.res_C + fk * 4);
    // exchange slots: this wave writes [wave][slot][lane], reads [wave ^ 1][slot][lane]; slot = q (x2), 2 + q (x1)
    // root A fragments [ct][s][lane]: K-step s of the packed weights = cat channels s*32 .. s*32+31 ([x2 | x1]); this wave
    // multiplies its own halves (s = wc, 2 + wc) first, then the partner's

    stage(cur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int par = 0, it = 0;
    for (;;) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();               // halo of `cur` has landed for every wave; buffer par ^ 1 and the exchange area are free
        __builtin_amdgcn_sched_barrier(0);
        if (it > 0) nn = __builtin_amdgcn_readfirstlane(tk[(it - 1) & 1]);      // drawn during the previous tile
        const bool more = nxt < total;
        if (more) stage(nxt, par ^ 1);

        const int n = cur / tpi, rr = cur - n * tpi;
        const int ty = rr / tiles_x, tx = rr - ty * tiles_x;
        // x1 tile (residual of the conv, second operand of the root)
        f16x4 rv[4][2];
        {
            // wave-uniform base (SGPR pair) + one 32-bit lane offset: the eight loads cost one address register
            const f16* rbase = a.res + ((size_t)(n * a.res_Hp + ty * 8 + 2 * wp + a.res_P) * a.res_Wp + tx * 32 + a.res_P) * a.res_C + g.res_coff + wc * 32;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f16* rp = rbase + (size_t)(((p >> 1) * a.res_Wp + (p & 1) * 16) * a.res_C);
#pragma unroll
                for (int c = 0; c < 2; ++c) rv[p][c] = *(const f16x4*)(rp + (lane_roff + (uint32_t)(c * 16)));
            }
        }
        // (left to the scheduler, the eight loads sink to the last taps of the MFMA loop; pinned here by a sched_barrier they
        // cost 16 registers across the loop: 14 spills at the 256-register budget of two waves per SIMD)

        f32x4 acc[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 bv = *(const f32x4*)(sbias + wc * 32 + c * 16 + fk * 4);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[c][p] = bv;
        }
        const uint32_t hb = lds_base + (uint32_t)(par * C64_BUF_PIECES * 16);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const uint32_t ck = dx < 0 ? ck_m : (dx > 0 ? ck_p : ck_0);
            const uint32_t xt0 = hb + lane_px + (uint32_t)((dy * C64_HALO_W + dx) * 128) + ck;
            const uint32_t xu = xt0 ^ (uint32_t)(((1 + dy) & 1) << 6), xw = xu ^ 64u;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f16x8 xf[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const uint32_t ad = (((p >> 1) ^ kk) ? xw : xu) + (uint32_t)(((p >> 1) * C64_HALO_W + (p & 1) * 16) * 128);
                    xf[p] = C64_LDS_F16X8(ad);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[t][kk][c], xf[p], acc[c][p], 0, 0, 0);
            }
        }
        // one more ticket (for the tile after `nn`): drawn here, where this wave's DMA has long landed (the compiler
        // waits for the returning atomic with vmcnt(0)); slot it & 1 was read by everyone two barriers ago
        if (!single && tid == 0) tk[it & 1] = (int)atomicAdd(ticket_ctr, 1u);

        // ---- x2 = ReLU(conv + x1) as B fragments of the root; the x1 registers likewise
        f16x8 bx2[4], bx1[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f16x4 h[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x4 v = acc[c][p];
                const f16x4 q4 = rv[p][c];
                v[0] += (float)q4[0]; v[1] += (float)q4[1]; v[2] += (float)q4[2]; v[3] += (float)q4[3];
                h[c] = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                h[c] = __builtin_elementwise_max(h[c], lo4);
            }
            bx2[p] = (f16x8){h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
            bx1[p] = (f16x8){rv[p][0][0], rv[p][0][1], rv[p][0][2], rv[p][0][3], rv[p][1][0], rv[p][1][1], rv[p][1][2], rv[p][1][3]};
        }
        // keep fragments 2q + wc (columns wc*16.., rows 2wp + q), hand fragments 2q + 1 - wc to the partner
        // (lane-derived LDS / store offsets of the root phase are recomputed per tile from an opaque copy of the lane id: the kernel
        // sits at the 256-register budget of two waves per SIMD, and loop-invariant values would be carried across the conv)
        uint32_t l2 = (uint32_t)lane;
        asm volatile("" : "+v"(l2));
        const uint32_t xc_wr = lds_xc + (uint32_t)(wave * 256 * 16) + l2 * 16, xc_rd = lds_xc + (uint32_t)((wave ^ 1) * 256 * 16) + l2 * 16;
        const uint32_t rw_lane = lds_rw + l2 * 16;
        const uint32_t fr2 = l2 & 15, fk2 = l2 >> 4;
        const uint32_t so2 = (fk2 & 1) * 16 + (fk2 >> 1) * 8;
        f16x8 kx2[2], kx1[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            kx2[q] = wc ? bx2[2 * q + 1] : bx2[2 * q];
            kx1[q] = wc ? bx1[2 * q + 1] : bx1[2 * q];
            C64_LDS_F16X8_W(xc_wr + (uint32_t)(q * 1024)) = wc ? bx2[2 * q] : bx2[2 * q + 1];
            C64_LDS_F16X8_W(xc_wr + (uint32_t)((2 + q) * 1024)) = wc ? bx1[2 * q] : bx1[2 * q + 1];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();               // every wave's fragments are in the exchange area
        __builtin_amdgcn_sched_barrier(0);
        f16x8 px2[2], px1[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            px2[q] = C64_LDS_F16X8(xc_rd + (uint32_t)(q * 1024));
            px1[q] = C64_LDS_F16X8(xc_rd + (uint32_t)((2 + q) * 1024));
        }

        // ---- root: out[64 channels] of 2 x 16 pixels, two output-channel tiles at a time
        f16* const obase = r.out + ((size_t)(n * r.o_Hp + ty * 8 + 2 * wp + r.o_P) * r.o_Wp + tx * 32 + wc * 16 + r.o_P) * r.o_C + r.o_coff;   // wave-uniform
        const uint32_t olane = fr2 * (uint32_t)r.o_C + so2;
        uint32_t pl[2][4];                          // packed 2x2-max candidates: [channel pair-of-tiles][dword]
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            f32x4 ra[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 bv = *(const f32x4*)(sbias + 64 + (cp * 2 + c) * 16 + fk2 * 4);
                ra[c][0] = bv; ra[c][1] = bv;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // j = 0: own x2 half, 1: partner's x2 half, 2: own x1 half, 3: partner's x1 half
                const int s = (j >> 1) * 2 + ((j & 1) ? 1 - wc : wc);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 wf = C64_LDS_F16X8(rw_lane + (uint32_t)((((cp * 2 + c) * 4 + s) * 64) * 16));
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f16x8 bf = j == 0 ? kx2[q] : (j == 1 ? px2[q] : (j == 2 ? kx1[q] : px1[q]));
                        ra[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, bf, ra[c][q], 0, 0, 0);
                    }
                }
            }
            uint32_t o[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint32_t u[2][2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x4 v = ra[c][q];
                    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    if (r.relu) h = __builtin_elementwise_max(h, (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f});
                    __builtin_memcpy(u[c], &h, 8);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
                o[q][0] = s0[0]; o[q][1] = s1[0]; o[q][2] = s0[1]; o[q][3] = s1[1];
                const u32x4 ov = {o[q][0], o[q][1], o[q][2], o[q][3]};
                if (NORM) *(u32x4*)(obase + (size_t)(q * r.o_Wp * r.o_C + cp * 32) + olane) = ov;
                if (S2D) {
                    // space-to-depth copy: this lane's pixel (row 2 wp + q, column wc * 16 + frow of the tile) has phase (q, frow & 1);
                    // wave-uniform base + a 32-bit lane offset recomputed here (the kernel has no register to keep it in)
                    const uint32_t loff = (fr2 >> 1) * (uint32_t)r.s_C + (fr2 & 1) * 64 + so2;
                    const f16* sb = r.s2d + ((size_t)(n * r.s_Hp + ty * 4 + wp + r.s_P) * r.s_Wp + tx * 16 + wc * 8 + r.s_P) * r.s_C + r.s_coff + q * 128 + cp * 32;
                    *(u32x4*)((f16*)sb + loff) = ov;
                }
            }
            if (POOL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) pl[cp][e] = cr_pkmax(o[0][e], o[1][e]);          // the two rows of the window
            }
        }
        if (POOL) {
            // the neighbouring column sits in lane ^ 1; even lanes then store channels so .. so+7, odd lanes 32 + so ..
            uint32_t m[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t mine = (fr2 & 1) ? pl[1][e] : pl[0][e];
                const uint32_t give = (fr2 & 1) ? pl[0][e] : pl[1][e];     // what the neighbour wants from me
                const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
                m[e] = cr_pkmax(mine, got);
            }
            f16* const pbase = r.pool + ((size_t)(n * r.p_Hp + ty * 4 + wp + r.p_P) * r.p_Wp + tx * 16 + wc * 8 + r.p_P) * r.p_C + r.p_coff;   // wave-uniform
            const u32x4 pv = {m[0], m[1], m[2], m[3]};
            *(u32x4*)(pbase + ((fr2 >> 1) * (uint32_t)r.p_C + (fr2 & 1) * 32 + so2)) = pv;
        }
        if (!more) break;
        // the next tile's halo (issued before this tile's loads and stores) must have landed; the stores may stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * NORM + POOL + 4 * S2D) : "memory");
        cur = nxt; nxt = nn;
        par ^= 1;
        ++it;
    }
}

bool conv64_halo_supported(const ConvKArgs& a, int groups);

hipError_t launch_conv64_root(const ConvKArgs& a, const RootKArgs& r, int cu_count, unsigned int* ticket_ctr, hipStream_t s) {
    const int total = (a.M / a.HmWm) * (a.Wm >> 5) * ((a.HmWm / a.Wm) >> 3);
    const int grid = cu_count < total ? cu_count : total;
    const int single = total <= cu_count ? 1 : 0;
    if (!r.out && !r.s2d) return hipErrorInvalidValue;
    if (r.pool && r.s2d && !r.out) hipLaunchKernelGGL((conv64_root_kernel<1, 1, 0>), dim3(grid), dim3(512), 0, s, a, r, ticket_ctr, single);
    else if (r.s2d && !r.out) hipLaunchKernelGGL((conv64_root_kernel<0, 1, 0>), dim3(grid), dim3(512), 0, s, a, r, ticket_ctr, single);
    else if (r.pool && r.s2d) hipLaunchKernelGGL((conv64_root_kernel<1, 1, 1>), dim3(grid), dim3(512), 0, s, a, r, ticket_ctr, single);
    else if (r.pool) hipLaunchKernelGGL((conv64_root_kernel<1, 0, 1>), dim3(grid), dim3(512), 0, s, a, r, ticket_ctr, single);
    else if (r.s2d) hipLaunchKernelGGL((conv64_root_kernel<0, 1, 1>), dim3(grid), dim3(512), 0, s, a, r, ticket_ctr, single);
    else hipLaunchKernelGGL((conv64_root_kernel<0, 0, 1>), dim3(grid), dim3(512), 0, s, a, r, ticket_ctr, single);
    return hipGetLastError();
}
