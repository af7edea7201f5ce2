// 3x3 / stride 1 / 64 -> 64 channel convolution (DLA-34 level2's three block convs, models/nets/dla.py:86-100;
// ResNet layer1's four, models/nets/resnet.py:55-72) as a persistent halo-tile kernel with the WHOLE filter bank in
// registers.
//
// These layers are HBM-bound (72.5 GFLOP against 0.25-0.38 GB at bs=32: 30 us of MFMA time, 50-75 us of HBM time), but
// the generic 128-pixel kernel (conv_mfma.hip) runs them at 120-150 us: with 64-channel tiles every (tap, chunk) K-step
// re-stages the pixel tile through L2 -> LDS (9x) and the 24 KB K-step buffer of 43 FLOP/B cannot keep the DMA busy.
// Here
//   * a workgroup (8 waves) owns an 8 x 32 pixel tile and stages its (8+2) x (32+2) halo ONCE (43.5 KB, LDS-DMA,
//     double-buffered: the next tile's halo lands while this one is multiplied); the nine taps read shifted rows of it
//     (bank swizzle of conv_mfma256_halo.hip: 16-byte chunk ^= (x ^ y << 2) & 7);
//   * a wave owns 32 output channels x 64 pixels and keeps its 9 x 2 x 2 weight fragments (144 VGPRs) for the whole
//     launch: no weight traffic after the first 36 loads, no LDS reads for the A operand;
//   * tiles are handed out by one atomic ticket counter per op (zeroed at the head of every forward by the runtime), so a
//     workgroup whose CU was still held by another stream's waves simply takes fewer tiles;
//   * epilogue: bias (+ residual) + ReLU, v_permlane16_swap pairs the two 16-channel MFMA tiles -> 16-byte stores.
// (Round 3, measured and removed: 8 x 16-pixel tiles on FOUR waves with two independent workgroups per CU, so that one multiplies
// while the other runs its epilogue / waits for its halo - same per-wave arithmetic, 250 VGPRs, 2 x 23 KB of halo buffers: 0.106 ms
// against 0.070 for the non-residual layers, 0.112 against 0.092 with a residual, same box.  Twice the items, half the row length
// per DMA, 1.41x instead of 1.33x halo bytes: the per-item costs outweigh the overlap.  The opposite, 16 x 32-pixel items (one (16+2) x
// (32+2) halo = 1.195x, computed as two 8-row passes so that the registers stay as here; half the barriers, tickets and DMA bursts):
// 0.076 against 0.070 without a residual, 0.089-0.094 against 0.090-0.095 with one.  Neither the item size nor the epilogue overlap is
// what bounds these layers: a tile moves 75-107 KB through one CU's memory path in 4.7-6.1 us = 16-17.5 GB/s per CU, ~70 % of the
// 23-25 GB/s a CU streams by LDS-DMA alone (MI355X_MICROARCH.md, ldsdma-fill).)
// K order: tap-major, 64 channels per tap as two 32-deep MFMAs (fp32 accumulate; the order of the sums differs from the
// generic kernel's, results agree to fp32 round-off).
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define C64_HALO_W 34
#define C64_HALO_PIECES (10 * C64_HALO_W * 8)      // 2720 16-byte pieces per halo
#define C64_BUF_PIECES 3072                        // 6 DMA instructions x 512 lanes (the overrun re-stages the last piece)
#define C64_DMA16 RT_DMA16                          // common.h: the one LDS-DMA definition
#define C64_LDS_F16X8(byte_addr) (*(const LDS_AS f16x8*)(uintptr_t)(byte_addr))

template <int RES, int S2D>
__global__ __launch_bounds__(512) void conv64_halo_kernel(const ConvKArgs a, unsigned int* ticket_ctr, const int single) {
    __shared__ __attribute__((aligned(128))) f16 lds[2 * C64_BUF_PIECES * 8];
    __shared__ int tk[3];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wp = wave >> 1;                  // 32-channel half, pixel-row pair of the tile
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
    f32x4 bv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) bv[c] = *(const f32x4*)(a.bias + g.bias_off + wc * 32 + c * 16 + fk * 4);

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
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;

    // Tickets: three drawn up front (current tile, next, the one after), one more per tile by thread 0, published through
    // LDS behind the next tile's barrier.  Draws past the end are harmless: the runtime zeroes the counter at the head of
    // every forward.
    // (ONE draw of three: three separate returning atomics per workgroup - 768 on one word, three round trips in a row
    // before the first MFMA - cost 8-10 us per launch; the word serves ~88 draws per microsecond)
    // `single` (no more work items than CUs: small batches): workgroup b takes item b and nothing else - with three tickets per
    // draw only a third of the workgroups would work, three items each in a row (bs=1 at 384 x 1280: 40 of 120)
    int cur, nxt, nn;
    if (single) {
        cur = blockIdx.x; nxt = nn = total;
    } else {
        if (tid == 0) tk[0] = (int)atomicAdd(ticket_ctr, 3u);
        __syncthreads();
        const int tk0 = __builtin_amdgcn_readfirstlane(tk[0]);
        cur = tk0; nxt = tk0 + 1; nn = tk0 + 2;
    }
    if (cur >= total) return;
    __syncthreads();                                // tk[0..1] are reused as the per-tile slots below

    auto halo_origin = [&](int v) -> size_t {
        const int n = v / tpi, r = v - n * tpi;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        return ((size_t)(n * a.in_Hp + ty * 8 - 1 + a.in_P) * a.in_Wp + tx * 32 - 1 + a.in_P) * a.in_C + g.in_coff;
    };
    auto stage = [&](int v, int par) {
        const f16* src = a.in + halo_origin(v);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            C64_DMA16(src + poff[i], __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((par * C64_BUF_PIECES + i * 512 + wave * 64) * 16)));
    };

    // LDS byte offsets of this lane's four pixel fragments at tap (0, 0) (halo row = 1 + tile row, halo column = 1 + column)
    //   fragment p: tile row 2*wp + (p >> 1), columns (p & 1) * 16 + frow
    const uint32_t lane_px = (uint32_t)(((2 * wp + 1) * C64_HALO_W + frow + 1) * 128);
    // swizzled 16-byte slot of this lane's k-chunk fk for the three column shifts dx = -1, 0, +1 (even halo row, k half 0)
    const uint32_t ck_m = (uint32_t)((((frow + 0) ^ fk) & 7) << 4), ck_0 = (uint32_t)((((frow + 1) ^ fk) & 7) << 4),
                   ck_p = (uint32_t)((((frow + 2) ^ fk) & 7) << 4);
    const f16 lo = a.relu ? (f16)0.f : (f16)(-__builtin_inff());
    const f16x4 lo4 = {lo, lo, lo, lo};
    const int so = (fk & 1) * 16 + (fk >> 1) * 8;

    stage(cur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int par = 0, it = 0;
    for (;;) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();               // halo of `cur` has landed for every wave; buffer par ^ 1 is free
        __builtin_amdgcn_sched_barrier(0);
        if (it > 0) nn = __builtin_amdgcn_readfirstlane(tk[(it - 1) & 1]);      // drawn during the previous tile
        const bool more = nxt < total;
        if (more) stage(nxt, par ^ 1);

        f32x4 acc[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const uint32_t hb = lds_base + (uint32_t)(par * C64_BUF_PIECES * 16);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            // address of fragment (p = 0, k half 0) at this tap; the other fragments are compile-time offsets from it and
            // a flip of bit 6 (an odd halo row swaps the two k-half slots): two address registers per tap
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

        // ---- epilogue of `cur`
        {
            const int n = cur / tpi, r = cur - n * tpi;
            const int ty = r / tiles_x, tx = r - ty * tiles_x;
            size_t opix[4];
            f16x4 rv[4][2];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int oy = ty * 8 + 2 * wp + (p >> 1), ox = tx * 32 + (p & 1) * 16 + frow;
                opix[p] = ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + g.out_coff + wc * 32;
                if (RES) {
                    const f16* rp = a.res + ((size_t)(n * a.res_Hp + oy + a.res_P) * a.res_Wp + ox + a.res_P) * a.res_C + g.res_coff + wc * 32 + fk * 4;
#pragma unroll
                    for (int c = 0; c < 2; ++c) rv[p][c] = *(const f16x4*)(rp + c * 16);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t u[2][2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    f32x4 v = acc[c][p] + bv[c];
                    if (RES) {
                        const f16x4 rr = rv[p][c];
                        v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3];
                    }
                    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    h = __builtin_elementwise_max(h, lo4);
                    __builtin_memcpy(u[c], &h, 8);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                *(u32x4*)((f16*)a.out + opix[p] + so) = o;
                if (S2D) {
                    // second copy in space-to-depth layout (ConvKArgs.s2d: the neck's up-fold reads the feature from its transposed conv's
                    // input grid): fragment p = tile row 2 wp + (p >> 1), column (p & 1) * 16 + frow -> phase (p >> 1, frow & 1)
                    uint32_t l2 = (uint32_t)lane;
                    asm volatile("" : "+v"(l2));        // (lane offset recomputed here: no register to carry it across the conv)
                    const uint32_t fr2 = l2 & 15, f2 = l2 >> 4;
                    const uint32_t loff = (fr2 >> 1) * (uint32_t)a.s_C + (fr2 & 1) * 64 + (f2 & 1) * 16 + (f2 >> 1) * 8;
                    const f16* sb = a.s2d + ((size_t)(n * a.s_Hp + ty * 4 + wp + a.s_P) * a.s_Wp + tx * 16 + (p & 1) * 8 + a.s_P) * a.s_C + a.s_coff + (p >> 1) * 128 + wc * 32;
                    *(u32x4*)((f16*)sb + loff) = o;
                }
            }
        }
        if (!more) break;
        // the next tile's halo (issued before this tile's loads and stores) must have landed; the stores may stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 + 4 * S2D) : "memory");
        cur = nxt; nxt = nn;
        par ^= 1;
        ++it;
    }
}

bool conv64_halo_supported(const ConvKArgs& a, int groups) {
    if (groups != 1 || a.cin != 64 || a.cout != 64 || a.ntaps != 9 || a.in_stride != 1 || a.out_scale != 1 || a.in_P < 1) return false;
    if (a.Wm % 32 || (a.HmWm / a.Wm) % 8 || a.M % a.HmWm) return false;
    const int pitch = a.in_Wp * a.in_C;
    for (int t = 0; t < 9; ++t)
        if (a.g[0].tap_off[t] != (t / 3 - 1) * pitch + (t % 3 - 1) * a.in_C) return false;     // 3x3, dilation 1, row-major taps
    return true;
}

hipError_t launch_conv64_halo(const ConvKArgs& a, int cu_count, unsigned int* ticket_ctr, hipStream_t s) {
    const int total = (a.M / a.HmWm) * (a.Wm >> 5) * ((a.HmWm / a.Wm) >> 3);
    int grid = cu_count < total ? cu_count : total;
    const int single = total <= cu_count ? 1 : 0;
    if (a.res && a.s2d) hipLaunchKernelGGL((conv64_halo_kernel<1, 1>), dim3(grid), dim3(512), 0, s, a, ticket_ctr, single);
    else if (a.res) hipLaunchKernelGGL((conv64_halo_kernel<1, 0>), dim3(grid), dim3(512), 0, s, a, ticket_ctr, single);
    else if (a.s2d) hipLaunchKernelGGL((conv64_halo_kernel<0, 1>), dim3(grid), dim3(512), 0, s, a, ticket_ctr, single);
    else hipLaunchKernelGGL((conv64_halo_kernel<0, 0>), dim3(grid), dim3(512), 0, s, a, ticket_ctr, single);
    return hipGetLastError();
}
