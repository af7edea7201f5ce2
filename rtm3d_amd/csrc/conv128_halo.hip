// 3x3 / stride 1 convolutions with multiples of 128 channels in and out on maps whose width is a multiple of 32 (DLA-34
// level3: 128 -> 128 on 48 x 160, models/nets/dla.py:86-100; ResNet-18/34 layer2, models/nets/resnet.py:55-72) as a
// persistent halo-tile kernel.
//
// The generic 128-pixel kernel (conv_mfma.hip) re-stages a 128-pixel x 64-channel operand tile per (tap, chunk) K-step:
// 64 FLOP per staged byte, and these layers ran at 650-860 TFLOP/s on it, bound by the L2 -> LDS DMA rate.  Here
//   * a workgroup (8 waves) owns a WORK ITEM = (8 x 32 pixel tile, 128 output channels).  Per 64-channel input chunk it
//     stages the (8+2) x (32+2) halo ONCE by LDS-DMA and the nine taps read shifted rows of it (layout and bank swizzle of
//     conv64_halo.hip): the pixel operand costs 1/9 of the bytes, 200 FLOP per staged byte overall.  (An 8 x 16 pixel x
//     256 channel item for DLA level4's 24 x 80 map was built and measured: 0.105-0.120 ms against 0.085-0.091 on the
//     generic kernel - 32 KB of weights per K-step for the same 32 MFMAs per wave; removed.)
//   * the unit of the pipeline is a PHASE = (item, chunk): nine K-steps (taps).  While phase p is multiplied out of halo
//     buffer p & 1, the halo of phase p + 1 (the item's next chunk, or the next item's first) lands in the other buffer,
//     one DMA instruction per wave and K-step;
//   * weights stream through a 3-slot ring of 16 KB K-step tiles (128 rows x 64 channels, the generic kernel's packed
//     layout for 128-channel tiles, so no second copy of the weights exists): the tile of K-step s + 2 is issued in K-step s.
//     Every wave issues exactly 3 DMA instructions per K-step (2 weight + 1 halo; halo slots past the end re-stage the
//     last piece), so the wait in front of a K-step is the immediate s_waitcnt vmcnt(3);
//   * one raw s_barrier per K-step; a wave owns 64 pixels x 64 channels (16 accumulator tiles);
//   * measured on this kernel and not kept (PMC: MFMA pipe 35-47 % busy, 0 LDS bank conflicts, waves 45 % of their cycles in
//     s_waitcnt): either DMA stream switched off (timing only) -2 % / 0 %; a 4-slot weight ring with three K-steps for a
//     tile to land 0 %; operand fragments of K-step s + 1 fetched during the MFMAs of K-step s (+80 VGPRs) -3 %; the two
//     wave groups half a K-step apart with two barriers per K-step (the conv256 stagger) +10 %; no LDS operand reads at all
//     (timing only) -12 %, no barriers -2 %: what is left is per-item cost (an item is 9.7 us of MFMA time, a conv256 tile 40);
//   * items are handed out by one atomic ticket counter per op (zeroed at the head of every forward by the runtime): ONE draw
//     of three up front, then one per item by thread 0 at the item's end.  (Measured and not kept: issuing the per-item
//     draw early and picking it up later through a sentinel - no gain once the up-front draws were one, and unsafe in
//     principle: the compiler may copy a register that the hardware is still going to write; letting it land in an AGPR
//     instead makes the allocator split the file 128/128 and spill.  A static first item per workgroup: same speed alone,
//     slower with the 3D decode beside it.);
//     consecutive items are the channel tiles of one pixel tile (shared halo in L2);
//   * (round 3, measured and not kept: the wave's bias values loaded once before the item loop instead of in every epilogue, so
//     that no compiler-visible vector-memory load - and no s_waitcnt vmcnt(0) in front of the epilogue arithmetic - is left
//     there: 0.074 / 0.087 ms with and without, same box; the drain is not what an item waits for);
//   * epilogue: bias (+ residual) + ReLU, v_permlane16_swap pairs two 16-channel MFMA tiles -> 16-byte stores.  (Keeping the
//     previous tile's stores and this tile's residual loads in flight across the first K-steps, with the wait counts
//     raised accordingly, was measured: no change - the residual layers are 12 us slower because they move 63 MB more.)
// K order: chunk-major, tap-minor (fp32 accumulate; the order of the sums differs from the generic kernel's tap-major
// order, results agree to fp32 round-off).
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define C128_NSW 3
#define C128_DMA16 RT_DMA16                         // common.h: the one LDS-DMA definition
#define C128_LDS_F16X8(byte_addr) (*(const LDS_AS f16x8*)(uintptr_t)(byte_addr))

template <int RES>
__global__ __launch_bounds__(512) void conv128_halo_kernel(const ConvKArgs a, unsigned int* ticket_ctr, const int single) {
    constexpr int TW = 32, CT = 128;
    constexpr int HW = TW + 2;                              // halo row pitch in pixels
    constexpr int HALO_PIECES = 10 * HW * 8;                // 2720 16-byte pieces per 64-channel halo
    constexpr int XI = (HALO_PIECES + 511) / 512;           // 6 DMA instructions per halo
    constexpr int XBUF_PIECES = XI * 512;
    constexpr int WSLOT_PIECES = CT * 8;                    // CT rows x 128 B
    constexpr int WI = WSLOT_PIECES / 512;                  // 2 weight DMA instructions per wave and K-step
    constexpr int WCN = CT / 64;                            // 2 64-channel wave columns
    constexpr int WPN = 8 / WCN;                            // 4 pixel groups of 64 (2 rows x 32)
    constexpr int ROWS = 8 / WPN;                           // tile rows per wave
    __shared__ __attribute__((aligned(128))) f16 lds[(2 * XBUF_PIECES + C128_NSW * WSLOT_PIECES) * 8];
    __shared__ int tk[3];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % WCN, wp = wave / WCN;               // 64-channel column, pixel group of the tile
    const int frow = lane & 15, fk = lane >> 4;
    const ConvGroupArgs& g = a.g[0];

    const int Hm = a.HmWm / a.Wm;
    const int tiles_x = a.Wm / TW, tpi = tiles_x * (Hm >> 3);
    const int NT = a.cout / CT, cpt = a.cpt;                  // channel tiles per pixel tile, 64-channel input chunks (even)
    const int total = (a.M / a.HmWm) * tpi * NT;

    // per-thread source offsets (elements, relative to the halo origin of chunk 0) of its halo pieces
    uint32_t poff[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        int p = i * 512 + tid;
        p = p < HALO_PIECES ? p : HALO_PIECES - 1;
        const int hq = p >> 3, hcs = p & 7;
        const int hy = hq / HW, hx = hq - hy * HW;
        poff[i] = (uint32_t)((hy * a.in_Wp + hx) * a.in_C + ((hcs ^ ((hx ^ (hy << 2)) & 7)) * 8));
    }
    const uint32_t lds_x = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;
    const uint32_t lds_w = lds_x + (uint32_t)(2 * XBUF_PIECES * 16);
    const f16* const wsrc = a.wgt + g.w_off + tid * 8;       // this thread's first piece of a K-step tile

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
    __syncthreads();                                // tk[0..1] are reused as the per-item slots below

    // item v = pixel tile v / NT, channel tile v % NT
    auto halo_origin = [&](int v) -> const f16* {
        const int pt = v / NT;
        const int n = pt / tpi, r = pt - n * tpi;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        return a.in + ((size_t)(n * a.in_Hp + ty * 8 - 1 + a.in_P) * a.in_Wp + tx * TW - 1 + a.in_P) * a.in_C + g.in_coff;
    };
    // one halo DMA instruction (pieces i * 512 .. i * 512 + 511) of the 64-channel chunk at `src` into halo buffer xb
    auto issue_x = [&](const f16* src, int xb, int i, uint32_t off) {
        C128_DMA16(src + off, __builtin_amdgcn_readfirstlane(lds_x + (uint32_t)((xb * XBUF_PIECES + i * 512 + wave * 64) * 16)));
    };
    // the K-step tile (channel tile nt, tap, chunk) of the weights into ring slot `slot`: WI DMA instructions per wave
    auto issue_w = [&](int nt, int tap, int chunk, int slot) {
        const f16* src = wsrc + (size_t)((nt * 9 + tap) * cpt + chunk) * (CT * 64);
#pragma unroll
        for (int i = 0; i < WI; ++i)
            C128_DMA16(src + i * 512 * 8, __builtin_amdgcn_readfirstlane(lds_w + (uint32_t)((slot * WSLOT_PIECES + i * 512 + wave * 64) * 16)));
    };

    // LDS byte offsets of this lane's four pixel fragments at tap (0, 0): halo row = 1 + tile row, halo column = 1 + column;
    //   fragment p: tile row 2 wp + (p >> 1), columns (p & 1) * 16 + frow
    const uint32_t lane_px = (uint32_t)(((ROWS * wp + 1) * HW + frow + 1) * 128);
    const uint32_t ck_m = (uint32_t)((((frow + 0) ^ fk) & 7) << 4), ck_0 = (uint32_t)((((frow + 1) ^ fk) & 7) << 4),
                   ck_p = (uint32_t)((((frow + 2) ^ fk) & 7) << 4);
    // weight fragments: row = wc * 64 + c * 16 + frow of the K-step tile, 16-byte slot (kk * 4 + fk) ^ (frow & 7)
    const uint32_t lane_w = (uint32_t)((wc * 64 + frow) * 128);
    const uint32_t wk0 = (uint32_t)(((fk ^ (frow & 7)) & 7) << 4), wk1 = (uint32_t)((((4 + fk) ^ (frow & 7)) & 7) << 4);
    const f16 lo = a.relu ? (f16)0.f : (f16)(-__builtin_inff());
    const f16x4 lo4 = {lo, lo, lo, lo};
    const int so = (fk & 1) * 16 + (fk >> 1) * 8;

    // prologue: halo of (cur, chunk 0), weight tiles of K-steps 0 and 1
    {
        const f16* src = halo_origin(cur);
#pragma unroll
        for (int i = 0; i < XI; ++i) issue_x(src, 0, i, poff[i]);
        issue_w(cur % NT, 0, 0, 0);
        issue_w(cur % NT, 1, 0, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int it = 0;
    f32x4 acc[4][4];
    for (;;) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool more = nxt < total;
        const int nt_cur = cur % NT, nt_nxt = more ? nxt % NT : nt_cur;
        const f16* const src_cur = halo_origin(cur);
        const f16* const src_nxt = more ? halo_origin(nxt) : src_cur;          // (no next item: a harmless re-stage)
#pragma unroll 1
        for (int chunk = 0; chunk < cpt; ++chunk) {
            // halo buffer of this phase = chunk & 1 (an even number of phases per item); the next phase's halo goes to the other
            const bool last = chunk + 1 == cpt;
            const f16* const src_next = last ? src_nxt : src_cur + 64 * (chunk + 1);
            const int nt_next = last ? nt_nxt : nt_cur, chunk_next = last ? 0 : chunk + 1;
            const int xb = chunk & 1;
            uint32_t hb = lds_x + (uint32_t)(xb * XBUF_PIECES * 16);
            asm volatile("" : "+v"(hb));            // keep the 9 x 2 operand addresses of a phase out of the loop-carried state
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // everything but the newest group of DMA instructions (issued one K-step ago) has landed: this K-step's
                // weight tile (issued two K-steps ago) and, in a phase's first K-step, the whole halo
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WI + 1) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();           // ... for every wave; ring slot (t + 2) % 3 and the other halo buffer are free
                __builtin_amdgcn_sched_barrier(0);
                if (chunk == 0 && t == 0 && it > 0) nn = __builtin_amdgcn_readfirstlane(tk[(it - 1) & 1]);   // drawn during the previous item
                {
                    if (t + 2 < 9) issue_w(nt_cur, t + 2, chunk, (t + 2) % C128_NSW);
                    else issue_w(nt_next, t + 2 - 9, chunk_next, (t + 2) % C128_NSW);
                    const int i = t < XI ? t : XI - 1;
                    issue_x(src_next, xb ^ 1, i, poff[i]);
                }
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const uint32_t ck = dx < 0 ? ck_m : (dx > 0 ? ck_p : ck_0);
                const uint32_t xt0 = hb + lane_px + (uint32_t)((dy * HW + dx) * 128) + ck;
                const uint32_t xu = xt0 ^ (uint32_t)(((1 + dy) & 1) << 6), xw = xu ^ 64u;   // (an odd halo row swaps the two k-half slots)
                const uint32_t wb = lds_w + (uint32_t)((t % C128_NSW) * WSLOT_PIECES * 16) + lane_w;
                // all 16 operand fragments of the K-step first (one LDS latency), then its 32 MFMAs
                f16x8 xf[2][4], wf[2][4];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int prow = p >> 1, pcol = (p & 1) * 16;
                        const uint32_t ad = (((prow & 1) ^ kk) ? xw : xu) + (uint32_t)((prow * HW + pcol) * 128);
                        xf[kk][p] = C128_LDS_F16X8(ad);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) wf[kk][c] = C128_LDS_F16X8(wb + (uint32_t)(c * 16 * 128) + (kk ? wk1 : wk0));
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][c], xf[kk][p], acc[c][p], 0, 0, 0);
            }
        }
        // one more ticket (for the item after `nn`); slot it & 1 was read by everyone at least one barrier ago
        if (!single && tid == 0) tk[it & 1] = (int)atomicAdd(ticket_ctr, 1u);

        // ---- epilogue of `cur`
        {
            const int pt = cur / NT;
            const int n = pt / tpi, r = pt - n * tpi;
            const int ty = r / tiles_x, tx = r - ty * tiles_x;
            const int cbase = nt_cur * CT + wc * 64;
            f32x4 bv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[c] = *(const f32x4*)(a.bias + g.bias_off + cbase + c * 16 + fk * 4);
            size_t opix[4];
            f16x4 rv[4][4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int prow = p >> 1, pcol = (p & 1) * 16;
                const int oy = ty * 8 + ROWS * wp + prow, ox = tx * TW + pcol + frow;
                opix[p] = ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + g.out_coff + cbase;
                if (RES) {
                    const f16* rp = a.res + ((size_t)(n * a.res_Hp + oy + a.res_P) * a.res_Wp + ox + a.res_P) * a.res_C + g.res_coff + cbase + fk * 4;
#pragma unroll
                    for (int c = 0; c < 4; ++c) rv[p][c] = *(const f16x4*)(rp + c * 16);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t u[4][2];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 v = acc[c][p] + bv[c];
                    if (RES) {
                        const f16x4 rr = rv[p][c];
                        v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3];
                    }
                    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    h = __builtin_elementwise_max(h, lo4);
                    __builtin_memcpy(u[c], &h, 8);
                }
#pragma unroll
                for (int c = 0; c < 4; c += 2) {
                    const auto s0 = __builtin_amdgcn_permlane16_swap(u[c][0], u[c + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(u[c][1], u[c + 1][1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    *(u32x4*)((f16*)a.out + opix[p] + c * 16 + so) = o;
                }
            }
        }
        if (!more) break;
        cur = nxt; nxt = nn;
        ++it;
    }
}

// Eligibility: 3x3, stride 1, dilation 1, cin and cout multiples of 128, 8 x 32 tiles cover the map exactly.
bool conv128_halo_supported(const ConvKArgs& a, int groups) {
    if (groups != 1 || a.cin % 128 || a.cout % 128 || a.ntaps != 9 || a.in_stride != 1 || a.out_scale != 1 || a.in_P < 1) return false;
    if (a.Wm % 32 || (a.HmWm / a.Wm) % 8 || a.M % a.HmWm) return false;
    const int pitch = a.in_Wp * a.in_C;
    for (int t = 0; t < 9; ++t)
        if (a.g[0].tap_off[t] != (t / 3 - 1) * pitch + (t % 3 - 1) * a.in_C) return false;     // row-major taps
    return true;
}

hipError_t launch_conv128_halo(const ConvKArgs& a, int cu_count, unsigned int* ticket_ctr, hipStream_t s) {
    const int total = (a.M / a.HmWm) * (a.Wm >> 5) * ((a.HmWm / a.Wm) >> 3) * (a.cout / 128);
    const int grid = cu_count < total ? cu_count : total;
    if (a.res) hipLaunchKernelGGL(conv128_halo_kernel<1>, dim3(grid), dim3(512), 0, s, a, ticket_ctr, total <= cu_count ? 1 : 0);
    else hipLaunchKernelGGL(conv128_halo_kernel<0>, dim3(grid), dim3(512), 0, s, a, ticket_ctr, total <= cu_count ? 1 : 0);
    return hipGetLastError();
}
