// 3x3 / stride 1 / 128 -> 128 channel convolution (DLA-34 level3's seven block convs, models/nets/dla.py:86-100;
// ResNet-18/34 layer2, models/nets/resnet.py:55-72) as a persistent halo-tile kernel.
//
// The generic 128-pixel kernel (conv_mfma.hip) re-stages a 128-pixel x 64-channel operand tile per (tap, chunk) K-step:
// 64 FLOP per staged byte, and these layers ran at 650-760 TFLOP/s on it, bound by the L2 -> LDS DMA rate.  Here
//   * a workgroup (8 waves) owns an 8 x 32 pixel tile x all 128 output channels; per 64-channel input chunk it stages the
//     (8+2) x (32+2) halo ONCE (43.5 KB, LDS-DMA) and the nine taps read shifted rows of it (layout and bank swizzle of
//     conv64_halo.hip): the pixel operand costs 1/9 of the bytes, 198 FLOP per staged byte overall;
//   * the unit of the pipeline is a PHASE = (tile, chunk): nine K-steps (taps).  While phase p is multiplied out of halo
//     buffer p & 1, the halo of phase p + 1 (the tile's second chunk, or the next tile's first) lands in the other buffer,
//     one DMA instruction per wave and K-step;
//   * weights stream through a 3-slot ring of 16 KB K-step tiles (128 rows x 64 channels, the generic kernel's packed
//     layout for 128-channel tiles, so no second copy of the weights exists): the tile of K-step s + 2 is issued in
//     K-step s.  Every wave issues exactly 3 DMA instructions per K-step (2 weight + 1 halo; halo slots past the end
//     re-stage the last piece), so the wait in front of a K-step is the immediate s_waitcnt vmcnt(3);
//   * one raw s_barrier per K-step; a wave owns 64 pixels x 64 channels (16 accumulator tiles);
//   * tiles are handed out by one atomic ticket counter per op (zeroed at the head of every forward by the runtime);
//   * epilogue: bias (+ residual) + ReLU, v_permlane16_swap pairs two 16-channel MFMA tiles -> 16-byte stores.  (Keeping the
//     previous tile's stores and this tile's residual loads in flight across the first K-steps, with the wait counts
//     raised accordingly, was measured: no change - the residual layers are 12 us slower because they move 63 MB more.)
// K order: chunk-major, tap-minor (fp32 accumulate; the order of the sums differs from the generic kernel's tap-major
// order, results agree to fp32 round-off).
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define C128_HALO_W 34
#define C128_HALO_PIECES (10 * C128_HALO_W * 8)     // 2720 16-byte pieces per 64-channel halo
#define C128_XBUF_PIECES 3072                       // 6 DMA instructions x 512 lanes
#define C128_WSLOT_PIECES 1024                      // 128 rows x 128 B
#define C128_NSW 3
#define C128_DMA16(gptr, lds_byte_addr) \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_byte_addr) : "memory", "m0")
#define C128_LDS_F16X8(byte_addr) (*(const LDS_AS f16x8*)(uintptr_t)(byte_addr))

template <int RES>
__global__ __launch_bounds__(512) void conv128_halo_kernel(const ConvKArgs a, unsigned int* ticket_ctr) {
    __shared__ __attribute__((aligned(128))) f16 lds[(2 * C128_XBUF_PIECES + C128_NSW * C128_WSLOT_PIECES) * 8];
    __shared__ int tk[3];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wp = wave >> 1;                  // 64-channel half, pixel-row pair of the tile
    const int frow = lane & 15, fk = lane >> 4;
    const ConvGroupArgs& g = a.g[0];

    const int Hm = a.HmWm / a.Wm;
    const int tiles_x = a.Wm >> 5, tpi = tiles_x * (Hm >> 3);
    const int total = (a.M / a.HmWm) * tpi;

    f32x4 bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[c] = *(const f32x4*)(a.bias + g.bias_off + wc * 64 + c * 16 + fk * 4);

    // per-thread source offsets (elements, relative to the halo origin of chunk 0) of its six halo pieces
    uint32_t poff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int p = i * 512 + tid;
        p = p < C128_HALO_PIECES ? p : C128_HALO_PIECES - 1;
        const int hq = p >> 3, hcs = p & 7;
        const int hy = hq / C128_HALO_W, hx = hq - hy * C128_HALO_W;
        poff[i] = (uint32_t)((hy * a.in_Wp + hx) * a.in_C + ((hcs ^ ((hx ^ (hy << 2)) & 7)) * 8));
    }
    const uint32_t lds_x = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;
    const uint32_t lds_w = lds_x + (uint32_t)(2 * C128_XBUF_PIECES * 16);
    const f16* const wsrc = a.wgt + g.w_off + tid * 8;       // this thread's first piece of a K-step tile

    if (tid == 0) {
        tk[0] = (int)atomicAdd(ticket_ctr, 1u);
        tk[1] = (int)atomicAdd(ticket_ctr, 1u);
        tk[2] = (int)atomicAdd(ticket_ctr, 1u);
    }
    __syncthreads();
    int cur = __builtin_amdgcn_readfirstlane(tk[0]);
    int nxt = __builtin_amdgcn_readfirstlane(tk[1]);
    int nn = __builtin_amdgcn_readfirstlane(tk[2]);
    if (cur >= total) return;
    __syncthreads();                                // tk[0..1] are reused as the per-tile slots below

    auto halo_origin = [&](int v) -> const f16* {
        const int n = v / tpi, r = v - n * tpi;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        return a.in + ((size_t)(n * a.in_Hp + ty * 8 - 1 + a.in_P) * a.in_Wp + tx * 32 - 1 + a.in_P) * a.in_C + g.in_coff;
    };
    // one halo DMA instruction (pieces i * 512 .. i * 512 + 511) of the 64-channel chunk at `src` into halo buffer xb
    auto issue_x = [&](const f16* src, int xb, int i, uint32_t off) {
        C128_DMA16(src + off, __builtin_amdgcn_readfirstlane(lds_x + (uint32_t)((xb * C128_XBUF_PIECES + i * 512 + wave * 64) * 16)));
    };
    // the K-step tile (tap, chunk) of the weights into ring slot `slot`: 2 DMA instructions per wave
    auto issue_w = [&](int tap, int chunk, int slot) {
        const f16* src = wsrc + (size_t)(tap * 2 + chunk) * (128 * 64);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            C128_DMA16(src + i * 512 * 8, __builtin_amdgcn_readfirstlane(lds_w + (uint32_t)((slot * C128_WSLOT_PIECES + i * 512 + wave * 64) * 16)));
    };

    // LDS byte offsets of this lane's four pixel fragments at tap (0, 0) (see conv64_halo.hip)
    const uint32_t lane_px = (uint32_t)(((2 * wp + 1) * C128_HALO_W + frow + 1) * 128);
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
        for (int i = 0; i < 6; ++i) issue_x(src, 0, i, poff[i]);
        issue_w(0, 0, 0);
        issue_w(1, 0, 1);
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
        const f16* const src_cur = halo_origin(cur);
        const f16* const src_nxt = more ? halo_origin(nxt) : src_cur;          // (no next tile: a harmless re-stage)
#pragma unroll 1
        for (int chunk = 0; chunk < 2; ++chunk) {
            // halo buffer of this phase = chunk (two phases per tile); the next phase's halo goes to the other one
            const f16* const src_next = chunk == 0 ? src_cur + 64 : src_nxt;
            uint32_t hb = lds_x + (uint32_t)(chunk * C128_XBUF_PIECES * 16);
            asm volatile("" : "+v"(hb));            // keep the 9 x 2 operand addresses of a phase out of the loop-carried state
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // everything but the newest group of 3 DMA instructions (issued one K-step ago) has landed: this K-step's
                // weight tile (issued two K-steps ago) and, in a phase's first K-step, the whole halo
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();           // ... for every wave; ring slot (t + 2) % 3 and the other halo buffer are free
                __builtin_amdgcn_sched_barrier(0);
                if (chunk == 0 && t == 0 && it > 0) nn = __builtin_amdgcn_readfirstlane(tk[(it - 1) & 1]);   // drawn during the previous tile
                {
                    const int t2 = t + 2 < 9 ? t + 2 : t + 2 - 9;
                    issue_w(t2, t + 2 < 9 ? chunk : chunk ^ 1, (t + 2) % C128_NSW);
                    const int i = t < 6 ? t : 5;
                    issue_x(src_next, chunk ^ 1, i, poff[i]);
                }
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const uint32_t ck = dx < 0 ? ck_m : (dx > 0 ? ck_p : ck_0);
                const uint32_t xt0 = hb + lane_px + (uint32_t)((dy * C128_HALO_W + dx) * 128) + ck;
                const uint32_t xu = xt0 ^ (uint32_t)(((1 + dy) & 1) << 6), xw = xu ^ 64u;
                const uint32_t wb = lds_w + (uint32_t)((t % C128_NSW) * C128_WSLOT_PIECES * 16) + lane_w;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    f16x8 xf[4], wf[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const uint32_t ad = (((p >> 1) ^ kk) ? xw : xu) + (uint32_t)(((p >> 1) * C128_HALO_W + (p & 1) * 16) * 128);
                        xf[p] = C128_LDS_F16X8(ad);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) wf[c] = C128_LDS_F16X8(wb + (uint32_t)(c * 16 * 128) + (kk ? wk1 : wk0));
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], xf[p], acc[c][p], 0, 0, 0);
                }
            }
        }
        // one more ticket (for the tile after `nn`); slot it & 1 was read by everyone at least one barrier ago
        if (tid == 0) tk[it & 1] = (int)atomicAdd(ticket_ctr, 1u);

        // ---- epilogue of `cur`
        {
            const int n = cur / tpi, r = cur - n * tpi;
            const int ty = r / tiles_x, tx = r - ty * tiles_x;
            size_t opix[4];
            f16x4 rv[4][4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int oy = ty * 8 + 2 * wp + (p >> 1), ox = tx * 32 + (p & 1) * 16 + frow;
                opix[p] = ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + g.out_coff + wc * 64;
                if (RES) {
                    const f16* rp = a.res + ((size_t)(n * a.res_Hp + oy + a.res_P) * a.res_Wp + ox + a.res_P) * a.res_C + g.res_coff + wc * 64 + fk * 4;
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

bool conv128_halo_supported(const ConvKArgs& a, int groups) {
    if (groups != 1 || a.cin != 128 || a.cout != 128 || a.ntaps != 9 || a.in_stride != 1 || a.out_scale != 1 || a.in_P < 1) return false;
    if (a.Wm % 32 || (a.HmWm / a.Wm) % 8 || a.M % a.HmWm) return false;
    const int pitch = a.in_Wp * a.in_C;
    for (int t = 0; t < 9; ++t)
        if (a.g[0].tap_off[t] != (t / 3 - 1) * pitch + (t % 3 - 1) * a.in_C) return false;     // 3x3, dilation 1, row-major taps
    return true;
}

hipError_t launch_conv128_halo(const ConvKArgs& a, int cu_count, unsigned int* ticket_ctr, hipStream_t s) {
    const int total = (a.M / a.HmWm) * (a.Wm >> 5) * ((a.HmWm / a.Wm) >> 3);
    const int grid = cu_count < total ? cu_count : total;
    if (a.res) hipLaunchKernelGGL(conv128_halo_kernel<1>, dim3(grid), dim3(512), 0, s, a, ticket_ctr);
    else hipLaunchKernelGGL(conv128_halo_kernel<0>, dim3(grid), dim3(512), 0, s, a, ticket_ctr);
    return hipGetLastError();
}
