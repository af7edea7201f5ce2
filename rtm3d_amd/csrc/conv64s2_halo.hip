// 3x3 / STRIDE 2 / 64 -> 128 channel convolution (DLA-34 level3's entry conv tree1.tree1.conv1, models/nets/dla.py:86-91 with
// stride 2; ResNet-18/34 layer2.0.conv1, models/nets/resnet.py:55-60) as a persistent halo-tile kernel with the whole filter
// bank in registers - conv64_halo.hip for a stride-2 layer.
//
// On the generic 128-pixel kernel this layer stages, per 128 output pixels, nine 16 KB pixel tiles (every input pixel 2.25
// times, gathered at stride 2) and nine 16 KB weight tiles through L2 -> LDS for 1152 MFMAs: 0.071-0.073 ms at bs=32 (495-512
// TFLOP/s), bound by what one CU's memory path moves, not by HBM or the matrix pipe.  Here
//   * a workgroup (8 waves) owns a 4 x 32 OUTPUT tile = 128 pixels x 128 channels and stages the (2*4+1) x (2*32+1) input halo
//     ONCE (74.9 KB by LDS-DMA, row by row, double-buffered: 2 x 74 KB of the 160 KB); the nine taps read rows / columns 2y + dy, 2x + dx of it;
//   * a wave (wc = wave & 3, wh = wave >> 2) owns 32 output channels x 64 pixels (output rows 2wh, 2wh+1) and keeps its
//     9 x 2 x 2 weight fragments (144 VGPRs) for the whole launch - no weight traffic after the first 36 loads;
//   * LDS bank swizzle for the stride-2 reads: lane frow reads halo column 2 frow + dx, so the 16-byte slot of a pixel's
//     k-chunk is XORed with (column >> 1) & 7 (with the stride-1 key, column & 7, eight consecutive lanes would hit four slots
//     twice); applied on the SOURCE side of the DMA, which copies linearly;
//   * tickets, epilogue (bias + ReLU, v_permlane16_swap -> 16-byte stores) as conv64_halo.hip.
// Measured (same box, bs=32, 1920 tiles): 0.051-0.054 ms against 0.071-0.074 on the generic kernel.  Over the batch size the launch
// time is 14.7 us + 4.9 us per tile and workgroup (conv64_halo.hip on its 256-pixel tiles: 15.3 us + 1.8 us): the per-tile cost is
// 2.3x the 2.1 us of MFMA time.  What it is NOT (each tried on one box, no change): the halo DMA (timing-only build without it: -1 us
// per tile), the MFMAs (without them: -1 us), LDS bank conflicts (48 % of the LDS cycles before the pixel-pair swap below, 0 after),
// operand prefetch distance (one or two K-steps ahead), the place of the ticket draw relative to the stores.
// K order: tap-major, 64 channels per tap as two 32-deep MFMAs (the generic kernel's order; fp32 accumulate).
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define S2_TH 4
#define S2_TW 32
#define S2_HALO_H (2 * S2_TH + 1)
#define S2_HALO_W (2 * S2_TW + 1)
#define S2_HALO_PIECES (S2_HALO_H * S2_HALO_W * 8)          // 4680 16-byte pieces per halo
#define S2_BUF_PIECES (((S2_HALO_PIECES + 63) / 64) * 64)   // 4736
// LDS-DMA with a wave-uniform 64-bit base in SGPRs and a 32-bit byte offset per lane (no 64-bit address registers)
#define S2_DMA16 RT_DMA16_SBASE                     // common.h: the one LDS-DMA definition
#define S2_LDS_F16X8(byte_addr) (*(const LDS_AS f16x8*)(uintptr_t)(byte_addr))

__global__ __launch_bounds__(512) void conv64s2_halo_kernel(const ConvKArgs a, unsigned int* ticket_ctr, const int single) {
    __shared__ __attribute__((aligned(128))) f16 lds[2 * S2_BUF_PIECES * 8];
    __shared__ __attribute__((aligned(16))) float sbias[128];
    __shared__ int tk[3];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 3, wh = wave >> 2;                  // 32-channel group, pair of output rows
    const int frow = lane & 15, fk = lane >> 4;
    const ConvGroupArgs& g = a.g[0];

    const int Hm = a.HmWm / a.Wm;
    const int tiles_x = a.Wm / S2_TW, tpi = tiles_x * (Hm / S2_TH);
    const int total = (a.M / a.HmWm) * tpi;

    // the filter bank of this wave's 32 output channels: [tap][k half][16-channel tile of 8], MFMA A fragments
    f16x8 wreg[9][2][2];
    {
        const f16* wb = a.wgt + g.w_off;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    wreg[t][kk][c] = *(const f16x8*)(wb + ((size_t)(((t * 2 + kk) * 8 + wc * 2 + c) * 64 + lane)) * 8);
    }
    if (tid < 128) sbias[tid] = a.bias[g.bias_off + tid];      // (the accumulators start at the bias; read per tile from LDS: 8 registers less)

    // DMA plan (registers are what this kernel is short of: 144 hold the filter bank).  Halo row hy = instruction hy: thread tid
    // stages piece (column tid >> 3, slot tid & 7) of the row's first 64 pixels - 512 consecutive pieces in LDS, one source
    // offset per row parity (the swizzle key ((hx >> 1) ^ (hy << 2)) & 7 depends on hy through its parity only) plus hy row
    // pitches added on the scalar side; the 65th column of row hy is staged by lanes 0-7 of wave hy & 7 (wave 0 also row 8).
    // LDS POSITION pl of a row holds halo pixel hx = pl ^ ((pl >> 1) & 1) (pixels 2k, 2k+1 swapped for odd k): a wave reads
    // same-parity pixels (2 frow + dx), which at a 128-byte pixel pitch would all sit in one half of the 256-byte bank row -
    // 2-way conflicts on every ds_read_b128 (measured: SQ_LDS_BANK_CONFLICT 48 % of the LDS cycles); with the swap consecutive
    // lanes alternate between the halves
    // Two input layouts: the ordinary padded NHWC map at full resolution, or (a.in_s2d) its SPACE-TO-DEPTH copy - full-resolution
    // pixel (y, x) at half-resolution pixel (y >> 1, x >> 1), channel slice ((y & 1) * 2 + (x & 1)) * 64 of a.in - which is what the
    // neck reads (plan.py: _neck_up_folds); reading it here too lets the producer skip the ordinary copy.  Halo column hx / row hy
    // = full-resolution column 2 ox0 - 1 + hx / row 2 oy0 - 1 + hy; element offsets relative to the tile's base pixel
    // (ordinary: (2 oy0 - 1, 2 ox0 - 1); s2d: half-resolution pixel (oy0 - 1, ox0 - 1)):
    const bool s2 = a.in_s2d != 0;
    const uint32_t inC = (uint32_t)a.in_C, rp = (uint32_t)(a.in_Wp * a.in_C);
    auto pxo = [&](uint32_t hx) -> uint32_t { return s2 ? ((hx + 1) >> 1) * inC + ((hx + 1) & 1) * 64 : hx * inC; };
    const uint32_t pl = (uint32_t)(tid >> 3), hcs = (uint32_t)(tid & 7);
    const uint32_t hxl = pl ^ ((pl >> 1) & 1);
    const uint32_t off_e = (pxo(hxl) + ((hcs ^ ((hxl >> 1) & 7)) * 8)) * 2;            // bytes
    const uint32_t off_o = (pxo(hxl) + ((hcs ^ (((hxl >> 1) ^ 4) & 7)) * 8)) * 2;
    // column 64 of row hy: ONE full 64-lane instruction on LDS pieces (hy * 65 + 64) * 8 .. + 63 = pixel (hy, 64) and, behind it,
    // pixels 0 .. 6 of row hy + 1 - lanes 8 .. 63 stage exactly what belongs there (the row instruction writes the same bytes:
    // no divergent branch around the asm, which cost SGPR -> scratch spills).  Row 8, the last: the lanes behind the halo re-read
    // pixel (8, 64) (the buffer ends in a 56-piece dead zone).
    const uint32_t cl = (uint32_t)(lane >> 3), sl = (uint32_t)(lane & 7);
    const uint32_t nxp = cl - 1, nxc = nxp ^ ((nxp >> 1) & 1);                                         // position / pixel in row hy + 1 (lanes >= 8)
    // offset of row hy + 1 relative to row hy: one row pitch (ordinary); s2d: rows alternate between the two row phases of one
    // half-resolution row (hy odd -> even: + 128 channels) and step to the next half-resolution row (hy even -> odd: + pitch - 128)
    const uint32_t drow_e = s2 ? rp - 128 : rp, drow_o = s2 ? 128u : rp;
    const uint32_t off_c_e = (cl == 0 ? pxo(64) + sl * 8                                               // hy even: key of (hy, 64) = 0, of row hy + 1: odd
                                      : drow_e + pxo(nxc) + ((sl ^ (((nxc >> 1) ^ 4) & 7)) * 8)) * 2;
    const uint32_t off_c_o = (cl == 0 ? pxo(64) + ((sl ^ 4) * 8)                                       // hy odd
                                      : drow_o + pxo(nxc) + ((sl ^ ((nxc >> 1) & 7)) * 8)) * 2;
    // (hy = 8, even, every lane: (pxo(64) + slot * 8) * 2, recomputed at its one use per tile: the kernel has no register to spare)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;

    // tickets: as conv64_halo.hip
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

    auto halo_origin = [&](int v) -> size_t {       // the tile's base pixel (see above)
        const int n = v / tpi, r = v - n * tpi;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        if (s2) return ((size_t)(n * a.in_Hp + ty * S2_TH - 1 + a.in_P) * a.in_Wp + tx * S2_TW - 1 + a.in_P) * a.in_C + g.in_coff;
        return ((size_t)(n * a.in_Hp + 2 * ty * S2_TH - 1 + a.in_P) * a.in_Wp + 2 * tx * S2_TW - 1 + a.in_P) * a.in_C + g.in_coff;
    };
    // element offset of halo row hy relative to the base pixel
    auto rowo = [&](int hy) -> size_t { return s2 ? (size_t)((hy + 1) >> 1) * rp + (size_t)(((hy + 1) & 1) * 128) : (size_t)hy * rp; };
    auto stage = [&](int v, int par) {
        const f16* src = a.in + halo_origin(v);
        const uint32_t lb = lds_base + (uint32_t)(par * S2_BUF_PIECES * 16);
#pragma unroll
        for (int hy = 0; hy < S2_HALO_H; ++hy)          // columns 0 .. 63 of row hy: LDS pieces (hy * 65) * 8 + tid
            S2_DMA16((hy & 1) ? off_o : off_e, src + rowo(hy), __builtin_amdgcn_readfirstlane(lb + (uint32_t)((hy * S2_HALO_W * 8 + wave * 64) * 16)));
        // column 64 of row `wave` (wave 0: also row 8)
        S2_DMA16((wave & 1) ? off_c_o : off_c_e, src + rowo(wave), __builtin_amdgcn_readfirstlane(lb + (uint32_t)(((wave * S2_HALO_W + 64) * 8) * 16)));
        if (wave == 0) {
            uint32_t l2 = (uint32_t)lane;
            asm volatile("" : "+v"(l2));            // not loop-invariant as far as the compiler can tell
            S2_DMA16((pxo(64) + (l2 & 7) * 8) * 2, src + rowo(8), __builtin_amdgcn_readfirstlane(lb + (uint32_t)(((8 * S2_HALO_W + 64) * 8) * 16)));
        }
    };

    // LDS byte offset of this lane's fragment p = 0 pixel at tap (0, 0): halo row 2 * (2 wh), halo column 2 * frow
    const uint32_t lane_px = (uint32_t)(((4 * wh) * S2_HALO_W + 2 * frow) * 128);
    // 16-byte slot of k-chunk (kk * 4 + fk): XOR key = ((hx >> 1) ^ (hy << 2)) & 7 with hx >> 1 = column + (dx == 2) (+ 16 for the
    // second fragment: no change mod 8), (hy << 2) & 7 = (dy & 1) << 2
    const uint32_t key0 = (uint32_t)(frow & 7), key1 = (uint32_t)((frow + 1) & 7);
    // position of halo pixel hx = 2 * column + dx: hx ^ (((hx >> 1)) & 1) = hx + 1 (dx = 0), hx - 1 (dx = 1) when the column is odd,
    // hx + 1 (dx = 2) when it is even (the 16-column offset of the second fragment does not change the parity)
    const uint32_t adj = (uint32_t)(frow & 1) * 128;
    const f16 lo = a.relu ? (f16)0.f : (f16)(-__builtin_inff());
    const f16x4 lo4 = {lo, lo, lo, lo};

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

        // The wave's two output rows one after the other: 8 accumulator + 8 operand registers per pass instead of 16 + 16 (with all
        // four fragments in flight the kernel sat at 255 registers and the compiler serialised read -> wait -> 2 MFMAs: 6.9 us per
        // tile against 2.3 us of MFMA time); the filter fragments are register-resident, so the second pass re-reads nothing
        // but its own pixels.
        uint32_t hb = lds_base + (uint32_t)(par * S2_BUF_PIECES * 16) + lane_px;
        asm volatile("" : "+v"(hb));                // keep the operand addresses out of the loop-carried state (they would be hoisted for both buffers)
        const int n = cur / tpi, r = cur - n * tpi;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        uint32_t fk2 = (uint32_t)fk;
        asm volatile("" : "+v"(fk2));               // (likewise: the store offset is recomputed per tile)
        const int so = (int)((fk2 & 1) * 16 + (fk2 >> 1) * 8);
        u32x4 res16[2][2];                          // both rows' packed results: all four stores go out behind the ticket draw
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            f32x4 acc[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 bv = *(const f32x4*)(sbias + wc * 32 + c * 16 + fk * 4);
                acc[c][0] = bv; acc[c][1] = bv;
            }
            // 18 K-steps (tap, k half), software-pipelined by hand: the two operand fragments of step s + 1 are requested before the
            // four MFMAs of step s (left to itself the compiler reads, waits, multiplies)
            auto frag_addr = [&](int st, int q) -> uint32_t {
                const int t = st >> 1, kk = st & 1, dy = t / 3, dx = t % 3;
                const uint32_t key = (dx == 2 ? key1 : key0) ^ (uint32_t)((dy & 1) << 2);
                const uint32_t swp = dx == 0 ? adj : (dx == 1 ? 0u - adj : 128u - adj);
                return hb + (uint32_t)(((dy + 2 * ph) * S2_HALO_W + dx) * 128) + (uint32_t)(q * 32 * 128) + swp + ((((uint32_t)(kk * 4 + fk)) ^ key) << 4);
            };
            // (operands two steps ahead: one step of four MFMAs - 64 cycles - does not cover the LDS latency under load)
            f16x8 x0[2] = {S2_LDS_F16X8(frag_addr(0, 0)), S2_LDS_F16X8(frag_addr(0, 1))};
            f16x8 x1[2] = {S2_LDS_F16X8(frag_addr(1, 0)), S2_LDS_F16X8(frag_addr(1, 1))};
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                f16x8 x2[2] = {x1[0], x1[1]};
                if (st + 2 < 18) { x2[0] = S2_LDS_F16X8(frag_addr(st + 2, 0)); x2[1] = S2_LDS_F16X8(frag_addr(st + 2, 1)); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[st >> 1][st & 1][c], x0[q], acc[c][q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                x0[0] = x1[0]; x0[1] = x1[1]; x1[0] = x2[0]; x1[1] = x2[1];
            }
            // ---- this row's results: bias is in, ReLU, fp16, 8 consecutive channels per lane
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint32_t u[2][2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x4 v = acc[c][q];
                    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    h = __builtin_elementwise_max(h, lo4);
                    __builtin_memcpy(u[c], &h, 8);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
                res16[ph][q] = (u32x4){s0[0], s1[0], s0[1], s1[1]};
            }
        }
        // one more ticket (for the tile after `nn`); slot it & 1 was read by everyone two barriers ago.  IN FRONT of the stores: the
        // compiler waits for the returning atomic with vmcnt(0), which here means this wave's halo DMA (long landed) - behind the
        // stores it would be their acknowledgement, a memory round trip per tile with every other wave waiting at the barrier
        // (that order cost 6.8 us per tile instead of ~3.5)
        if (!single && tid == 0) tk[it & 1] = (int)atomicAdd(ticket_ctr, 1u);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int oy = ty * S2_TH + 2 * wh + ph, ox = tx * S2_TW + q * 16 + frow;
                const size_t opix = ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + g.out_coff + wc * 32;
                *(u32x4*)((f16*)a.out + opix + so) = res16[ph][q];
            }
        if (!more) break;
        // the next tile's halo (issued before this tile's stores) must have landed; the 4 stores may stay in flight
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        cur = nxt; nxt = nn;
        par ^= 1;
        ++it;
    }
}

bool conv64s2_halo_supported(const ConvKArgs& a, int groups) {
    if (groups != 1 || a.cin != 64 || a.cout != 128 || a.ntaps != 9 || a.in_stride != 2 || a.out_scale != 1 || a.in_P < 1 || a.res) return false;
    if (a.Wm % S2_TW || (a.HmWm / a.Wm) % S2_TH || a.M % a.HmWm) return false;
    const int pitch = a.in_Wp * a.in_C;
    for (int t = 0; t < 9; ++t)
        if (a.g[0].tap_off[t] != (t / 3 - 1) * pitch + (t % 3 - 1) * a.in_C) return false;     // 3x3, dilation 1, row-major taps
    return true;
}

hipError_t launch_conv64s2_halo(const ConvKArgs& a, int cu_count, unsigned int* ticket_ctr, hipStream_t s) {
    const int total = (a.M / a.HmWm) * (a.Wm / S2_TW) * ((a.HmWm / a.Wm) / S2_TH);
    const int grid = cu_count < total ? cu_count : total;
    hipLaunchKernelGGL(conv64s2_halo_kernel, dim3(grid), dim3(512), 0, s, a, ticket_ctr, total <= cu_count ? 1 : 0);
    return hipGetLastError();
}
