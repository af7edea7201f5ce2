// DLA-34 stem, fused (models/nets/dla.py:259-279):
//   base_layer (7x7, 3 -> 16, BN, ReLU) + level0 (3x3, 16 -> 16, BN, ReLU) [+ level1 (3x3 stride 2, 16 -> 32, BN, ReLU)]
// in one kernel.  Each of these layers is HBM-bound on its own (0.22 + 0.25 + 0.18 ms at bs=32: the two 16-channel
// full-resolution maps are written once and read back once, 2.0 GB of the 2.7 GB the three layers move); fused, the
// intermediate maps live in LDS only and the kernel moves 0.13 GB in + 0.25 GB out.
//
// One 256-thread workgroup per 16 x 32 full-resolution tile (several per CU: ~40 KB of LDS, ~100 registers - the
// latency of the loads is covered by occupancy, no DMA ring needed):
//   1. the window of the NHWC4 fp16 image (8 B per pixel) the tile needs -> LDS;
//   2. base_layer on the halo the next stage needs: register-direct MFMA, K-step = one filter row (7 taps x 4 channels +
//      one zero tap = 32: lane group fk holds the pixels x+2fk, x+2fk+1, which are adjacent in the NHWC4 window -> one
//      16-byte operand).  Two 16-column strips are walked DOWN the rows by a wave: the operand of filter row ky at output
//      row r is the operand of filter row ky - 1 at row r + 1, so one row down keeps 6 of the 7 operands in registers and
//      reads ONE new 16 bytes.  bias + ReLU, fp16, -> LDS as [row][column][16 channels]; pixels outside the image are
//      written as ZERO: they are the next layer's zero padding, not a convolution of padded input;
//   3. level0 from that LDS map: K-step = two taps x 16 channels, bias + ReLU -> global NHWC (two-layer form) or -> LDS
//      (three-layer form, again with zeros outside the image);
//   4. (three-layer form) level1 on the 8 x 16 half-resolution tile from the level0 map in LDS -> global NHWC.
// Weight fragments (7 + 5 + 10 A operands) stay in registers.  Same packing as conv_smallc.hip (cin = 4 / cin = 16 layouts).
#include "common.h"
#include "../../include/rtm3d_hip.h"

// ReLU on the packed fp16 result (2 v_pk_max_f16 instead of 4 canonicalising + 4 clamping v_max_f32 per fragment; same values:
// max commutes with rounding)
#define SF_RELU4(acc) __builtin_elementwise_max((f16x4){(f16)(acc)[0], (f16)(acc)[1], (f16)(acc)[2], (f16)(acc)[3]}, (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f})
// timing-only builds (-DSF_TIMING_NO_LOAD / _NO_S2 / _NO_S3 / _NO_S4 / _NO_W; never defined in the product): drop one stage's work, keep the rest
#ifdef SF_TIMING_NO_S2
#define SF_S2 0
#else
#define SF_S2 1
#endif
#ifdef SF_TIMING_NO_S3
#define SF_S3 0
#else
#define SF_S3 1
#endif
#ifdef SF_TIMING_NO_S4
#define SF_S4 0
#else
#define SF_S4 1
#endif
#define SF_TH 16
#define SF_TW 32

template <int L1>
__global__ __launch_bounds__(256, 4) void stem_fused_kernel(const StemFusedArgs a) {
    // regions, in full-resolution pixels relative to the tile origin (y0, x0):
    //   level0 map: rows -L1 .. 15, columns -L1 .. 31  (the stride-2 3x3 needs one row / column before the tile)
    //   base map  : one more on every side;  image window: 3 more on every side (+ the zero-weighted 8th tap)
    constexpr int L0H = SF_TH + L1, L0W = SF_TW + L1;
    constexpr int BH = L0H + 2, BW = L0W + 2, BO = L1 + 1;           // base map size, offset of its origin (-BO)
    constexpr int XH = BH + 6, XW = BW + 8, XO = BO + 3;             // window size, offset of its origin (-XO)
    constexpr int XT_HALVES = XH * XW * 4, L0_HALVES = L1 ? L0H * L0W * 16 : 0;
    // the window is dead once the base map is complete, the level0 map is born after that: they share storage
    __shared__ __attribute__((aligned(16))) f16 xt[XT_HALVES > L0_HALVES ? XT_HALVES : L0_HALVES];
    __shared__ __attribute__((aligned(16))) f16 bt[BH * BW * 16];
    f16* const l0t = xt;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fk = lane >> 4;
    const int tpi = a.tiles_x * a.tiles_y;
    f16x8 wb[7], wl[5];
#ifdef SF_TIMING_NO_W
#define SF_WLOAD(p) ((f16x8){(f16)lane, (f16)1.f, (f16)lane, (f16)2.f, (f16)lane, (f16)3.f, (f16)lane, (f16)4.f})
#else
#define SF_WLOAD(p) (*(const f16x8*)(p))
#endif
#pragma unroll
    for (int s = 0; s < 7; ++s) wb[s] = SF_WLOAD(a.w_base + (s * 64 + lane) * 8);
#pragma unroll
    for (int s = 0; s < 5; ++s) wl[s] = SF_WLOAD(a.w_l0 + (s * 64 + lane) * 8);
    const f32x4 bb = *(const f32x4*)(a.b_base + fk * 4), bl = *(const f32x4*)(a.b_l0 + fk * 4);
    const int tile = xcd_contiguous_index(blockIdx.x, gridDim.x);   // neighbouring tiles (overlapping image windows) through one L2
    const int n = tile / tpi, r0 = tile - n * tpi;
    const int ty = r0 / a.tiles_x, tx = r0 - ty * a.tiles_x;
    const int y0 = ty * SF_TH, x0 = tx * SF_TW;

    // Tile-uniform edge flags: only tiles that touch the image border pay for the per-pixel inside / outside tests (the
    // 7x7's zero padding, the zero rows / columns the next layer must see); for the ~87 % interior tiles of a 384 x 1280
    // image they are scalar branches not taken.  (Round 3: the kernel was bound by vector-instruction issue - ~1300 VALU
    // instructions per wave and tile against 138 MFMAs per SIMD - so the per-pixel index divisions, bounds tests, bias
    // adds and tap-offset arithmetic went: rows per wave and one lane per column in stage 1, biases as the accumulators'
    // initial values, tap offsets computed once per lane.  The 43 % LDS bank-conflict share of the profile is the 8-byte stores of
    // the two 16-channel maps (16 lanes, 32-byte pixel stride: 4-way); a 16-byte half swap keyed on bit 2 of the pixel index makes
    // them 2-way and keeps the ds_read_b128 operand reads conflict-free, at three more VALU instructions per operand read - measured
    // on one box: 0.309 -> 0.341 ms.  The conflicts are not what the kernel waits for; the swap is not in.
    // Round 4, same-box A/Bs: ReLU as v_pk_max_f16 on the converted pair instead of fmaxf on the fp32 values (which costs a
    // canonicalising v_max per element on top of the clamp: 8 instructions per fragment against 2) 0.300 -> 0.287 ms; two
    // fragments per step in stage 3 (two independent MFMA chains, operand reads of one under the multiplies of the other)
    // 0.288 vs 0.287; both 16-column strips per wave in stage 2 (+28 registers: spills at four waves per SIMD) 0.355; the stage-3
    // loop fully unrolled 0.301 vs 0.300; three instead of four waves per SIMD allowed 0.300.  Time follows the instruction
    // count (~4 cycles per vector instruction + 16 per MFMA of a wave and tile), not latency.
    // Later in round 4 (tools/gpu_variants.sh, one box): stage 3 walked as strips with compile-time row offsets instead of
    // flattened 16-pixel fragments (a division and ~20 vector instructions per fragment) 0.289 -> 0.266 ms; the border tests of
    // stages 2 / 3 as passes of their own (branch-free row loops: the compiler overlaps one row's epilogue with the next row's
    // MFMAs) and buffer loads with scalar row offsets in stage 1: no further change (0.265).  Timing-only builds (SF_TIMING_*):
    // without the image loads 0.246, without stage 2 / 3 / 4 0.199 / 0.214 / 0.214 (stage 2 = its 294 MFMAs per tile at the
    // MFMA rate), without the weight-fragment loads (88 KB per tile, L2 hits) 0.250.  Workgroups that loop over tiles with all
    // weight fragments held in registers: 224 registers, two workgroups per CU: 0.357; at three (168 registers) or four (weights
    // re-loaded per tile) the compiler spills.  What is left besides the MFMAs (554 per tile for 363 useful: zero taps and
    // channels, halo recompute = 0.12 ms at the MFMA rate) is the per-workgroup serial chain: load -> 5 barriers -> store.)
    const bool x_edge = (y0 - XO < 0) | (y0 - XO + XH > a.H) | (x0 - XO < 0) | (x0 - XO + XW > a.W);
    const bool b_edge = (y0 - BO < 0) | (y0 - BO + BH > a.H) | (x0 - BO < 0) | (x0 - BO + BW > a.W);

    // ---- 1. image window: straight from the caller's fp32 NCHW batch (a.x_nchw: the NHWC4 conversion pass and its
    // tensor are skipped; pixels outside the image are the 7x7's zero padding), or from the padded NHWC4 fp16 tensor
    // (filled by rtm3d_preprocess_batch).  In the latter, rows above the padded tensor (first tile row of image 0 in the
    // three-layer form) are clamped to its first row - a zero border row, and every base pixel that would use them lies
    // outside the image anyway.  Wave w takes window rows w, w + 4, ...; lane = window column (XW <= 64).
    static_assert(XW <= 64, "one lane per window column");
    {
        constexpr int NR = (XH + 3) / 4;
        const int gx = x0 - XO + lane;
        if (a.x_nchw) {
            const size_t plane = (size_t)a.H * a.W;
            float v[NR][3];
            if (!x_edge) {
                // interior tiles (~87 %): one buffer descriptor per image, the row / plane offset in a scalar register, the
                // lane's column in the vector offset - no per-row predicates, no 64-bit vector address arithmetic
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x_nchw + (size_t)n * 3 * plane), 0, (int)(3 * plane * 4), 0x00020000);
                const int lane4 = lane * 4, plane4 = (int)(plane * 4);
                const int s0 = ((y0 - XO + wave) * a.W + (x0 - XO)) * 4;
#pragma unroll
                for (int p = 0; p < 3; ++p) v[NR - 1][p] = 0.f;             // the only row a wave may not have (wave + 4 (NR - 1) >= XH)
                __builtin_assume(wave >= 0 && wave < 4);
#ifdef SF_TIMING_NO_LOAD
#pragma unroll
                for (int i = 0; i < NR; ++i) v[i][0] = v[i][1] = v[i][2] = 0.25f;
                if (false) {
#else
                if (lane < XW) {
#endif
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        if (wave + 4 * i < XH) {
                            const int so = s0 + i * 16 * a.W;
#pragma unroll
                            for (int p = 0; p < 3; ++p)
                                v[i][p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane4, so + p * plane4, 0));
                        }
                    }
                }
            } else {
                const float* img = a.x_nchw + (size_t)n * 3 * plane + gx;
                const bool colok = lane < XW && gx >= 0 && gx < a.W;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int r = wave + 4 * i, gy = y0 - XO + r;               // wave-uniform
                    const bool ok = colok && r < XH && gy >= 0 && gy < a.H;
                    v[i][0] = v[i][1] = v[i][2] = 0.f;
                    if (ok) {
                        const float* q = img + (size_t)gy * a.W;
                        v[i][0] = q[0]; v[i][1] = q[plane]; v[i][2] = q[2 * plane];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = wave + 4 * i;
                if (r < XH && lane < XW) *(f16x4*)(xt + (r * XW + lane) * 4) = (f16x4){(f16)v[i][0], (f16)v[i][1], (f16)v[i][2], (f16)0.f};
            }
        } else {
            const f16* img = a.x4 + (size_t)n * a.x_Hp * a.x_Wp * 4 + (ptrdiff_t)(gx + a.x_P) * 4;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = wave + 4 * i;
                int pr = y0 - XO + r + a.x_P;
                pr = pr > 0 ? pr : 0;
                if (r < XH && lane < XW) *(f16x4*)(xt + (r * XW + lane) * 4) = *(const f16x4*)(img + (ptrdiff_t)pr * a.x_Wp * 4);
            }
        }
    }
    __syncthreads();

    // ---- 2. base_layer on the BH x BW halo (accumulators start at the bias)
    // (addresses: one per-lane base per wave, rows are compile-time offsets from it - no per-row address arithmetic)
    auto base_store = [&](const f32x4& acc, f16* dst) { *(f16x4*)dst = SF_RELU4(acc); };
    auto window16p = [&](const f16* xp) -> f16x8 {         // two adjacent window pixels: 16 bytes, 8-byte aligned
        const f16x4 lo = *(const f16x4*)xp, hi = *(const f16x4*)(xp + 4);
        return (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto window16 = [&](int r, int c) -> f16x8 { return window16p(xt + (r * XW + c + 2 * fk) * 4); };   // pixels (r, c + 2fk), (r, c + 2fk + 1)
    {
        // columns 0-31: wave w walks strip w & 1 over rows RA * (w >> 1) .. (the second half is one row shorter when BH is odd)
        constexpr int RA = (BH + 1) / 2;
        const int c = (wave & 1) * 16 + frow, rbase = (wave >> 1) * RA;
        const int nrows = SF_S2 ? ((wave >> 1) ? BH - RA : RA) : 0;
        const f16* const xw = xt + (rbase * XW + c + 2 * fk) * 4;
        f16* const bst = bt + (rbase * BW + c) * 16 + fk * 4;
        f16x8 win[7];
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) win[ky] = window16p(xw + ky * XW * 4);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            if (i < nrows) {
                f32x4 acc = bb;
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ky], win[(i + ky) % 7], acc, 0, 0, 0);
                if (i + 1 < nrows) win[i % 7] = window16p(xw + (i + 7) * XW * 4);     // filter row 6 of the next output row
                base_store(acc, bst + i * BW * 16);
            }
        }
        // columns 32 .. BW-1 (2 or 3 columns, BH rows): ordinary fragments, one per wave
        constexpr int RC = BW - 32, RPIX = RC * BH, RFRAGS = (RPIX + 15) / 16;
        static_assert(RFRAGS <= 4, "remainder fragments must fit the four waves");
        if (SF_S2 && wave < RFRAGS) {
            int q = wave * 16 + frow;
            q = q < RPIX ? q : RPIX - 1;                     // the spare lanes of the last fragment redo its last pixel
            const int r = q / RC, cc = 32 + (q - r * RC);
            f32x4 acc = bb;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ky], window16(r + ky, cc), acc, 0, 0, 0);
            base_store(acc, bt + (r * BW + cc) * 16 + fk * 4);
        }
    }
    __syncthreads();
    // tiles on the image border (~13 %): base pixels outside the image are the next layer's ZERO padding, not a convolution
    // of padded input.  Done as a pass of its own so that the row loop above carries no branch (the compiler then runs one
    // row's conversion and store under the next row's MFMAs).
    if (b_edge) {
        for (int q = tid; q < BH * BW; q += 256) {
            const int r = q / BW, c = q - r * BW;
            const int gy = y0 - BO + r, gx = x0 - BO + c;
            if (!(gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)) {
                *(u32x4*)(bt + q * 16) = (u32x4){0u, 0u, 0u, 0u};
                *(u32x4*)(bt + q * 16 + 8) = (u32x4){0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();
    }

    // K-step s of the two 3x3 layers = taps 2s, 2s + 1 (lane group fk >> 1 picks one) x 16 channels (k half fk & 1);
    // the 10th tap has zero weights, any valid address will do.  Tap offsets in halves, per lane, once.
    int tap_y[5], tap_x[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        int t = 2 * s + (fk >> 1);
        t = t < 9 ? t : 8;
        tap_y[s] = t / 3; tap_x[s] = t - tap_y[s] * 3;
    }

    // ---- 3. level0 on the L0H x L0W map, walked like the base layer: wave w takes 16-column strip w & 1 over the upper
    // (w < 2) or lower half of the rows - every operand read and LDS store is a per-lane base plus a compile-time row
    // offset - and, in the three-layer form, the lower-half waves (one row fewer) share the 33rd column as two fragments
    // (lane = row).  Round 4: the former fragment = 16 consecutive pixels of the flattened map cost a division and ~20
    // vector instructions per fragment beside its 5 MFMAs.
    {
        constexpr int RA0 = (L0H + 1) / 2;
        const int half = wave >> 1, rbase = half * RA0, col = (wave & 1) * 16 + frow;
        const f16* const bp = bt + (rbase * BW + col) * 16;
        int toff[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) toff[s] = (tap_y[s] * BW + tap_x[s]) * 16 + (fk & 1) * 8;
        const bool tl_edge = L1 && ((y0 == 0) | (x0 == 0));
        auto frag = [&](const f16* src) -> f16x4 {
            f32x4 acc = bl;
#pragma unroll
            for (int s = 0; s < 5; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], *(const f16x8*)(src + toff[s]), acc, 0, 0, 0);
            return SF_RELU4(acc);
        };
        f16* const lst = l0t + (rbase * L0W + col) * 16 + fk * 4;
        f16* gst = L1 ? nullptr : a.out + ((size_t)(n * a.o_Hp + y0 + rbase + a.o_P) * a.o_Wp + x0 + col + a.o_P) * a.o_C + a.o_coff + fk * 4;
        const int grow = a.o_Wp * a.o_C;
#pragma unroll
        for (int i = 0; i < RA0; ++i) {
            if (SF_S3 && (i < L0H - RA0 || half == 0)) {                     // the lower half is one row shorter when L0H is odd
                f16x4 h = frag(bp + i * BW * 16);
                if (L1) {
                    *(f16x4*)(lst + i * L0W * 16) = h;
                } else {
                    *(f16x4*)gst = h;
                    gst += grow;
                }
            }
        }
        if (SF_S3 && L1 && half == 1) {
            // column 32, rows (wave & 1) * 16 + lane row: 17 pixels in two fragments; the spare lanes redo row 16
            constexpr int CR = L0W - 1;
            const int rr = (wave & 1) * 16 + frow;
            const bool live = rr < L0H;
            const int row = live ? rr : L0H - 1;
            f16x4 h = frag(bt + (row * BW + CR) * 16);
            if (live) *(f16x4*)(l0t + (row * L0W + CR) * 16 + fk * 4) = h;
        }
        if (!L1) return;
        __syncthreads();
        // first tile row / column of the image: level1's zero padding above / left of it (a pass of its own, as for the base map)
        if (tl_edge) {
            if (y0 == 0 && tid < L0W) {
                *(u32x4*)(l0t + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
                *(u32x4*)(l0t + tid * 16 + 8) = (u32x4){0u, 0u, 0u, 0u};
            }
            if (x0 == 0 && tid >= 64 && tid < 64 + L0H) {
                *(u32x4*)(l0t + (tid - 64) * L0W * 16) = (u32x4){0u, 0u, 0u, 0u};
                *(u32x4*)(l0t + (tid - 64) * L0W * 16 + 8) = (u32x4){0u, 0u, 0u, 0u};
            }
            __syncthreads();
        }
    }

    // ---- 4. level1 (3x3, stride 2, 16 -> 32) on the 8 x 16 half-resolution tile: fragment = output row, 16 columns;
    // output (oy, ox) reads level0-map pixels (2 oy + dy, 2 ox + dx), dy, dx in 0..2 (the map starts one pixel before the tile)
    {
        f16x8 w1[2][5];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s = 0; s < 5; ++s) w1[c][s] = SF_WLOAD(a.w_l1 + ((c * 5 + s) * 64 + lane) * 8);
        f32x4 b1[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) b1[c] = *(const f32x4*)(a.b_l1 + c * 16 + fk * 4);
        const int so = (fk & 1) * 16 + (fk >> 1) * 8;
        int t1off[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) t1off[s] = (tap_y[s] * L0W + tap_x[s] + 2 * frow) * 16 + (fk & 1) * 8;
#pragma unroll
        for (int j = 0; j < (SF_S4 ? 2 : 0); ++j) {
            const int oy = wave * 2 + j;
            const f16* lp = l0t + (2 * oy * L0W) * 16;
            f32x4 acc[2] = {b1[0], b1[1]};
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const f16x8 xf = *(const f16x8*)(lp + t1off[s]);
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[c][s], xf, acc[c], 0, 0, 0);
            }
            uint32_t u[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f16x4 h = SF_RELU4(acc[c]);
                __builtin_memcpy(u[c], &h, 8);
            }
            // rows (16-lane groups) 1,3 of the c = 0 registers <-> rows 0,2 of the c = 1 registers: 8 consecutive channels per lane
            const auto s0 = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            f16* op = a.out + ((size_t)(n * a.o_Hp + (y0 >> 1) + oy + a.o_P) * a.o_Wp + (x0 >> 1) + frow + a.o_P) * a.o_C + a.o_coff + so;
            *(u32x4*)op = o;
        }
    }
}

hipError_t launch_stem_fused(const StemFusedArgs& a, hipStream_t s) {
    const dim3 grid(a.B * a.tiles_x * a.tiles_y), block(256);
    if (a.w_l1) hipLaunchKernelGGL(stem_fused_kernel<1>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(stem_fused_kernel<0>, grid, block, 0, s, a);
    return hipGetLastError();
}
