// Halo-tile form of the persistent 256x256 implicit-GEMM convolution for DILATED 3x3 layers (the dilation-6 head conv, 256 -> 4 x 256:
// models/nets/header.py:12-16) - the schedule of conv_mfma256_halo.hip on a tile that lives on the ROW SUB-LATTICE of the dilation.
//
// The generic persistent kernel (conv_mfma256.hip) stages a 256-pixel x 64-channel operand half-tile pair per (tap, chunk): nine
// 32 KB copies of nearly the same pixels per chunk, and the rows a tile needs for dy = +-6 are the dy = 0 rows of tiles that ran a
// whole K loop earlier (measured, round 5: 6.3 GB of fabric traffic per launch against 2.5 GB of algorithmic bytes).  A halo over a
// dense 8 x 32 tile would be (8 + 12) x (32 + 12) pixels = 110 KB per chunk.  Here the output tile is 8 rows of ONE RESIDUE CLASS
// modulo D (rows y0, y0 + D, ..., y0 + 7 D) x 32 consecutive columns: its taps (dy, dx) in {-D, 0, D}^2 touch 10 input rows at a
// pitch of D rows x (32 + 2 D) columns - 10 x 44 pixels x 64 channels = 55 KB per chunk for D = 6, staged ONCE per chunk.
//
// Two such halos and the 64 KB weight ring do not fit the LDS (112.6 + 64 + 4 KB of bias > 160 KB), so the halo rows live in a RING
// of 16 row slots (88 KB): row h of chunk number c (counted across tiles) sits in slot (10 c + h) & 15.  With the taps in dy-major
// order (the packing order of the weights) rows die early - row 0 after tap 2, row 1 after tap 5 - and that is exactly what the next
// chunk's rows 6 and 7 need; its rows 0 .. 5 go to the six slots the current chunk never uses, its rows 8 and 9 are staged during
// its own first tap (first read at taps 3 and 6).  One halo row = (32 + 2 D) pixels x 8 pieces = 352 pieces = 8 waves x 44 lanes:
// one DMA instruction per wave, as in the halo kernel, two per K-tile (phases 2 and 3):
//   tap     0          1          2          3          4        5       6       7        8
//   P2 / P3 c.8 / c.9  n.0 / n.1  n.2 / n.3  n.4 / n.5  n.6 / =  = / =   = / =   n.7 / =  = / =        ("=": the last row again)
// so every K-tile issues the halo kernel's 0 + 2 + 2 + 2 DMA instructions and the counted waits are its immediates (vmcnt(6), 22 in a
// tile's first K-tile).  Everything new is issued at least one whole K-tile before its first read (n.7: tap 7 -> read in phase 3 of
// the next chunk's tap 0) and at least two K-tiles after the last read of the row it replaces.
// Weights, K order (chunk-major over the packed K-tiles tap * cpt + chunk), tickets, bias-in-LDS, 16-byte swapped stores: the halo
// kernel's.  The operand addresses of a K-tile are four wave-uniform row-slot addresses (the ring wraps) added to one per-lane
// column offset, computed a K-tile ahead in phase 4's load segment.
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define LT_HALF_ELEMS (128 * 64)              // one weight half-tile: 128 rows x 64 halves = 16 KB
#define LT_WRING_ELEMS (4 * LT_HALF_ELEMS)    // two K-tiles x (WA, WB)
#define LT_RING 16                            // halo row slots
#define LT_MAX_BIAS 1024


#define LT_DMA16_SBASE_LANES RT_DMA16_SBASE_LANES
#define LT_DMA16_SBASE RT_DMA16_SBASE
#define LT_LDS_F16X8(byte_addr) (*(const LDS_AS f16x8*)(uintptr_t)(byte_addr))

template <int D>
__global__ __launch_bounds__(512) void conv_mfma256_lattice_kernel(const ConvKArgs a, const int groups, const int nbias, unsigned int* tile_ctr) {
    constexpr int HW = 32 + 2 * D;                 // halo row: pixels
    constexpr int ROW_PIECES = HW * 8;             // 16-byte pieces
    constexpr int ROW_B = ROW_PIECES * 16;         // bytes (a multiple of 128: the k-half slot bit of an address survives adding it)
    static_assert(HW <= 64, "one DMA lane per piece of a wave's share of a halo row");
    static_assert((LT_WRING_ELEMS + LT_RING * ROW_PIECES * 8) * 2 + (LT_MAX_BIAS + 4) * 4 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(128))) f16 lds[LT_WRING_ELEMS + LT_RING * ROW_PIECES * 8];
    __shared__ __attribute__((aligned(16))) float lds_bias[LT_MAX_BIAS + 4];       // + two ticket words
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wc = wave >> 1;
    const int T = a.ksteps, CPT = a.cpt;
    for (int i = tid; i < nbias; i += 512) lds_bias[i] = a.bias[i];

    // tile list (as conv_mfma256_persistent_kernel): position v -> (group, pixel tile, channel tile)
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
    const int chunk = (a.MT + 7) >> 3;
    int mt_here = a.MT - xcd * chunk;
    mt_here = mt_here < 0 ? 0 : (mt_here > chunk ? chunk : mt_here);
    const int jbs = mt_here * a.NT;
    const int vtotal = jbs * groups;
    int* const lds_ticket = (int*)(lds_bias + LT_MAX_BIAS);
    const int last_draw = vtotal + per_xcd - 1;
    // (the second ticket is drawn behind the prologue's wait: conv_mfma256_persistent_kernel)
    if (tid == 0) {
        const int t0 = (int)atomicAdd(&tile_ctr[xcd], 1u);
        if (t0 == last_draw) tile_ctr[xcd] = 0u;
        lds_ticket[0] = t0;
    }
    __syncthreads();
    const int v = __builtin_amdgcn_readfirstlane(lds_ticket[0]);
    if (v >= vtotal) return;

    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;
    const uint32_t ring_b = lds_base + (uint32_t)LT_WRING_ELEMS * 2u;      // byte address of row slot 0
    const int Hm = a.HmWm / a.Wm;
    // tiles of an image: (residue class of the row modulo D) x (blocks of 8 lattice rows) x (blocks of 32 columns); column blocks
    // run fastest, then the row blocks of a class (neighbours share halo rows through L2)
    const int tiles_x = a.Wm >> 5, RB = (Hm / D) >> 3, tpi = tiles_x * RB * D;

    // a lane's source offset inside a halo row: piece rp = wave * HW + lane of the row's HW x 8, i.e. pixel rp >> 3, 16-byte chunk
    // rp & 7, which lands in LDS slot chunk ^ (pixel & 7) of the pixel's 128 bytes (the DMA writes pieces linearly: the swizzle is
    // in the SOURCE order)
    constexpr unsigned long long xlanes = (1ull << HW) - 1;
    uint32_t roff;
    {
        const int rp = wave * HW + (lane < HW ? lane : HW - 1);
        const int rhx = rp >> 3, rhcs = rp & 7;
        roff = (uint32_t)(rhx * a.in_C + ((rhcs ^ (rhx & 7)) * 8)) * 2u;
    }
    // tile descriptors (current / next): halo origin in the input tensor, weight base, indices
    const char *xb_c, *xb_n;                // halo pixel (0,0), channel 0 of the group slice
    const f16 *wb_c, *wb_n;
    int gi_c, gi_n, nt_c, nt_n, n_c, n_n, y0_c, y0_n, tx_c, tx_n;
    auto locate = [&](int vv, const char*& xb, const f16*& wb, int& gi, int& nt, int& n, int& y0, int& tx) {
        gi = vv / jbs;
        const int jb = vv - gi * jbs;
        // Channel tiles in PAIRS: the two tiles of a pair on adjacent tickets, the XCD's whole pixel list per pair.  Round 6, fabric fetch
        // per launch (FETCH_SIZE x 2, tools/gpu_pmc_fetch_variants.sh) / time: all four channel tiles adjacent (4.7 MB of weights per
        // round through a 4 MB L2) 2.95 GB / 3.32-3.34 ms; pairs 2.41 GB / 3.31-3.39; one tile per sweep (pixel rows fetched four
        // times) 3.06 GB / 3.32-3.45.  The time does not move (power cap); the pairs move least.
        int q;
        if (a.NT % 2 == 0) {
            const int per = mt_here * 2;
            const int grp = jb / per, rem = jb - grp * per;
            q = rem >> 1;
            nt = grp * 2 + (rem & 1);
        } else { q = jb / a.NT; nt = jb - q * a.NT; }
        const int mt = xcd * chunk + q;
        n = mt / tpi;
        const int r = mt - n * tpi;
        const int ty = r / tiles_x;
        tx = r - ty * tiles_x;
        const int res = ty / RB, rb = ty - res * RB;
        y0 = res + rb * 8 * D;                                  // first output row of the tile; the others follow at a pitch of D
        const ConvGroupArgs& g = a.g[gi];
        xb = (const char*)(a.in + ((size_t)(n * a.in_Hp + y0 - D + a.in_P) * a.in_Wp + tx * 32 - D + a.in_P) * a.in_C + g.in_coff);
        wb = a.wgt + g.w_off + (size_t)nt * T * (256 * 64);
    };
    bool live_n = false;
    locate(v, xb_c, wb_c, gi_c, nt_c, n_c, y0_c, tx_c);
    xb_n = xb_c; wb_n = wb_c; gi_n = gi_c; nt_n = nt_c; n_n = n_c; y0_n = y0_c; tx_n = tx_c;

    const int pitch_b = D * a.in_Wp * a.in_C * 2;       // one halo row down (D image rows), bytes

    // weight half-tile (0 = WA, 1 = WB) of the packed K-tile at wk into ring buffer par
    const uint32_t wvoff = (uint32_t)tid * 16u;
    auto stage_w1 = [&](int half, int i, const f16* wk, int par) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((par * 2 + half) * LT_HALF_ELEMS + (i * 512 + wave * 64) * 8) * 2u);
        LT_DMA16_SBASE(wvoff, wk + half * LT_HALF_ELEMS + i * 512 * 8, dst);
    };
    auto stage_w = [&](int half, const f16* wk, int par) {
#pragma unroll
        for (int i = 0; i < 2; ++i) stage_w1(half, i, wk, par);
    };
    // halo row r of the chunk whose row 0 is at `origin` (bytes) and whose ring base is `rbase`
    // (Round 6, same box: a repeated row slot - "=" in the table - cut down to ONE 16-byte piece through the lane mask, 45 KB less
    // staging per chunk: heads.conv_d6 3.35-3.39 ms against 3.31-3.33 with whole rows again; the same in the halo kernel: no change.)
    auto stage_row = [&](const char* origin, int r, int rbase) {
        const char* src = origin + r * pitch_b;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_b + (uint32_t)(((rbase + r) & (LT_RING - 1)) * ROW_B + wave * HW * 16));
        LT_DMA16_SBASE_LANES(roff, src, dst, xlanes);
    };

    f32x4 acc[2][2][2][4];     // [pixel half][W half][channel tile][pixel tile]
    const int frow = lane & 15, fk = lane >> 4;
    const int sw0 = ((0 * 4 + fk) ^ (frow & 7)) * 8, sw1 = ((1 * 4 + fk) ^ (frow & 7)) * 8;
    const uint32_t wrow_b0 = (uint32_t)(((wc * 32 + frow) * 64 + sw0) * 2), wrow_b1 = (uint32_t)(((wc * 32 + frow) * 64 + sw1) * 2);
    // this lane's byte offset inside a halo row for the three column shifts dx = -D, 0, +D (k-half 0; k-half 1 is the same ^ 64):
    // pixel frow + dxi * D (+ 16 for the second pixel tile: the same swizzle key), 16-byte slot fk ^ (pixel & 7)
    uint32_t xl[3];
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) xl[dxi] = (uint32_t)((frow + dxi * D) * 128 + ((((frow + dxi * D) ^ fk) & 7) << 4));
    const int so_ch = (fk & 1) * 16 + (fk >> 1) * 8;
    const f16 lo = a.relu ? (f16)0.f : (f16)(-__builtin_inff());
    const f16x4 lo4 = {lo, lo, lo, lo};

    // ---- prologue (once per workgroup): all ten rows of chunk 0 (ring base 0), WA(0), WB(0), WA(1), WB(1); everything lands
    for (int r = 0; r < 10; ++r) stage_row(xb_c, r, 0);
    stage_w(0, wb_c, 0);
    stage_w(1, wb_c, 0);
    stage_w(0, wb_c + (size_t)CPT * (256 * 64), 1);                  // K-tile 1 = (chunk 0, tap 1): packed index 1 * CPT + 0
    stage_w(1, wb_c + (size_t)CPT * (256 * 64), 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (tid == 0) {
        const int t1 = (int)atomicAdd(&tile_ctr[xcd], 1u);
        if (t1 == last_draw) tile_ctr[xcd] = 0u;
        lds_ticket[1] = t1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int vnext = __builtin_amdgcn_readfirstlane(lds_ticket[1]);
    if (wave >= 4) __builtin_amdgcn_s_barrier();          // waves 4-7 run one barrier behind

    f16x8 xf[4][2], wa[2][2], wb[2][2];
    int sp = 0;                 // weight ring buffer of the current K-tile
    int rbase = 0;              // ring base of the current chunk (even; + 10 mod 16 per chunk, across tiles)
    int tpar = 0;

// fragment p of pixel half I: lattice row I * 4 + wp * 2 + (p >> 1) of the tile (its slot address is in xa[I][p >> 1][k half]),
// columns (p & 1) * 16 + frow
#define LT_LOAD_X(I)                                                                            \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                             \
        xf[p][0] = LT_LDS_F16X8(xa[I][p >> 1][0] + (p & 1) * 2048);                             \
        xf[p][1] = LT_LDS_F16X8(xa[I][p >> 1][1] + (p & 1) * 2048);                             \
    }
#define LT_LOAD_W(dstf, HALF)                                                                   \
    _Pragma("unroll") for (int cc = 0; cc < 2; ++cc) {                                          \
        dstf[cc][0] = LT_LDS_F16X8(wad0 + (HALF) * LT_HALF_ELEMS * 2 + cc * 2048);              \
        dstf[cc][1] = LT_LDS_F16X8(wad1 + (HALF) * LT_HALF_ELEMS * 2 + cc * 2048);              \
    }
#define LT_SEG_SYNC(VM)                                                                         \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    __builtin_amdgcn_s_barrier();                                                               \
    __builtin_amdgcn_sched_barrier(0);
#define LT_MMA(i, j, wfrag, TAILBAR)                                                            \
    __builtin_amdgcn_s_setprio(1);                                                              \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                            \
        _Pragma("unroll") for (int cc = 0; cc < 2; ++cc)                                        \
            _Pragma("unroll") for (int p = 0; p < 4; ++p)                                       \
                acc[i][j][cc][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag[cc][kk], xf[p][kk], acc[i][j][cc][p], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if (TAILBAR) __builtin_amdgcn_s_barrier();                                                  \
    __builtin_amdgcn_sched_barrier(0);
// operand addresses of K-tile (tap TP) of the chunk with ring base RBASE: four wave-uniform row-slot addresses + the lane's column
// offset for the tap's dx
#define LT_XADDR_SEL(TP)                                                                        \
    {                                                                                           \
        const int dyi_ = (TP) / 3, dxi_ = (TP) - dyi_ * 3;                                      \
        xsel[0] = dxi_ == 0 ? xl[0] : (dxi_ == 1 ? xl[1] : xl[2]);                              \
        xsel[1] = xsel[0] ^ 64u;                                                                \
    }
#define LT_XADDR_ROWS(I_, TP, RBASE)                                                            \
    {                                                                                           \
        const int dyi_ = (TP) / 3;                                                              \
        _Pragma("unroll") for (int pr_ = 0; pr_ < 2; ++pr_) {                                   \
            const uint32_t ra_ = ring_b + (uint32_t)((((RBASE) + dyi_ + (I_) * 4 + wp * 2 + pr_) & (LT_RING - 1)) * ROW_B); \
            xa[I_][pr_][0] = xsel[0] + ra_; xa[I_][pr_][1] = xsel[1] + ra_;                     \
        }                                                                                       \
    }
#define LT_XADDR(TP, RBASE) { LT_XADDR_SEL(TP) LT_XADDR_ROWS(0, TP, RBASE) LT_XADDR_ROWS(1, TP, RBASE) }
// All eight addresses of the NEXT K-tile in phase 4's load segment (no operand reads).  Round 6, same box (heads.conv_d6, ms): the
// second pixel half's four in the next K-tile's phase 2 instead (the lightest segment with reads) 3.41-3.43 against 3.35-3.40; no
// address arithmetic at all (timing only, wrong results) 3.31-3.33; the generic persistent kernel 3.41-3.42.
#define LT_XADDR_NEXT                                                                           \
    {                                                                                           \
        int tapn_ = tap + 1, rbn_ = rbase;                                                      \
        if (tapn_ == 9) { tapn_ = 0; rbn_ = (rbase + 10) & (LT_RING - 1); }                     \
        LT_XADDR(tapn_, rbn_)                                                                   \
    }
#define LT_XADDR_PIN asm volatile("" : "+v"(xa[0][0][0]), "+v"(xa[0][0][1]), "+v"(xa[0][1][0]), "+v"(xa[0][1][1]),  \
                                       "+v"(xa[1][0][0]), "+v"(xa[1][0][1]), "+v"(xa[1][1][0]), "+v"(xa[1][1][1]));
// the halo row DMA of row slot idx = tap * 2 + s (s = 0: phase 2, 1: phase 3) of the current chunk: see the table in the header.
// Straight-line scalar code on purpose (a branch in a load segment splits the block and the scheduler's fences with it: the first
// form of this macro - nested conditionals - compiled to six scalar branches per K-tile and ran 8 % slower than the generic kernel):
// the two chunk origins are loop-carried SGPR pairs (org_c / org_n, moved once per chunk).
// The row index comes out of a packed table (4 bits per tap, one 64-bit constant per phase: a shift and a mask - compare-and-add
// arithmetic on the tap index was lowered to VECTOR instructions with a quarter-rate multiply and three v_readfirstlane).
#define LT_ROW(S)                                                                               \
    {                                                                                           \
        /* rows by tap, phase 2: c.8 n.0 n.2 n.4 n.6 (6) (6) n.7 (7); phase 3: c.9 n.1 n.3 n.5 (6) (6) (6) (7) (7) */ \
        const unsigned long long tbl_ = (S) ? 0x776665319ull : 0x776664208ull;                  \
        const int r_ = (int)(tbl_ >> (tap * 4)) & 15;                                           \
        const bool nx_ = tap != 0;                                                              \
        stage_row(nx_ ? org_n : org_c, r_, nx_ ? rbase + 10 : rbase);                           \
    }
// chunk origins after the move to chunk CH of the current tile: its own, and that of the chunk staged during it (the next
// chunk of the tile, or chunk 0 of the next tile - descriptor n, which aliases c when there is none)
#define LT_ORG_SET(CH) { org_c = xb_c + (CH) * 128; org_n = (CH) + 1 == CPT ? xb_n : xb_c + ((CH) + 1) * 128; }
#define LT_WS_HOOK ws_p = (ch2 >= CPT ? wb_n : wb_c) + (size_t)(tap2 * CPT + (ch2 >= CPT ? ch2 - CPT : ch2)) * (256 * 64);
#define LT_WAD_SET(SP) { const uint32_t wb_ = lds_base + (uint32_t)(SP) * (2 * LT_HALF_ELEMS * 2); wad_c[0] = wb_ + wrow_b0; wad_c[1] = wb_ + wrow_b1; }
// One K-tile (ch, tap).  K-tile kt+2 = (ch2, tap2); a chunk index == CPT means chunk 0 of the next tile (descriptor n, which
// aliases c when there is none: the re-staged data lands in slots that are free).
#define LT_STEP(VM, LAST)                                                                       \
    {                                                                                           \
        const uint32_t wad0 = wad_c[0], wad1 = wad_c[1];                                        \
        int tap1 = tap + 1, ch1 = ch;                                                           \
        if (tap1 == 9) { tap1 = 0; ch1 = ch + 1; }                                              \
        int tap2 = tap1 + 1, ch2 = ch1;                                                         \
        if (tap2 == 9) { tap2 = 0; ch2 = ch1 + 1; }                                             \
        LT_LOAD_X(0)                                                                            \
        LT_LOAD_W(wa, 0)                                                                        \
        LT_SEG_SYNC(VM)                                                                         \
        LT_MMA(0, 0, wa, 1)                                                                     \
        LT_LOAD_W(wb, 1)                                                                        \
        LT_ROW(0)                                                                               \
        LT_WS_HOOK stage_w1(0, 0, ws_p, sp);                                                    \
        LT_SEG_SYNC(VM)                                                                         \
        LT_MMA(0, 1, wb, 1)                                                                     \
        LT_LOAD_X(1)                                                                            \
        LT_ROW(1)                                                                               \
        stage_w1(0, 1, ws_p, sp);                                                               \
        LT_SEG_SYNC(VM)                                                                         \
        LT_MMA(1, 1, wb, 1)                                                                     \
        stage_w(1, ws_p, sp);                                                                   \
        LT_WAD_SET(sp ^ 1) asm volatile("" : "+v"(wad_c[0]), "+v"(wad_c[1]));                   \
        LT_XADDR_NEXT LT_XADDR_PIN                                                              \
        LT_SEG_SYNC(VM)                                                                         \
        LT_MMA(1, 0, wa, !(LAST))                                                               \
        sp ^= 1;                                                                                \
        {                                                                                       \
            const bool wrap_ = tap == 8;                                                        \
            const char* oc_ = org_c; const char* on_ = org_n;                                   \
            LT_ORG_SET(ch + 1)                                                                  \
            org_c = wrap_ ? org_c : oc_; org_n = wrap_ ? org_n : on_;                           \
            ch = wrap_ ? ch + 1 : ch; rbase = wrap_ ? (rbase + 10) & (LT_RING - 1) : rbase;     \
            tap = wrap_ ? 0 : tap + 1;                                                          \
        }                                                                                       \
    }

    uint32_t xsel[2];                       // the lane's column offset for the K-tile's dx, both k halves
    uint32_t xa[2][2][2];                   // this lane's operand addresses of the NEXT K-tile to run: [pixel half][row of the pair][k half]
    uint32_t wad_c[2];
    const char *org_c, *org_n;              // origin (halo row 0) of the current chunk and of the chunk staged during it
    LT_WAD_SET(0)
    const f16* ws_p = wb_c;
    LT_XADDR(0, 0)
    for (;;) {
        // ticket of the tile after next (see conv_mfma256_persistent_kernel)
        int ticket = vnext;
        const bool draw = vnext < vtotal;
        if (wave == 0 && draw) {
            const unsigned inc = 1u, off = (unsigned)xcd * 4u;
            unsigned long long saved_exec;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                         : "=&v"(ticket), "=&s"(saved_exec) : "v"(off), "v"(inc), "s"(tile_ctr) : "memory");
        }
        live_n = draw;
        if (live_n) locate(vnext, xb_n, wb_n, gi_n, nt_n, n_n, y0_n, tx_n);
        // the accumulators start at the bias (fp32, from LDS): the epilogue has no add
        {
            const float* bp = lds_bias + a.g[gi_c].bias_off + nt_c * 256 + wc * 32 + fk * 4;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const f32x4 b4 = *(const f32x4*)(bp + j * 128 + cc * 16);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int p = 0; p < 4; ++p) acc[i][j][cc][p] = b4;
                }
        }
        int ch = 0, tap = 0;
        LT_ORG_SET(0)
        LT_STEP(22, 0)                                         // 6 DMA instructions per K-tile + the previous tile's 16 stores
        if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(6)" : "+v"(ticket) : : "memory");      // the atomic is older than this K-tile's DMAs
            if (lane == 0) {
                lds_ticket[tpar] = ticket;
                if (draw && ticket == last_draw) tile_ctr[xcd] = 0u;
            }
        }
        for (int kt = 1; kt < T - 1; ++kt) { LT_STEP(6, 0) }
        LT_STEP(6, 1)
        // last MFMA segment had no trailing barrier: waves 0-3 take it before their epilogue, waves 4-7 after
        if (wave < 4) __builtin_amdgcn_s_barrier();

        // ---- epilogue: 16 independent 16-byte stores, no loads from global memory
        // (Round 6, same box: the epilogue of the three accumulator quadrants that are complete after phases 1-3 of the LAST K-tile
        // issued in that K-tile's load segments - ~45 vector instructions + 4 stores each, behind the segment's DMA, counted waits
        // 6 | 10 | 14 | 18 and 22 | 18 | 14 | 6 in the next tile's first K-tile - so that three quarters of the stores drain under
        // the tile's own MFMAs: heads.conv_d6 3.367 / 3.367 / 3.386 against 3.386 / 3.385 / 3.373 ms.  Within the noise: at the power
        // cap the tile boundary is not what this kernel waits for.  Removed.)
        {
            const ConvGroupArgs& g = a.g[gi_c];
            const int cbase = g.out_coff + nt_c * 256 + wc * 32 + so_ch;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                size_t opix[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int oy = y0_c + (i * 4 + wp * 2 + (p >> 1)) * D;
                    const int ox = tx_c * 32 + (p & 1) * 16 + frow;
                    opix[p] = ((size_t)(n_c * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + cbase;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        uint32_t u[2][2];
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc) {
                            const f32x4 vv = acc[i][j][cc][p];
                            f16x4 h = {(f16)vv[0], (f16)vv[1], (f16)vv[2], (f16)vv[3]};
                            h = __builtin_elementwise_max(h, lo4);
                            __builtin_memcpy(u[cc], &h, 8);
                        }
                        const auto s0 = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
                        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        *(u32x4*)((f16*)a.out + opix[p] + j * 128) = o;
                    }
            }
        }
        if (wave >= 4) __builtin_amdgcn_s_barrier();
        if (!live_n) break;
        vnext = __builtin_amdgcn_readfirstlane(lds_ticket[tpar]);
        tpar ^= 1;
        xb_c = xb_n;
        wb_c = wb_n; gi_c = gi_n; nt_c = nt_n; n_c = n_n; y0_c = y0_n; tx_c = tx_n;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();           // pair the extra barrier of waves 4-7
}

// Eligibility: 3x3 taps at (dy, dx) in {-D, 0, D}^2 in row-major order, stride 1, the same for every group; the map is covered by
// tiles of 8 lattice rows x 32 columns (H % (8 D) == 0, W % 32 == 0); the border holds the dilation.  Returns D (0: not eligible).
int conv_mfma256_lattice_dilation(const ConvKArgs& a, int groups) {
    if (a.res || a.in_stride != 1 || a.out_scale != 1 || a.ntaps != 9 || a.ksteps < 9 || a.ksteps != 9 * a.cpt) return 0;
    const int D = 6;                                 // the one instantiated dilation (header.py:13)
    const int Hm = a.HmWm / a.Wm;
    if (a.in_P < D || a.Wm % 32 || Hm % (8 * D) || a.M % a.HmWm) return 0;
    if (a.MT != (a.M / a.HmWm) * (a.Wm / 32) * (Hm / 8)) return 0;
    const int pitch = a.in_Wp * a.in_C;
    for (int g = 0; g < groups; ++g) {
        if (a.g[g].out_oy || a.g[g].out_ox) return 0;
        for (int t = 0; t < 9; ++t)
            if (a.g[g].tap_off[t] != (t / 3 - 1) * D * pitch + (t % 3 - 1) * D * a.in_C) return 0;
    }
    return D;
}

hipError_t launch_conv_mfma256_lattice(const ConvKArgs& a, int groups, int nbias, int cu_count, unsigned int* tile_ctr, hipStream_t s) {
    const int per_xcd = cu_count / 8;          // one workgroup per CU whatever the tile count (launch_conv_mfma256)
    const dim3 grid(per_xcd * 8, 1, 1), block(512, 1, 1);
    hipLaunchKernelGGL((conv_mfma256_lattice_kernel<6>), grid, block, 0, s, a, groups, nbias, tile_ctr);
    return hipGetLastError();
}
