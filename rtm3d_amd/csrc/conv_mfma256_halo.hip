// Halo-tile form of the persistent 256x256 implicit-GEMM convolution, for layers whose taps all lie
// within +-1 pixel (the dilation-1 head conv: 4 groups x 256->256, 3x3; the four sub-pixel phases of the neck's 4x4 / stride-2
// transposed convs: 2x2 taps each).
//
// conv_mfma256.hip stages a 256-pixel x 64-channel operand half-tile pair once per (tap, chunk): every
// input pixel crosses L2 -> LDS nine times per chunk, and the rows a tile needs for dy = +-1 are fetched
// again from beyond L2 (they were the dy = 0 rows of a tile that ran a whole K loop earlier): measured
// 14.1 GB of fabric traffic per launch for 4.0 GB of algorithmic bytes.  Here the output tile is an
// 8 x 32 pixel block and a workgroup stages its (8+2) x (32+2) HALO of one 64-channel chunk ONCE
// (43.5 KB, double-buffered); the nine taps of that chunk read shifted rows of it (the swizzle
// chunk ^= pixel & 7 keeps ds_read_b128 conflict-free for every shift).  The K loop runs chunk-major
// (kt = chunk * ntaps + tap) over the same packed weights (K-tile index tap * cpt + chunk).
//   pixel operand DMA per 64-channel chunk: 43.5 KB instead of 9 x 32 KB; weights unchanged (9 x 32 KB).
//
// Schedule, tickets, bias-in-LDS, 16-byte swapped stores: as the persistent kernel of conv_mfma256.hip.
// Differences: the weight ring holds only WA/WB half-tiles (64 KB); the halo of the NEXT chunk (or of the
// next tile's first chunk) is staged one halo ROW per DMA instruction (34 lanes of every wave, the row a running cursor in SGPRs),
// two rows per K-tile with nine taps, three with four, so that every K-tile issues the same 0+2+2+2 (0+2+2+3) DMA instructions -
// P2 / P3: a halo row + one half of WA, P4: WB (+ a row), all of K-tile kt+2 - and the counted s_waitcnt stays an immediate
// (vmcnt(6) / vmcnt(7); the tap count is a template parameter).
// When there is no next tile the same instructions re-stage data of the current tile into ring slots
// that are already free: no dummy slot (the LDS is full) and no run-time counts.  (First version: six
// 512-lane halo DMAs on taps 0-2 and per-phase counts dispatched through a switch: the scalar code of
// that dispatch made the load segments longer than the partner wave's MFMA segment, -19 %.)
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

#define HALF_ELEMS (128 * 64)                 // one weight half-tile: 128 rows x 64 halves = 16 KB
#define WRING_ELEMS (4 * HALF_ELEMS)          // two K-tiles x (WA, WB)
#define HALO_W 34
#define HALO_PIX (10 * HALO_W)                // 340 pixels
#define HALO_PIECES (HALO_PIX * 8)            // 2720 16-byte pieces per chunk
#define HALO_BUF_PIECES 2912                  // buffer size (the halo itself: 2720 pieces)
#define HALO_ELEMS (HALO_BUF_PIECES * 8)
#define HALO_MAX_BIAS 1024
#define H_STAMP_KT 3                          // (diagnostic build -DC256_STAMPS only; the LDS has 1 KB to spare: K-tiles 4..6)
#define H_STAMP_N 20

// per group, all taps packed into one 64-bit word (4 bits per tap: dy+1 in bits 0-1, dx+1 in bits 2-3) that lives
// in SGPRs for the whole tile: an s_load per K-tile would put its latency in front of the operand reads
struct HaloTaps { unsigned long long taps[RT_MAX_GROUPS]; };

// LDS-DMA as inline asm (m0 = LDS base of the wave's run, one 16-byte piece per active lane).  The compiler
// must not know these write LDS: its waitcnt pass treats every visible ds_read as possibly aliasing a pending
// LDS-DMA and puts s_waitcnt vmcnt(0) in front of it, draining the DMA ring in every phase; ordering against
// the DMA is SEG_SYNC's counted vmcnt + barrier.  The operand reads stay ordinary loads, so that the hazard
// recogniser sees them: with the reads hidden in asm instead, the register allocator handed a renamed
// accumulator's old registers to a ds_read whose data landed before a queued MFMA had read them as SrcC.
#define DMA16 RT_DMA16                              // common.h: the one LDS-DMA definition
#define DMA16_SBASE_LANES RT_DMA16_SBASE_LANES
#define DMA16_SBASE RT_DMA16_SBASE                  // wave-uniform 64-bit base in SGPRs + 32-bit byte offset per lane
#define LDS_F16X8(byte_addr) (*(const LDS_AS f16x8*)(uintptr_t)(byte_addr))
#define LDS_F32X4(byte_addr) (*(const LDS_AS f32x4*)(uintptr_t)(byte_addr))

template <int STATS, int NTAP>          // NTAP: 9 (3 x 3 layers) or 4 (one sub-pixel phase of a 4 x 4 / stride-2 transposed conv)
__global__ __launch_bounds__(512) void conv_mfma256_halo_kernel(const ConvKArgs a, const HaloTaps ht, const int groups, const int nbias,
                                                                unsigned int* tile_ctr, float* stat_out, const int one_list) {
    __shared__ __attribute__((aligned(128))) f16 lds[WRING_ELEMS + 2 * HALO_ELEMS];
    __shared__ __attribute__((aligned(16))) float lds_bias[HALO_MAX_BIAS + 4];     // + two ticket words
#ifdef C256_STAMPS
    __shared__ unsigned long long lds_stamp[2][H_STAMP_KT][H_STAMP_N];
    int stamp_tile = 0, stamp_kt = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wc = wave >> 1;
    const int T = a.ksteps, CPT = a.cpt;
    for (int i = tid; i < nbias; i += 512) lds_bias[i] = a.bias[i];

    // tile list (as conv_mfma256_persistent_kernel): position v -> (group, pixel tile, channel tile)
    // (one_list: a launch of at most one round draws from ONE list with one workgroup per CU - conv_mfma256_persistent_kernel)
    const int xcd = one_list ? 0 : blockIdx.x & 7, per_xcd = one_list ? gridDim.x : gridDim.x >> 3;
    const int chunk = one_list ? a.MT : (a.MT + 7) >> 3;
    int mt_here = a.MT - xcd * chunk;
    mt_here = mt_here < 0 ? 0 : (mt_here > chunk ? chunk : mt_here);
    const int jbs = mt_here * a.NT;
    const int vtotal = jbs * groups;
    int* const lds_ticket = (int*)(lds_bias + HALO_MAX_BIAS);
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
    const int Hm = a.HmWm / a.Wm;
    const int tiles_x = a.Wm >> 5, tpi = tiles_x * (Hm >> 3);      // 8 x 32 pixel tiles per image

    // Halo staging.  The halo of the NEXT chunk (10 rows x 34 pixels x 64 channels) is staged one ROW per DMA instruction - 34 pixels
    // x 8 pieces = 272 = 8 waves x 34 lanes - while the current chunk's taps are multiplied: a lane's source offset is the same in
    // every row but for the row's parity in the swizzle key (roff_e / roff_o, two registers for the whole kernel), and the row itself
    // is a wave-uniform cursor in SGPRs (xs_src, xs_dst) that moves one row per instruction, beside the MFMAs of the segment
    // that follows.  LDS image: piece p = row * 272 + pixel * 8 + chunk at p * 16 bytes.
    //   NTAP == 9: two rows per K-tile (phases 2 and 3): rows 0 .. 9 in taps 0 .. 4, then the cursor swings between rows 8 and 9
    //              (same bytes again) - the instruction count per K-tile must not change, the counted waits are immediates;
    //   NTAP == 4: three rows per K-tile (phases 2, 3 and 4): rows 0 .. 8 in taps 0 .. 2, then 9, 8, 7.
    // What the last K-tile of a chunk issues may still be in flight when the next chunk's first K-tile reads its operands (a read
    // in phase p is covered by the wait of phase p-1, which lets the DMA of the four phases before it stay pending), so it must
    // not carry anything new beyond phase 2, whose row - the bottom one - is first read in phase 3 of that K-tile: covered.
    // (History: piece-linear slices, 2*NTAP-1 per chunk, ~17 vector instructions each with a division by 34, in the load segments
    // of phases 2 and 3 - which is what the partner wave's MFMA segment waits for; round 5: rows for the 3x3 layers, -2.3 %; then the
    // row address itself as a running cursor instead of ~25 scalar instructions per slice, and rows for the transposed convs too.)
    constexpr unsigned long long xlanes = (1ull << 34) - 1;       // lanes of a row DMA (issued by every wave: RT_DMA16_SBASE_LANES)
    uint32_t roff_e, roff_o;
    {
        const int rp = wave * 34 + (lane < 34 ? lane : 33);
        const int rhx = rp >> 3, rhcs = rp & 7;
        roff_e = (uint32_t)(rhx * a.in_C + ((rhcs ^ (rhx & 7)) * 8)) * 2u;        // even halo row: key = hx & 7          (BYTES: the
        roff_o = (uint32_t)(rhx * a.in_C + ((rhcs ^ ((rhx ^ 4) & 7)) * 8)) * 2u;  // odd halo row:  key = (hx ^ 4) & 7     lane offset of DMA16_SBASE)
    }
    // tile descriptors (current / next): halo origin in the input tensor, weight base, indices
    const char *xb_c, *xb_n;                // halo pixel (0,0), channel 0 of the group slice
    const f16 *wb_c, *wb_n;
    int gi_c, gi_n, nt_c, nt_n, n_c, n_n, ty_c, ty_n, tx_c, tx_n;
    auto locate = [&](int vv, const char*& xb, const f16*& wb, int& gi, int& nt, int& n, int& ty, int& tx) {
        // (group-major order.  Round 5, same-box A/B: the four sub-pixel phases of a transposed conv - which read the SAME input
        // halo - on consecutive tickets instead, so that three of four halo fetches hit L2: no change, 0.507 vs 0.506 ms, although
        // these launches fetch their input 4-5 times over, 0.63 GB against 0.13 GB: the fabric traffic is not what they wait for.)
        gi = vv / jbs;
        const int jb = vv - gi * jbs;
        const int q = jb / a.NT;
        nt = jb - q * a.NT;
        const int mt = xcd * chunk + q;
        n = mt / tpi;
        const int r = mt - n * tpi;
        ty = r / tiles_x;
        tx = r - ty * tiles_x;
        const ConvGroupArgs& g = a.g[gi];
        xb = (const char*)(a.in + ((size_t)(n * a.in_Hp + ty * 8 - 1 + a.in_P) * a.in_Wp + tx * 32 - 1 + a.in_P) * a.in_C + g.in_coff);
        wb = a.wgt + g.w_off + (size_t)nt * T * (256 * 64);
    };
    bool live_n = false;
    locate(v, xb_c, wb_c, gi_c, nt_c, n_c, ty_c, tx_c);
    xb_n = xb_c; wb_n = wb_c; gi_n = gi_c; nt_n = nt_c; n_n = n_c; ty_n = ty_c; tx_n = tx_c;

    const ptrdiff_t pitch_b = (ptrdiff_t)a.in_Wp * a.in_C * 2;     // one halo row down, in bytes
    constexpr int ROW_LDS_B = 272 * 16;
    const char* xs_src;
    uint32_t xs_dst;
    // cursor to row 0 of chunk ch of the tile at xb, into halo buffer hp
    auto xs_reset = [&](const char* xb, int ch, int hp) {
        xs_src = xb + ch * 128;
        xs_dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(WRING_ELEMS + hp * HALO_ELEMS + wave * 34 * 8) * 2u);
    };
    // (Round 6, same box: the row slots that only repeat a staged row - taps 5 .. 8 with nine taps - cut down to ONE 16-byte piece
    // through the lane mask, 35 KB less staging per chunk: heads.conv_d1 and the transposed convs unchanged within +-0.5 %.)
    auto xs_issue = [&](uint32_t roff) { DMA16_SBASE_LANES(roff, xs_src, xs_dst, xlanes); };
    auto xs_move = [&](bool down) {
        xs_src += down ? pitch_b : -pitch_b;
        xs_dst += down ? (uint32_t)ROW_LDS_B : (uint32_t)-ROW_LDS_B;
    };

    // weight half-tile (0 = WA, 1 = WB) of the packed K-tile at wk into ring buffer par
    const uint32_t wvoff = (uint32_t)tid * 16u;
    // one of the two DMA instructions (i = 0, 1) of a weight half-tile
    auto stage_w1 = [&](int half, int i, const f16* wk, int par) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((par * 2 + half) * HALF_ELEMS + (i * 512 + wave * 64) * 8) * 2u);
        DMA16_SBASE(wvoff, wk + half * HALF_ELEMS + i * 512 * 8, dst);
    };
    auto stage_w = [&](int half, const f16* wk, int par) {
        const f16* ws = wk + half * HALF_ELEMS;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((par * 2 + half) * HALF_ELEMS + (i * 512 + wave * 64) * 8) * 2u);
            DMA16_SBASE(wvoff, ws + i * 512 * 8, dst);
        }
    };

    f32x4 acc[2][2][2][4];     // [pixel half][W half][channel tile][pixel tile]
    const int frow = lane & 15, fk = lane >> 4;
    const int sw0 = ((0 * 4 + fk) ^ (frow & 7)) * 8, sw1 = ((1 * 4 + fk) ^ (frow & 7)) * 8;
    const uint32_t wrow_b0 = (uint32_t)(((wc * 32 + frow) * 64 + sw0) * 2), wrow_b1 = (uint32_t)(((wc * 32 + frow) * 64 + sw1) * 2);
    // halo byte offset of this lane's fragment (half 0, pixel tile 0) at tap (0,0): tile row wp*2, column frow;
    // the other fragments are compile-time offsets from it, a tap adds shift*128 (wave-uniform)
    const uint32_t xlane = (uint32_t)(((wp * 2 + 1) * HALO_W + frow + 1) * 128);
    // swizzled 16-byte slot of this lane's k-chunk fk for the three column shifts dx = -1, 0, +1
    // (the k-half 1 slot and an odd halo row are each the same slot ^ 64 bytes)
    const uint32_t ck_m = (uint32_t)((((frow + 0) ^ fk) & 7) << 4), ck_0 = (uint32_t)((((frow + 1) ^ fk) & 7) << 4),
                   ck_p = (uint32_t)((((frow + 2) ^ fk) & 7) << 4);
    const int so_ch = (fk & 1) * 16 + (fk >> 1) * 8;
    const f16 lo = a.relu ? (f16)0.f : (f16)(-__builtin_inff());
    const f16x4 lo4 = {lo, lo, lo, lo};

    // ---- prologue (once per workgroup): halo of chunk 0, WA(0), WB(0), WA(1), WB(1); everything lands
    xs_reset(xb_c, 0, 0);
    for (int r = 0; r < 10; ++r) { xs_issue((r & 1) ? roff_o : roff_e); xs_move(true); }
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
    int hpar = 0;               // halo buffer of the current chunk
    int tpar = 0;

#define LOAD_X_H(I)                                                                             \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                             \
        /* halo row of fragment p = 1 + dy + I*4 + wp*2 + (p>>1): its parity swaps the two k-half slots */ \
        const uint32_t a0 = (p >> 1) ? xw : xu, a1 = (p >> 1) ? xu : xw;                         \
        xf[p][0] = LDS_F16X8(a0 + (((I) * 4 + (p >> 1)) * HALO_W + (p & 1) * 16) * 128);        \
        xf[p][1] = LDS_F16X8(a1 + (((I) * 4 + (p >> 1)) * HALO_W + (p & 1) * 16) * 128);        \
    }
#define LOAD_W_H(dstf, HALF)                                                                    \
    _Pragma("unroll") for (int cc = 0; cc < 2; ++cc) {                                          \
        dstf[cc][0] = LDS_F16X8(wad0 + (HALF) * HALF_ELEMS * 2 + cc * 2048);                    \
        dstf[cc][1] = LDS_F16X8(wad1 + (HALF) * HALF_ELEMS * 2 + cc * 2048);                    \
    }
// Diagnostic build only (-DC256_STAMPS, see conv_mfma256.hip): s_memtime stamps of waves 0 and 4 in the K loop of the second tile.
#ifdef C256_STAMPS
#define HSTAMP(k) if ((wave & 3) == 0 && stamp_tile == 1 && stamp_kt >= 4 && stamp_kt < 4 + H_STAMP_KT) { unsigned long long ts_; \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory"); if (lane == 0) lds_stamp[wave >> 2][stamp_kt - 4][k] = ts_; }
#define HSTAMP_NEXT ++stamp_kt;
#else
#define HSTAMP(k)
#define HSTAMP_NEXT
#endif
#define SEG_SYNC_H(VM, K0)                                                                      \
    HSTAMP((K0) + 0)                                                                            \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");                           \
    HSTAMP((K0) + 1)                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    __builtin_amdgcn_s_barrier();                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    HSTAMP((K0) + 2)
#define MMA_H(i, j, wfrag, FIRST, TAILBAR, K0) MMA_HX(i, j, wfrag, FIRST, TAILBAR, K0, , )
#define MMA_HX(i, j, wfrag, FIRST, TAILBAR, K0, PRE, PIN)                                       \
    __builtin_amdgcn_s_setprio(1);                                                              \
    PRE                                                                                         \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                            \
        _Pragma("unroll") for (int cc = 0; cc < 2; ++cc)                                        \
            _Pragma("unroll") for (int p = 0; p < 4; ++p)                                       \
                acc[i][j][cc][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag[cc][kk], xf[p][kk], acc[i][j][cc][p], 0, 0, 0); \
    PIN                                                                                         \
    __builtin_amdgcn_s_setprio(0);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    HSTAMP((K0) + 3)                                                                            \
    if (TAILBAR) __builtin_amdgcn_s_barrier();                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    HSTAMP((K0) + 4)
// LDS addresses of this lane's pixel-operand fragments for tap `tp` of the halo in buffer hp (xu / xw: the two k-half slots of an
// even halo row; LOAD_X_H swaps them on odd rows).  Round 5: computed a K-tile AHEAD, in phase 4's load segment - the one without
// operand reads - instead of at the head of phase 1's, whose reads wait for them.  (First placed beside the MFMAs of phase 4, with a
// scalar load and a branch for the next tile's tap word: heads.conv_d1 3.336 -> 3.28 ms, transposed convs -1 %, from moving them out
// of the MFMA segment and keeping both tap words in SGPRs for the tile.)
#define HALO_XADDR(TW, TP, HP, XU, XW)                                                          \
    {                                                                                           \
        const uint32_t hb_ = lds_base + (uint32_t)(WRING_ELEMS + (HP) * HALO_ELEMS) * 2;          \
        const int tbits_ = (int)((TW) >> (4 * (TP))) & 15;                                      \
        const int dys_ = (tbits_ & 3) - 1, dxs_ = (tbits_ >> 2) - 1;                              \
        const uint32_t ck_ = dxs_ < 0 ? ck_m : (dxs_ > 0 ? ck_p : ck_0);                          \
        const uint32_t xt0_ = hb_ + xlane + (uint32_t)((dys_ * HALO_W + dxs_) * 128) + ck_;       \
        XU = xt0_ ^ (uint32_t)(((1 + dys_) & 1) << 6); XW = XU ^ 64u;                            \
    }
#define HALO_XADDR_NOW const uint32_t xu = xu_c, xw = xw_c;
#define HALO_XADDR_NEXT                                                                         \
    {                                                                                           \
        int tapn_ = tap + 1, hparn_ = hpar;                                                     \
        unsigned long long twn_ = tapword;                                                      \
        if (tapn_ == NTAP) { tapn_ = 0; hparn_ ^= 1; if (ch + 1 == CPT) twn_ = tapword_n; }     \
        HALO_XADDR(twn_, tapn_, hparn_, xu_c, xw_c)                                             \
    }
#define HALO_XADDR_PIN asm volatile("" : "+v"(xu_c), "+v"(xw_c));
// (Same-box A/Bs of round 5 on this kernel, heads.conv_d1 / fusion_up5.2 / kfpn_up3 ms: addresses a K-tile ahead 3.63 / 0.500 / 0.464
// against 3.64 / 0.507 / 0.470 computed in place - kept; a load segment's DMA issued in FRONT of its operand reads instead of
// behind them 3.85 / 0.547 / 0.465 - the reads then queue behind the DMA's address path; the halo-slice source addresses
// computed beside the MFMAs 4.02 / 0.527 / 0.469: their quarter-rate integer instructions outlast the MFMA gaps - rejected.)
// One K-tile (ch, tap).  K-tiles kt+1 / kt+2 = (ch1, tap1) / (ch2, tap2); a chunk index == CPT means chunk 0 of
// the next tile (descriptor n, which aliases c when there is none: the re-staged data lands in free slots).
// WB's DMA is issued in phase 4 of the K-tile two before its use, next to WA's (a load segment without operand reads), not in
// phase 1 of the K-tile before (behind 12 operand reads): its target half was last read in phase 2, and with six DMA
// instructions per K-tile either way the counted waits retire every half-tile at the same barrier as before.  Same box, round 5,
// three interleaved rounds: heads.conv_d1 3.665 / 3.702 / 3.669 -> 3.558 / 3.549 / 3.561 ms, fusion_up5.2 0.510 -> 0.500,
// kfpn_up3 0.468 -> 0.471.  The halo slices do not follow: the second slice beside
// the first in phase 2 (phase 3 then only reads) 3.51 -> 3.72 ms, both slices in phase 4 (all six DMAs in the read-free segment,
// phases 1-3 only read) 3.51 -> 3.57 and fusion_up5.2 0.50 -> 0.56 - one slice behind the reads of phases 2 and 3 each stays.
// The halo-row cursor (xs_src, xs_dst) moves in the load segment, between the segment's operand reads and its row DMA: scalar
// instructions only, which run while the reads are in flight.  (Same box, round 5, heads.conv_d1: the same instructions beside the
// MFMAs of the segment before 3.37 -> 3.51 ms - a wave issues in order, and scalar work between its MFMAs opens bubbles in a pipe
// that is busy 16 cycles out of 16; the row DMA itself moved beside the MFMAs 3.61; behind an explicit wait for the reads 3.52.)
// xr_a / xr_b: the lane offsets (row parity) of the K-tile's rows.
#define XS_TO_ROW_A     /* straight-line on purpose: a branch splits the block and the scheduler's fences with it */ \
    {                                                                                           \
        const char* rs_ = (xn ? xb_n : xb_c) + (xn ? 0 : ch + 1) * 128;                         \
        const uint32_t rd_ = lds_base + (uint32_t)(WRING_ELEMS + (hpar ^ 1) * HALO_ELEMS + wave * 34 * 8) * 2u; \
        xs_move(NTAP == 9 ? tap <= 4 : true);                   /* from the last row of the K-tile before */ \
        xs_src = tap == 0 ? rs_ : xs_src; xs_dst = tap == 0 ? rd_ : xs_dst;                     \
        if constexpr (NTAP == 4) { xr_a = (tap & 1) ? roff_o : roff_e; xr_b = (tap & 1) ? roff_e : roff_o; } \
    }
#define XS_TO_ROW_B xs_move(NTAP == 9 ? true : tap != 3);
#define XS_TO_ROW_C xs_move(tap != 3);
#define WS_HOOK ws_p = (ch2 >= CPT ? wb_n : wb_c) + (size_t)(tap2 * CPT + (ch2 >= CPT ? ch2 - CPT : ch2)) * (256 * 64);
// The four weight DMA instructions of K-tile kt+2 are spread two per phase with the halo rows: WA's behind the row DMAs of phases 2
// and 3, WB's in phase 4 (every target half was last read two or more phases earlier; six - seven with four taps - instructions
// per K-tile as before, so the counted waits retire each half-tile at the same barrier as before).  In-kernel stamps of the form
// with all four in phase 4 (profiles/r05_c256_stamps_final.txt): load segments 204 / 158 / 60 / 352 cycles against a 256-cycle MFMA
// segment of the partner wave.  Same box, three rounds: heads.conv_d1 3.208 -> 3.164 ms, the transposed convs 0.455 -> 0.446.
#define WSPLIT_P2 WS_HOOK stage_w1(0, 0, ws_p, sp);
#define WSPLIT_P3 stage_w1(0, 1, ws_p, sp);
#define WSPLIT_P4 stage_w(1, ws_p, sp);
#define WAD_SET(SP) { const uint32_t wb_ = lds_base + (uint32_t)(SP) * (2 * HALF_ELEMS * 2); wad_c[0] = wb_ + wrow_b0; wad_c[1] = wb_ + wrow_b1; }
#define STEP_H(VM, FIRST, LAST)                                                                 \
    {                                                                                           \
        const uint32_t wad0 = wad_c[0], wad1 = wad_c[1];   /* weight-operand addresses in ring buffer sp: computed a K-tile ahead (phase 4) */ \
        HALO_XADDR_NOW                                                                          \
        int tap1 = tap + 1, ch1 = ch;                                                           \
        if (tap1 == NTAP) { tap1 = 0; ch1 = ch + 1; }                                            \
        int tap2 = tap1 + 1, ch2 = ch1;                                                         \
        if (tap2 == NTAP) { tap2 = 0; ch2 = ch1 + 1; }                                           \
        const bool xn = ch + 1 == CPT;               /* the halo staged now is the next tile's chunk 0 */ \
        LOAD_X_H(0)                                                                             \
        LOAD_W_H(wa, 0)                                                                         \
        SEG_SYNC_H(VM, 0)                                                                       \
        MMA_H(0, 0, wa, FIRST, 1, 0)                                                            \
        LOAD_W_H(wb, 1)                                                                         \
        XS_TO_ROW_A                                                                             \
        xs_issue(xr_a);                                                                         \
        WSPLIT_P2                                                                               \
        SEG_SYNC_H(VM, 5)                                                                       \
        MMA_H(0, 1, wb, FIRST, 1, 5)                                                            \
        LOAD_X_H(1)                                                                             \
        XS_TO_ROW_B                                                                             \
        xs_issue(xr_b);                                                                         \
        WSPLIT_P3                                                                               \
        SEG_SYNC_H(VM, 10)                                                                      \
        MMA_H(1, 1, wb, FIRST, 1, 10)                                                           \
        WSPLIT_P4                                                                               \
        if constexpr (NTAP == 4) { XS_TO_ROW_C xs_issue(xr_a); }                                \
        WAD_SET(sp ^ 1) asm volatile("" : "+v"(wad_c[0]), "+v"(wad_c[1]));                      \
        HALO_XADDR_NEXT HALO_XADDR_PIN   /* phase 4's load segment has no operand reads: the next K-tile's fragment addresses */ \
        SEG_SYNC_H(VM, 15)                                                                      \
        MMA_H(1, 0, wa, FIRST, !(LAST), 15)                                                     \
        HSTAMP_NEXT                                                                             \
        sp ^= 1;                                                                                \
        if (++tap == NTAP) { tap = 0; ++ch; hpar ^= 1; }                                         \
    }

    uint32_t xu_c, xw_c;                    // this lane's pixel-operand fragment addresses of the NEXT K-tile to run (HALO_XADDR_NEXT)
    uint32_t xr_a = roff_e, xr_b = roff_o;
    uint32_t wad_c[2];
    WAD_SET(0)
    const f16* ws_p = wb_c;
    HALO_XADDR(ht.taps[gi_c], 0, 0, xu_c, xw_c)
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
        // successor tile (its ticket was drawn a tile ago): its weights are staged from K-tile T-2 on, its
        // halo during the last chunk - which is the first K-tile already when cin = 64
        live_n = draw;
        if (live_n) locate(vnext, xb_n, wb_n, gi_n, nt_n, n_n, ty_n, tx_n);
        // the accumulators start at the bias (fp32, from LDS) instead of at zero: the epilogue then has no add (64 packed adds per
        // lane and tile less, in the one part of a tile that no MFMA overlaps)
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
        const unsigned long long tapword = ht.taps[gi_c], tapword_n = ht.taps[gi_n];     // (both in SGPRs for the whole tile: no load, no branch in the K loop)
        int ch = 0, tap = 0;
        // (the counted waits are immediates: 6 DMA instructions per K-tile with nine taps, 7 with four)
        if constexpr (NTAP == 9) STEP_H(22, 1, 0) else STEP_H(23, 1, 0)                    // + the previous tile's 16 stores
        if (wave == 0) {
            if constexpr (NTAP == 9) asm volatile("s_waitcnt vmcnt(6)" : "+v"(ticket) : : "memory");   // the atomic is older than this K-tile's DMAs
            else asm volatile("s_waitcnt vmcnt(7)" : "+v"(ticket) : : "memory");
            if (lane == 0) {
                lds_ticket[tpar] = ticket;
                if (draw && ticket == last_draw) tile_ctr[xcd] = 0u;
            }
        }
        for (int kt = 1; kt < T - 1; ++kt) { if constexpr (NTAP == 9) STEP_H(6, 0, 0) else STEP_H(7, 0, 0) }
        if constexpr (NTAP == 9) STEP_H(6, 0, 1) else STEP_H(7, 0, 1)
        // last MFMA segment had no trailing barrier: waves 0-3 take it before their epilogue, waves 4-7 after
        if (wave < 4) __builtin_amdgcn_s_barrier();

        // ---- epilogue: 16 independent 16-byte stores, no loads from global memory
        {
            const ConvGroupArgs& g = a.g[gi_c];
            const int cbase = g.out_coff + nt_c * 256 + wc * 32 + so_ch;
            f16x4 hq[2][4][2][2];          // fp16 results [pixel half][pixel tile][channel half][channel tile]
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                size_t opix[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int oy = (ty_c * 8 + i * 4 + wp * 2 + (p >> 1)) * a.out_scale + g.out_oy;
                    const int ox = (tx_c * 32 + (p & 1) * 16 + frow) * a.out_scale + g.out_ox;
                    opix[p] = ((size_t)(n_c * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + cbase;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        uint32_t u[2][2];
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc) {
                            const f32x4 vv = acc[i][j][cc][p];                      // (bias: the accumulators started at it)
                            f16x4 h = {(f16)vv[0], (f16)vv[1], (f16)vv[2], (f16)vv[3]};
                            h = __builtin_elementwise_max(h, lo4);
                            if (STATS) hq[i][p][j][cc] = h;
                            __builtin_memcpy(u[cc], &h, 8);
                        }
                        const auto s0 = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
                        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
#ifdef HALO_T_NOSTORE
                        if (a.relu == 12345)        // (timing only: the epilogue without its 16 stores per lane.  Round 5, same box: heads.conv_d1
                                                    //  3.76 -> 3.62 ms (2 GB written), the 96 x 320 transposed convs 0.523 -> 0.510: what hiding the stores could buy)
#endif
                        *(u32x4*)((f16*)a.out + opix[p] + j * 128) = o;
                    }
            }
            if (STATS) {
                // Spatial-softmax partials of this wave's 128 pixels x 64 channels (keypoint_fpn_fusion.py:67: the
                // fusion then needs no pass over the map to find them): per channel the max and sum exp(v - max)
                // of the STORED fp16 values.  Lane (frow, fk) holds pixels (i, p, frow) x channels (j, cc, fk*4+e):
                // reduce over (i, p) in the lane, over frow with row rotations (DPP); lanes frow == 0 write.
                // Layout [image][chunk = (group * tiles_per_image + tile) * 2 + wp][256 channels][max, sum].
#define ROW_ROR_MAX(N)                                                                                           \
    {                                                                                                            \
        const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mu[d], 0x120 + (N), 0xf, 0xf, false);  \
        f16x2 a_, b_;                                                                                            \
        __builtin_memcpy(&a_, &mu[d], 4); __builtin_memcpy(&b_, &o_, 4);                                         \
        a_ = __builtin_elementwise_max(a_, b_);                                                                  \
        __builtin_memcpy(&mu[d], &a_, 4);                                                                        \
    }
#define ROW_ROR_ADD(N) S[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, S[e]), 0x120 + (N), 0xf, 0xf, false));
                const size_t chunk = (size_t)(gi_c * tpi + ty_c * tiles_x + tx_c) * 2 + wp;
                float* const srow = stat_out + (((size_t)n_c * (groups * tpi * 2) + chunk) * 256 + wc * 32 + fk * 4) * 2;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        f16x4 mx = hq[0][0][j][cc];
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int p = 0; p < 4; ++p) mx = __builtin_elementwise_max(mx, hq[i][p][j][cc]);
                        uint32_t mu[2];
                        __builtin_memcpy(mu, &mx, 8);
#pragma unroll
                        for (int d = 0; d < 2; ++d) { ROW_ROR_MAX(8) ROW_ROR_MAX(4) ROW_ROR_MAX(2) ROW_ROR_MAX(1) }
                        __builtin_memcpy(&mx, mu, 8);
                        const f32x4 M = {(float)mx[0], (float)mx[1], (float)mx[2], (float)mx[3]};
                        // exp(h - M) = exp2(h * log2(e) - M * log2(e)): one mixed-precision fma (fp16 source, fp32 result: no separate
                        // conversion) + v_exp_f32 + the add per element instead of cvt, sub, mul, exp, add
                        const float L2E = 1.4426950408889634f;
                        const f32x4 nMl = {-M[0] * L2E, -M[1] * L2E, -M[2] * L2E, -M[3] * L2E};
                        float S[4] = {0.f, 0.f, 0.f, 0.f};     // (scalars: update_dpp on ext-vector elements was miscompiled)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int p = 0; p < 4; ++p) {
                                const f16x4 h = hq[i][p][j][cc];
#pragma unroll
                                for (int e = 0; e < 4; ++e) S[e] += __builtin_amdgcn_exp2f(__builtin_fmaf((float)h[e], L2E, nMl[e]));
                            }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { ROW_ROR_ADD(8) ROW_ROR_ADD(4) ROW_ROR_ADD(2) ROW_ROR_ADD(1) }
                        if (frow == 0) {
                            float* o = srow + (j * 128 + cc * 16) * 2;
                            *(f32x4*)o = (f32x4){M[0], S[0], M[1], S[1]};
                            *(f32x4*)(o + 4) = (f32x4){M[2], S[2], M[3], S[3]};
                        }
                    }
            }
        }
        if (wave >= 4) __builtin_amdgcn_s_barrier();
#ifdef C256_STAMPS
        ++stamp_tile; stamp_kt = 0;
#endif
        if (!live_n) break;
        vnext = __builtin_amdgcn_readfirstlane(lds_ticket[tpar]);
        tpar ^= 1;
        xb_c = xb_n;
        wb_c = wb_n; gi_c = gi_n; nt_c = nt_n; n_c = n_n; ty_c = ty_n; tx_c = tx_n;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();           // pair the extra barrier of waves 4-7
#ifdef C256_STAMPS
    if ((blockIdx.x == 0 || blockIdx.x == 101) && (wave & 3) == 0 && groups == 4 && NTAP == 9) {       // (the head conv d1)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        unsigned long long* dst = (unsigned long long*)(tile_ctr + 1024) + 2048 + (blockIdx.x ? 1 : 0) * 1024 + (wave >> 2) * 512;
        for (int i = lane; i < H_STAMP_KT * H_STAMP_N; i += 64) dst[i] = (&lds_stamp[wave >> 2][0][0])[i];
        if (lane == 16) dst[H_STAMP_KT * H_STAMP_N + 16] = (unsigned long long)T;
    }
#endif
}

// Eligibility: every tap within +-1 pixel, stride 1, 8 x 32 tiles cover the output exactly.
bool conv_mfma256_halo_supported(const ConvKArgs& a, int groups, HaloTaps* ht) {
    if (a.res || a.in_stride != 1 || a.in_P < 1 || (a.ntaps != 9 && a.ntaps != 4)) return false;      // the kernel is instantiated for 9 and 4 taps
    if (a.Wm % 32 || (a.HmWm / a.Wm) % 8 || a.M % a.HmWm) return false;
    const int pitch = a.in_Wp * a.in_C;
    for (int g = 0; g < RT_MAX_GROUPS; ++g) ht->taps[g] = 0;
    for (int g = 0; g < groups; ++g)
        for (int t = 0; t < a.ntaps; ++t) {
            const int off = a.g[g].tap_off[t];
            int dy = (off + pitch + pitch / 2) / pitch - 1;          // round(off / pitch) for |dy| <= 1
            const int rem = off - dy * pitch;
            if (rem % a.in_C) return false;
            const int dx = rem / a.in_C;
            if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
            ht->taps[g] |= (unsigned long long)((dy + 1) | ((dx + 1) << 2)) << (4 * t);
        }
    return true;
}

hipError_t launch_conv_mfma256_halo(const ConvKArgs& a, const HaloTaps& ht, int groups, int nbias, int cu_count, unsigned int* tile_ctr, float* stat_out, hipStream_t s) {
    const int per_xcd = cu_count / 8;
    const int one_list = (long long)a.MT * a.NT * groups <= cu_count ? 1 : 0;      // (launch_conv_mfma256)
#ifdef HALO_T_NOSTATS
    stat_out = nullptr;     // (timing only: what do the softmax partials cost their producers?  round 5, same box: 0.504 / 0.510 / 0.513 ms
                            //  with them, 0.467 / 0.475 / 0.486 without = 2.3 us of a 31 us tile, 128 v_exp_f32 per lane)
#endif
    const dim3 grid(per_xcd * 8, 1, 1), block(512, 1, 1);
    if (a.ntaps == 9) {
        if (stat_out) hipLaunchKernelGGL((conv_mfma256_halo_kernel<1, 9>), grid, block, 0, s, a, ht, groups, nbias, tile_ctr, stat_out, one_list);
        else hipLaunchKernelGGL((conv_mfma256_halo_kernel<0, 9>), grid, block, 0, s, a, ht, groups, nbias, tile_ctr, stat_out, one_list);
    } else if (a.ntaps == 4) {
        if (stat_out) hipLaunchKernelGGL((conv_mfma256_halo_kernel<1, 4>), grid, block, 0, s, a, ht, groups, nbias, tile_ctr, stat_out, one_list);
        else hipLaunchKernelGGL((conv_mfma256_halo_kernel<0, 4>), grid, block, 0, s, a, ht, groups, nbias, tile_ctr, stat_out, one_list);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}
