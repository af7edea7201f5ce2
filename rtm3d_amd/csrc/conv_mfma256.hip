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

template <int RES>
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
    // All loads (bias, residual) are issued before the first store and the stores form one
    // dependency-free run: a load between two stores would make the compiler wait vmcnt(0), i.e. for
    // the previous store's acknowledgement as well, and serialise 32 round trips per thread.
    f32x4 bv[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c)
            bv[j][c] = *(const f32x4*)(a.bias + g.bias_off + ntile * 256 + j * 128 + wc * 32 + c * 16 + fk * 4);
    const float lo = a.relu ? 0.f : -__builtin_inff();
    const int cbase = ntile * 256 + wc * 32 + fk * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        size_t opix[4];
        f16x4 rv[4][2][2];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // rows past M were staged from pixel M-1 (see xoff), so they hold its result: storing them
            // to its address again is a same-value write and keeps the epilogue branch-free
            int m = mtile * 256 + i * 128 + wp * 64 + p * 16 + frow;
            m = m < a.M ? m : a.M - 1;
            const int n = m / a.HmWm, rem = m - n * a.HmWm;
            const int y = rem / a.Wm, x = rem - y * a.Wm;
            const int oy = y * a.out_scale + g.out_oy, ox = x * a.out_scale + g.out_ox;
            opix[p] = ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + g.out_coff + cbase;
            if (RES) {
                const f16* rp = a.res + ((size_t)(n * a.res_Hp + oy + a.res_P) * a.res_Wp + ox + a.res_P) * a.res_C + g.res_coff + cbase;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int c = 0; c < 2; ++c) rv[p][j][c] = *(const f16x4*)(rp + j * 128 + c * 16);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    f32x4 v = acc[i][j][c][p] + bv[j][c];
                    if (RES) {
                        const f16x4 r = rv[p][j][c];
                        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
                    }
                    const f16x4 h = {(f16)fmaxf(v[0], lo), (f16)fmaxf(v[1], lo), (f16)fmaxf(v[2], lo), (f16)fmaxf(v[3], lo)};
                    *(f16x4*)((f16*)a.out + opix[p] + j * 128 + c * 16) = h;
                }
    }
}

// ------------------------------------------------------------------------------------------------
// Persistent form (default when the K loop has >= 4 tiles).
// One workgroup per CU draws output tiles from its XCD's list (ticket counter):
//   * the half-tile DMA pipeline runs straight across tile boundaries: the last two K-tiles of a
//     tile already stage the first two of the next one (descriptor "n"), so there is no prologue,
//     no drain and no workgroup launch between tiles;
//   * the bias vector sits in LDS (a VMEM load in the epilogue would make the compiler drain vmcnt);
//     the epilogue is 16 global_store_dwordx4
//     per thread (v_permlane16_swap pairs the two 16-channel MFMA tiles so a lane owns 8 consecutive
//     channels) with no wait behind them: the first K-tile of the next tile runs its four phases
//     with s_waitcnt vmcnt(8+16) (the 16 stores are younger than the DMA it needs);
//   * pixel -> (image, row, column) uses a float-estimate division (quotients are tiny).
// Measured (in-kernel stamps, bs=32): the one-tile kernel spends 1.8 us in its prologue, 5.4 us in
// its epilogue and ~3.8 us between workgroups per tile.  That is 16 % of a 36-K-tile head tile but
// 35-60 % of the 16- and 4-K-tile tiles of the transposed-conv phases and 1x1 convs, which is where
// this form pays: -12 % and -25 % on those; the head convs stay at ~1.2 PFLOP/s in either form - on
// non-zero data they run at the rate the chip's power management allows (1.9 GHz in-kernel clock;
// the same binary on all-zero activations: 1.64 PFLOP/s).
#define CONV256_MAX_BIAS 2048
// LDS-DMA as inline asm (m0 = LDS base of the wave's 1 KB run, one 16-byte piece per lane).  The compiler
// must not know these write LDS: its waitcnt pass treats every visible ds_read as possibly aliasing a pending
// LDS-DMA and puts s_waitcnt vmcnt(0) in front of it, draining the DMA ring in every phase; ordering against
// the DMA is SEG_SYNC_N's counted vmcnt + barrier.  The operand reads stay ordinary loads so that the hazard
// recogniser sees them (with the reads hidden in asm instead, a renamed accumulator's old registers can be
// handed to a ds_read whose data lands before a queued MFMA has read them as SrcC: conv_mfma256_halo.hip).
#define DMA16 RT_DMA16                              // common.h: the one LDS-DMA definition
#define DMA16_NT RT_DMA16_NT
#define DMA16_SBASE RT_DMA16_SBASE
#define DMA16_SBASE_NT RT_DMA16_SBASE_NT
#define LDS_F16X8(byte_addr) (*(const LDS_AS f16x8*)(uintptr_t)(byte_addr))
#define LOAD_X_N(SLOT)                                                                      \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                         \
        xf[p][0] = LDS_F16X8(xaddr0 + (SLOT) * HALF_ELEMS * 2 + p * 2048);                  \
        xf[p][1] = LDS_F16X8(xaddr1 + (SLOT) * HALF_ELEMS * 2 + p * 2048);                  \
    }
#define LOAD_W_N(dstf, SLOT)                                                                \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                         \
        dstf[c][0] = LDS_F16X8(waddr0 + (SLOT) * HALF_ELEMS * 2 + c * 2048);                \
        dstf[c][1] = LDS_F16X8(waddr1 + (SLOT) * HALF_ELEMS * 2 + c * 2048);                \
    }
// (timing-only switches, never in the product build: -DC256_T_NOVM drops the counted vmcnt wait, -DC256_T_NOX / -DC256_T_NOW
// skip the pixel / weight staging of the persistent kernel, -DC256_T_WSAME makes every tile stream channel tile 0's weights:
// results are wrong, the per-op time says what the piece costs)
#ifdef C256_T_NOVM
#define SEG_SYNC_N(VM)                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    __builtin_amdgcn_s_barrier();                                                           \
    __builtin_amdgcn_sched_barrier(0);
#else
#define SEG_SYNC_N(VM)                                                                      \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");                       \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    __builtin_amdgcn_s_barrier();                                                           \
    __builtin_amdgcn_sched_barrier(0);
#endif
#define MMA_N(i, j, wfrag, FIRST, TAILBAR)                                                  \
    __builtin_amdgcn_s_setprio(1);                                                          \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                        \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                       \
            _Pragma("unroll") for (int p = 0; p < 4; ++p)                                   \
                acc[i][j][c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag[c][kk], xf[p][kk], acc[i][j][c][p], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    if (TAILBAR) __builtin_amdgcn_s_barrier();                                              \
    __builtin_amdgcn_sched_barrier(0);
// Diagnostic build only (-DC256_STAMPS; tools/gpu_c256_stamps.sh): waves 0 and 4 of every workgroup stamp s_memtime behind every
// wait and barrier of the K loop of the SECOND tile they process (5 stamps per phase: operand reads issued and returned | counted
// vmcnt wait | barrier | MFMAs issued | tail barrier) and around that tile's epilogue; workgroups 0 and 101 dump theirs behind the
// ticket counters (rtm3d_ctx_debug_read_words).  A stamp costs ~100 cycles (s_memtime + its wait): read the phases against each
// other, not as absolute times.  The product build contains none of this.
//
// What the round-5 stamps and timing-only builds said about heads.conv_d6 (profiles/r05_c256_*.txt):
//   * MFMA segments take their 256 cycles; what an interval waits for is the partner wave's LOAD segment (12 / 4 / 8 / 0 operand
//     reads + DMA issue: ~340 / 220 / 420 / 100 cycles before the change below, the two middle ones inflated by the tap-table
//     lookup: a scalar load and its wait in front of the DMA) -> the lookup moved a K-tile ahead: d6 3.73 -> 3.58 ms;
//   * dropping the counted vmcnt wait changes nothing (the DMA has landed by then: latency is not what a phase waits for);
//   * reads rebalanced to 8 / 0 / 8 / 0 (WB fragments read beside the MFMAs of phase 1, the next K-tile's WA fragments behind those
//     of phase 4): no change in time - and unsound as built: a half-tile is retired by each wave's OWN counted wait, waves 4-7 run
//     one barrier behind, so a read must stay two barriers behind the retiring wait (the round-1 rule "retired one phase before its
//     first ds_read"); it would need the staging order XB(t+1) | XA(t+2) | WA(t+2) | WB(t+2).  Not kept;
//   * the matrix pipe is 70 % busy at 1.65 GHz under the profiler, against 65 % at 1.71 GHz before: the chip trades the recovered
//     cycles for clock (power cap), as the microbenchmark predicts (tools/microbench/mfma_rate.hip: the same loop with 64 KB of
//     LDS-DMA per K-tile from L2 holds 1.69 GHz at 85 % = 1.44 PFLOP/s on random data; without any staging 1.70-1.76).
#ifdef C256_STAMPS
#define C256_STAMP_KT 20
#define C256_STAMP_N 20
#define STAMP(k) if ((wave & 3) == 0 && stamp_tile == 1 && t < C256_STAMP_KT) { unsigned long long ts_; \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory"); if (lane == 0) lds_stamp[wave >> 2][t][k] = ts_; }
#define TSTAMP(k) if ((wave & 3) == 0 && stamp_tile < 4) { unsigned long long ts_; \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory"); if (lane == 0) lds_tstamp[wave >> 2][stamp_tile][k] = ts_; }
#else
#define STAMP(k)
#define TSTAMP(k)
#endif
#ifdef C256_STAMPS
#define SEG_SYNC_S(VM, K0)                                                                  \
    STAMP((K0) + 0)                                                                         \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");                       \
    STAMP((K0) + 1)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    __builtin_amdgcn_s_barrier();                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    STAMP((K0) + 2)
#define MMA_S(i, j, wfrag, FIRST, TAILBAR, K0)                                              \
    __builtin_amdgcn_s_setprio(1);                                                          \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                        \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                       \
            _Pragma("unroll") for (int p = 0; p < 4; ++p)                                   \
                acc[i][j][c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag[c][kk], xf[p][kk], acc[i][j][c][p], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    STAMP((K0) + 3)                                                                         \
    if (TAILBAR) __builtin_amdgcn_s_barrier();                                              \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    STAMP((K0) + 4)
#else
#define SEG_SYNC_S(VM, K0) SEG_SYNC_N(VM)
#define MMA_S(i, j, wfrag, FIRST, TAILBAR, K0) MMA_N(i, j, wfrag, FIRST, TAILBAR)
#endif
// Where WB's DMA is issued.  Phase 1 of the K-tile before its use (12 operand reads in the same load segment), or - with
// -DC256_WB_IN_P4 - phase 4 of the K-tile before that (no operand reads there; legal with the same counted waits: its target
// half was last read in phase 2).  Same box, round 5: the halo kernel gains 3 % from the move (heads.conv_d1 3.64 -> 3.54 ms),
// this kernel loses 0.7 % (heads.conv_d6 3.547 -> 3.572, backbone 2.31 -> 2.33): here phase 4 already carries the tap-table
// lookup and every phase stages two DMAs, so the move only unbalances them.  Kept in phase 1.  (Likewise XA from phase 3 into
// phase 4 - legal with the same counted wait: heads.conv_d6 3.49 -> 3.57, the level-4 convs +2.5 %; WB behind XB in phase 2, with
// phase 1's wait tightened to vmcnt(4): 3.380 -> 3.391, the folded 96 x 320 neck conv +1.5 %.)
// Round 5, after the stamps of the final form (profiles/r05_c256_stamps_final.txt: load segments 346 / 170 / 205 / 255 cycles against the
// partner wave's 256-cycle MFMA segment): WB's two DMA instructions are split - one stays behind phase 1's twelve reads, the other
// goes behind XB in phase 2, the lightest segment.  Phase 1's wait is then vmcnt(5) (the five youngest: its own WB instruction,
// WA x 2, XA x 2), which retires XB and both halves of WB one barrier before their first read, as before; the other three
// waits stay vmcnt(8).  Same box, three rounds: heads.conv_d6 3.430 -> 3.394 ms.  (Stamps after it: 339 / 168 / 217 / 272 - phase 1 is its
// twelve reads, not its DMA; reading the k-half-1 fragments - six of the twelve - behind the barrier, under the first eight MFMAs: 3.460 -> 3.496.)
#if !defined(C256_WB_IN_P4)
#define STAGE_WB_P1 stage_w1(SLOT_WB, 0, t + 1, sp ^ 1);
#define STAGE_WB_P2 stage_w1(SLOT_WB, 1, t + 1, sp ^ 1);
#define STAGE_WB_P4
#define VM1_8 5
#define VM1_24 21
#else
#define STAGE_WB_P1
#define STAGE_WB_P2
#define STAGE_WB_P4 stage(SLOT_WB, t + 2, sp, 0);
#define VM1_8 8
#define VM1_24 24
#endif
#define OPA_SET(SP)                                                                         \
    {                                                                                       \
        const uint32_t bufb_ = lds_base + (uint32_t)(SP) * (BUF_ELEMS * 2);                 \
        opa[0] = bufb_ + xrow_b0; opa[1] = bufb_ + xrow_b1; opa[2] = bufb_ + wrow_b0; opa[3] = bufb_ + wrow_b1; \
    }
#define STEP_N(VM, VM1, FIRST, LAST)                                                          \
    {                                                                                       \
        /* this lane's operand addresses in the K-tile's buffer: computed a K-tile ahead, in phase 4's load segment (no operand reads there) */ \
        const uint32_t xaddr0 = opa[0], xaddr1 = opa[1], waddr0 = opa[2], waddr1 = opa[3];  \
        LOAD_X_N(SLOT_XA)                                                                   \
        LOAD_W_N(wa, SLOT_WA)                                                               \
        STAGE_WB_P1                                                                         \
        SEG_SYNC_S(VM1, 0)                                                                  \
        MMA_S(0, 0, wa, FIRST, 1, 0)                                                        \
        LOAD_W_N(wb, SLOT_WB)                                                               \
        stage(SLOT_XB, t + 1, sp ^ 1, koff1);                                               \
        STAGE_WB_P2                                                                         \
        SEG_SYNC_S(VM, 5)                                                                   \
        MMA_S(0, 1, wb, FIRST, 1, 5)                                                        \
        LOAD_X_N(SLOT_XB)                                                                   \
        stage(SLOT_XA, t + 2, sp, koff2);                                                   \
        SEG_SYNC_S(VM, 10)                                                                  \
        MMA_S(1, 1, wb, FIRST, 1, 10)                                                       \
        /* the lookup for K-tile position t + 3 sits in this, the shortest load segment (no operand reads): its latency ends */ \
        /* under the segment's own wait, and it is first used two load segments later */      \
        int kchan3;                                                                         \
        int kraw3 = koff_parts(t + 3, kchan3);                                              \
        stage(SLOT_WA, t + 2, sp, 0);                                                       \
        OPA_SET(sp ^ 1)                                                                     \
        asm volatile("" : "+v"(opa[0]), "+v"(opa[1]), "+v"(opa[2]), "+v"(opa[3]));          \
        STAGE_WB_P4                                                                         \
        SEG_SYNC_S(VM, 15)                                                                  \
        asm volatile("" : "+s"(kraw3));           /* first use of the loaded word: behind the segment's wait */ \
        const int koff3 = kraw3 + kchan3;                                                   \
        MMA_S(1, 0, wa, FIRST, !(LAST), 15)                                                 \
        koff1 = koff2; koff2 = koff3;                                                       \
        sp ^= 1;                                                                            \
    }

// XNT = 1: a 1x1 conv with one channel tile reads every input byte once - its pixel operand goes through a non-temporal DMA
// (see conv_mfma.hip).  A template parameter, not a run-time flag: as a flag it cost the head convs 12 SGPRs and 20 B of scratch.
template <int RES, int XNT>
__global__ __launch_bounds__(512) void conv_mfma256_persistent_kernel(const ConvKArgs a, const int groups, const int nbias, unsigned int* tile_ctr, const int one_list) {
    __shared__ __attribute__((aligned(16))) f16 lds[2 * BUF_ELEMS + HALF_ELEMS];
    __shared__ __attribute__((aligned(16))) float lds_bias[CONV256_MAX_BIAS + 4];   // + two ticket words
#ifdef C256_STAMPS
    __shared__ unsigned long long lds_stamp[2][C256_STAMP_KT][C256_STAMP_N];
    __shared__ unsigned long long lds_tstamp[2][4][4];
    int stamp_tile = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wc = wave >> 1;
    const int T = a.ksteps;
    // the op's whole bias vector lives in LDS and is read with ds_read in the epilogue, so that no
    // vector-memory load sits between the DMA stream and the stores
    for (int i = tid; i < nbias; i += 512) lds_bias[i] = a.bias[i];
    __syncthreads();

    // tile list of this workgroup: same XCD-contiguous order the one-tile kernel gets from the
    // dispatcher, position v -> (group, pixel tile, channel tile)
    // one_list (round 6; launches with no more tiles than CUs, e.g. DLA level 4: 240 tiles): ONE list and ONE counter for the whole
    // launch, and one workgroup per CU whatever the tile count.  Such a launch is a single round, and with a list per XCD it stays
    // one only while no XCD has more than (32 - its tiles) CUs held by another stream's workgroups - the previous batch's 3D decode
    // holds a few CUs for milliseconds, and every level-4 conv that met three of them on one XCD paid a second round (65 -> 100-110 us
    // in the kernel trace, tools/gpu_trace_variants.sh).  With one list any 240 free CUs of the chip make it one round; what the
    // per-XCD lists buy - neighbouring tiles through one L2 - matters for launches that stream many tiles per CU, not for these.
    const int xcd = one_list ? 0 : blockIdx.x & 7, per_xcd = one_list ? gridDim.x : gridDim.x >> 3;
    const int chunk = one_list ? a.MT : (a.MT + 7) >> 3;
    int mt_here = a.MT - xcd * chunk;
    mt_here = mt_here < 0 ? 0 : (mt_here > chunk ? chunk : mt_here);
    const int jbs = mt_here * a.NT;
    const int vtotal = jbs * groups;
    // Tiles are handed out by a per-XCD ticket counter, so a workgroup that starts late (its CU was still
    // held by another stream's waves) or runs slower simply takes fewer tiles.  Every workgroup draws
    // tickets until it gets one >= vtotal, i.e. vtotal + per_xcd draws per XCD and launch; the last draw
    // puts the counter back to zero for the next launch.
    int* const lds_ticket = (int*)(lds_bias + CONV256_MAX_BIAS);
    const int last_draw = vtotal + per_xcd - 1;
    // Tickets are drawn two tiles ahead: slot (i & 1) of lds_ticket holds tile i's, and is refilled with
    // tile i+2's while tile i runs.
    // The FIRST ticket now, the second one behind the prologue's wait (round 6): a launch with about as many tiles as workgroups
    // is one round only if every workgroup's first draw comes before anybody's second - drawn back to back, the workgroups that
    // start a microsecond early take two tiles each and the late ones find the list empty (seen as 0.108 instead of 0.071 ms on
    // a level-4 conv).  The prologue's operands take 2-3 us to land: by then every workgroup that is going to start has drawn.
    if (tid == 0) {
        const int t0 = (int)atomicAdd(&tile_ctr[xcd], 1u);
        if (t0 == last_draw) tile_ctr[xcd] = 0u;
        lds_ticket[0] = t0;
    }
    __syncthreads();
    const int v = __builtin_amdgcn_readfirstlane(lds_ticket[0]);
    if (v >= vtotal) return;

    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;
    const int rr = tid >> 3, cs = tid & 7;
    const float rcp_hw = 1.0f / (float)a.HmWm, rcp_w = 1.0f / (float)a.Wm;
    uint32_t xo_c[2][2], xo_n[2][2];        // DMA source offsets of the current / next tile
    const f16 *wb_c, *wb_n;
    int gi_c, gi_n, mt_c, mt_n, nt_c, nt_n;
    auto locate = [&](int vv, uint32_t (&xo)[2][2], const f16*& wb, int& gi, int& mt, int& nt) {
        gi = vv / jbs;
        const int jb = vv - gi * jbs;
        const int q = jb / a.NT;
        nt = jb - q * a.NT;
        mt = xcd * chunk + q;
        const ConvGroupArgs& g = a.g[gi];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int m = mt * 256 + h * 128 + i * 64 + rr;
                m = m < a.M ? m : a.M - 1;
                const int n = div_small_q(m, a.HmWm, rcp_hw), rem = m - n * a.HmWm;
                const int y = div_small_q(rem, a.Wm, rcp_w), x = rem - y * a.Wm;
                const uint32_t pix = (uint32_t)((n * a.in_Hp + y * a.in_stride + a.in_P) * a.in_Wp + x * a.in_stride + a.in_P);
                xo[h][i] = (pix * (uint32_t)a.in_C + (uint32_t)g.in_coff + (uint32_t)((cs ^ (rr & 7)) * 8)) * 2u;      // BYTES
            }
#ifdef C256_T_WSAME
        wb = a.wgt + g.w_off;        // timing-only: every tile streams channel tile 0's weights (L2-resident for sure)
#else
        wb = a.wgt + g.w_off + (size_t)nt * T * (256 * 64);
#endif
    };
    bool live_n;
    locate(v, xo_c, wb_c, gi_c, mt_c, nt_c);
    // until the real successor is located (after the first K-tile) "n" aliases "c"
    xo_n[0][0] = xo_c[0][0]; xo_n[0][1] = xo_c[0][1]; xo_n[1][0] = xo_c[1][0]; xo_n[1][1] = xo_c[1][1];
    wb_n = wb_c; gi_n = gi_c; mt_n = mt_c; nt_n = nt_c; live_n = false;

    // stage half-tile `slot` of the K-tile kpos steps into the current tile (kpos >= T: next tile;
    // no next tile: same addresses into the dummy slot so the DMA count per phase stays constant)
    // channel / tap offset of the pixel operand of K-tile position kpos (current tile, or the next one's when kpos >= T).  Looked up
    // a K-tile AHEAD, in the load segment of phase 4 - the one without operand reads - (STEP_N: koff3), and carried in SGPRs (koff1 / koff2 = positions t + 1 / t + 2):
    // the table lives in the kernel arguments, and a scalar load issued inside a load segment puts its latency - and, through the
    // shared lgkmcnt counter, that of every ds_read issued before it - in front of the segment's DMA (in-kernel stamps, round 5:
    // +80 ... +120 cycles on the load segments of phases 2 and 3, which the partner wave's 256-cycle MFMA segment then waits for).
    auto koff_parts = [&](int kpos, int& chan) -> int {      // returns the table entry (a scalar LOAD), chan = the chunk's channel offset
        const bool in_cur = kpos < T;
        const int k = in_cur ? kpos : kpos - T;
        const int tap = k / a.cpt;
        chan = (k - tap * a.cpt) * 64;
        return a.g[in_cur ? gi_c : gi_n].tap_off[tap];
    };
    auto koff_of = [&](int kpos) -> int { int c; const int r = koff_parts(kpos, c); return r + c; };
    // Sources are a wave-uniform 64-bit base in SGPRs plus a 32-bit byte offset per lane (xo_* / wvoff, fixed for the tile): no
    // 64-bit vector address arithmetic in the load segments (round 5: two to four VALU instructions per DMA before, two of them
    // 64-bit adds).  The launcher keeps tensors past 4 GB off this kernel.
    const uint32_t wvoff = (uint32_t)tid * 16u;
    // one of the two DMA instructions (i = 0, 1) of weight half-tile `slot` (SLOT_WA / SLOT_WB) of K-tile position kpos
    auto stage_w1 = [&](int slot, int i, int kpos, int par) {
        const bool in_cur = kpos < T;
        const int k = in_cur ? kpos : kpos - T;
        const uint32_t dst0 = lds_base + (uint32_t)((in_cur || live_n) ? par * BUF_ELEMS + slot * HALF_ELEMS : 2 * BUF_ELEMS) * 2u;
        const f16* ws = (in_cur ? wb_c : wb_n) + (size_t)k * (256 * 64) + (slot - 2) * HALF_ELEMS;
        DMA16_SBASE(wvoff, ws + i * 512 * 8, __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)((i * 512 + wave * 64) * 16)));
    };
    auto stage = [&](int slot, int kpos, int par, int koff) {
        const bool in_cur = kpos < T;
        const int k = in_cur ? kpos : kpos - T;
        const uint32_t dst0 = lds_base + (uint32_t)((in_cur || live_n) ? par * BUF_ELEMS + slot * HALF_ELEMS : 2 * BUF_ELEMS) * 2u;
        if (slot < 2) {
#ifdef C256_T_NOX
            return;
#endif
            const f16* xbase = a.in + (ptrdiff_t)koff;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t xo = in_cur ? xo_c[slot][i] : xo_n[slot][i];
                if (XNT) DMA16_SBASE_NT(xo, xbase, __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)((i * 512 + wave * 64) * 16)));
                else DMA16_SBASE(xo, xbase, __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)((i * 512 + wave * 64) * 16)));
            }
        } else {
#ifdef C256_T_NOW
            return;
#endif
            const f16* ws = (in_cur ? wb_c : wb_n) + (size_t)k * (256 * 64) + (slot - 2) * HALF_ELEMS;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                DMA16_SBASE(wvoff, ws + i * 512 * 8, __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)((i * 512 + wave * 64) * 16)));
        }
    };

    f32x4 acc[2][2][2][4];
    const int frow = lane & 15, fk = lane >> 4;
    const int sw0 = ((0 * 4 + fk) ^ (frow & 7)) * 8, sw1 = ((1 * 4 + fk) ^ (frow & 7)) * 8;
    // LDS byte addresses (within a K-tile buffer) of this lane's operand rows, k-halves 0 and 1
    const uint32_t xrow_b0 = (uint32_t)(((wp * 64 + frow) * 64 + sw0) * 2), xrow_b1 = (uint32_t)(((wp * 64 + frow) * 64 + sw1) * 2);
    const uint32_t wrow_b0 = (uint32_t)(((wc * 32 + frow) * 64 + sw0) * 2), wrow_b1 = (uint32_t)(((wc * 32 + frow) * 64 + sw1) * 2);
    // after the permlane swap a lane stores channels [so, so+8) of its wave's 32-channel run
    const int so = (fk & 1) * 16 + (fk >> 1) * 8;
    const f16 lo = a.relu ? (f16)0.f : (f16)(-__builtin_inff());
    const f16x4 lo4 = {lo, lo, lo, lo};

    // ---- prologue (once per workgroup): tile 0 complete, XA(1), WA(1) in flight
    stage(SLOT_XA, 0, 0, koff_of(0)); stage(SLOT_WA, 0, 0, 0); stage(SLOT_WB, 0, 0, 0); stage(SLOT_XB, 0, 0, koff_of(0));
    stage(SLOT_XA, 1, 1, koff_of(1)); stage(SLOT_WA, 1, 1, 0);
#ifndef C256_WB_IN_P4
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#else
    stage(SLOT_WB, 1, 1, 0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#endif
    if (tid == 0) {                                       // the second ticket (see above); published by the barrier below
        const int t1 = (int)atomicAdd(&tile_ctr[xcd], 1u);
        if (t1 == last_draw) tile_ctr[xcd] = 0u;
        lds_ticket[1] = t1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int vnext = __builtin_amdgcn_readfirstlane(lds_ticket[1]);
    if (wave >= 4) __builtin_amdgcn_s_barrier();          // waves 4-7 run one barrier behind

    f16x8 xf[4][2], wa[2][2], wb[2][2];
    int sp = 0;                                           // LDS buffer of the current K-tile
    int tpar = 0;
    int koff1 = koff_of(1), koff2 = koff_of(2);           // T >= 4: both inside the first tile
    bool first_tile = true;
    uint32_t opa[4];
    OPA_SET(0)

    for (;;) {
        // draw the ticket of the tile after next (unless the last draw already came back empty); it is
        // published through LDS after the first K-tile.  Inline asm: were this a returning VMEM op the
        // compiler knows about, it would drain vmcnt(0) at the join; the explicit vmcnt(8) below - the
        // atomic is older than the first K-tile's 8 DMA instructions - is the wait for its result.
        int ticket = vnext;
        const bool draw = vnext < vtotal;
        if (wave == 0 && draw) {
            // one lane only: 64 same-address atomics per draw serialise in L2 (measured: 3x slower kernels)
            const unsigned inc = 1u, off = (unsigned)xcd * 4u;
            unsigned long long saved_exec;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                         : "=&v"(ticket), "=&s"(saved_exec) : "v"(off), "v"(inc), "s"(tile_ctr) : "memory");
        }
        // the accumulators start at the bias (fp32, from LDS) instead of at zero: the epilogue then has no add (64 packed adds per
        // lane and tile less, in the one part of a tile that no MFMA overlaps)
        {
            const float* bp = lds_bias + a.g[gi_c].bias_off + nt_c * 256 + wc * 32 + fk * 4;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x4 b4 = *(const f32x4*)(bp + j * 128 + c * 16);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int p = 0; p < 4; ++p) acc[i][j][c][p] = b4;
                }
        }
        int t = 0;
        TSTAMP(0)
        // The first K-tile's waits count the previous tile's 16 stores as well.  The workgroup's FIRST tile has none before it: with
        // the same count its waits would let the prologue's K-tile-1 half-tiles stay in flight past their first read.
        if (first_tile) STEP_N(8, VM1_8, 1, 0) else STEP_N(24, VM1_24, 1, 0)
        first_tile = false;
        if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(ticket) : : "memory");
            if (lane == 0) {
                lds_ticket[tpar] = ticket;
                if (draw && ticket == last_draw) tile_ctr[xcd] = 0u;
            }
        }
        // successor tile: its first half-tiles are staged from K-tile T-2 of this one
        live_n = draw;
        if (live_n) locate(vnext, xo_n, wb_n, gi_n, mt_n, nt_n);
        for (t = 1; t < T - 1; ++t) STEP_N(8, VM1_8, 0, 0)
        STEP_N(8, VM1_8, 0, 1)
        // The last MFMA segment has no trailing barrier.  Waves 0-3 take it here, before their
        // epilogue, waves 4-7 (one barrier behind) after theirs: otherwise each group would sit at a
        // barrier for the whole of the other group's epilogue (measured: 2 x 2.8 us per tile).
        if (wave < 4) __builtin_amdgcn_s_barrier();
        TSTAMP(1)

        // ---- epilogue of the current tile: no loads, 16 independent 16-byte stores
        {
            const ConvGroupArgs& g = a.g[gi_c];
            const int cbase = g.out_coff + nt_c * 256 + wc * 32 + so;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                size_t opix[4];
                f16x4 rv[4][2][2];       // residual (RES): loaded for the whole half before its first store
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    // rows past M were staged from pixel M-1 and hold its result: a same-value write
                    int m = mt_c * 256 + i * 128 + wp * 64 + p * 16 + frow;
                    m = m < a.M ? m : a.M - 1;
                    const int n = div_small_q(m, a.HmWm, rcp_hw), rem = m - n * a.HmWm;
                    const int y = div_small_q(rem, a.Wm, rcp_w), x = rem - y * a.Wm;
                    const int oy = y * a.out_scale + g.out_oy, ox = x * a.out_scale + g.out_ox;
                    opix[p] = ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + cbase;
                    if (RES) {
                        const f16* rp = a.res + ((size_t)(n * a.res_Hp + oy + a.res_P) * a.res_Wp + ox + a.res_P) * a.res_C
                                        + g.res_coff + nt_c * 256 + wc * 32 + fk * 4;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int c = 0; c < 2; ++c) rv[p][j][c] = *(const f16x4*)(rp + j * 128 + c * 16);
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        uint32_t u[2][2];
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            f32x4 vv = acc[i][j][c][p];                  // (bias: the accumulators started at it)
                            if (RES) {
                                const f16x4 r = rv[p][j][c];
                                vv[0] += (float)r[0]; vv[1] += (float)r[1]; vv[2] += (float)r[2]; vv[3] += (float)r[3];
                            }
                            f16x4 h = {(f16)vv[0], (f16)vv[1], (f16)vv[2], (f16)vv[3]};
                            h = __builtin_elementwise_max(h, lo4);       // ReLU (or -inf) on packed halves
                            __builtin_memcpy(u[c], &h, 8);
                        }
                        // rows (16-lane groups) 1,3 of the c=0 registers <-> rows 0,2 of the c=1 registers
                        const auto s0 = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
                        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        *(u32x4*)((f16*)a.out + opix[p] + j * 128) = o;
                    }
            }
        }
        TSTAMP(2)
        if (wave >= 4) __builtin_amdgcn_s_barrier();
        TSTAMP(3)
#ifdef C256_STAMPS
        ++stamp_tile;
#endif
        if (!live_n) break;
        vnext = __builtin_amdgcn_readfirstlane(lds_ticket[tpar]);
        tpar ^= 1;
        xo_c[0][0] = xo_n[0][0]; xo_c[0][1] = xo_n[0][1]; xo_c[1][0] = xo_n[1][0]; xo_c[1][1] = xo_n[1][1];
        wb_c = wb_n; gi_c = gi_n; mt_c = mt_n; nt_c = nt_n;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();           // pair the extra barrier of waves 4-7
#ifdef C256_STAMPS
    if ((blockIdx.x == 0 || blockIdx.x == 101) && (wave & 3) == 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        unsigned long long* dst = (unsigned long long*)(tile_ctr + 1024) + (blockIdx.x ? 1 : 0) * 1024 + (wave >> 2) * 512;
        for (int i = lane; i < C256_STAMP_KT * C256_STAMP_N; i += 64) dst[i] = (&lds_stamp[wave >> 2][0][0])[i];
        if (lane < 16) dst[C256_STAMP_KT * C256_STAMP_N + lane] = (&lds_tstamp[wave >> 2][0][0])[lane];
        if (lane == 16) dst[C256_STAMP_KT * C256_STAMP_N + 16] = (unsigned long long)T;
    }
#endif
}

struct HaloTaps { unsigned long long taps[RT_MAX_GROUPS]; };
bool conv_mfma256_halo_supported(const ConvKArgs& a, int groups, HaloTaps* ht);
hipError_t launch_conv_mfma256_halo(const ConvKArgs& a, const HaloTaps& ht, int groups, int nbias, int cu_count, unsigned int* tile_ctr, float* stat_out, hipStream_t s);
// conv_mfma256_lattice.hip: the halo form for dilated 3x3 layers (tiles on the row sub-lattice of the dilation)
int conv_mfma256_lattice_dilation(const ConvKArgs& a, int groups);
hipError_t launch_conv_mfma256_lattice(const ConvKArgs& a, int groups, int nbias, int cu_count, unsigned int* tile_ctr, hipStream_t s);

static int device_cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        n = prop.multiProcessorCount;
    }
    return n;
}

// Whether launch_conv_mfma256 sends this conv to the halo-tile kernel (the only one that can emit the
// spatial-softmax partials of its output, `stat_out`).
bool conv_mfma256_uses_halo(const ConvKArgs& a, int groups) {
    int nbias = 0;
    for (int g = 0; g < groups; ++g) nbias = a.g[g].bias_off + a.cout > nbias ? a.g[g].bias_off + a.cout : nbias;
    nbias = (nbias + 255) / 256 * 256;
    HaloTaps ht;
    return a.ksteps >= 4 && !a.res && (a.ntaps == 9 || a.ntaps == 4) && nbias <= 1024 && conv_mfma256_halo_supported(a, groups, &ht);
}

// Whether launch_conv_mfma256 sends this conv to the row-sub-lattice halo kernel (dilated 3x3 layers, conv_mfma256_lattice.hip).
bool conv_mfma256_uses_lattice(const ConvKArgs& a, int groups) {
#ifdef C256_T_NOLATTICE
    return false;
#endif
    int nbias = 0;
    for (int g = 0; g < groups; ++g) nbias = a.g[g].bias_off + a.cout > nbias ? a.g[g].bias_off + a.cout : nbias;
    nbias = (nbias + 255) / 256 * 256;
    const unsigned long long in_bytes = (unsigned long long)(a.M / a.HmWm) * a.in_Hp * a.in_Wp * a.in_C * 2ull;
    return a.ksteps >= 4 && nbias <= 1024 && in_bytes < (1ull << 32) && conv_mfma256_lattice_dilation(a, groups) != 0;
}

hipError_t launch_conv_mfma256(const ConvKArgs& a, int groups, unsigned int* tile_ctr, float* stat_out, hipStream_t s) {
    dim3 block(512, 1, 1);
    int nbias = 0;
    for (int g = 0; g < groups; ++g) nbias = a.g[g].bias_off + a.cout > nbias ? a.g[g].bias_off + a.cout : nbias;
    nbias = (nbias + 255) / 256 * 256;      // channel tiles read whole 256-float runs (the bias array is padded to that)
    // the persistent kernel addresses its pixel operand as SGPR base + 32-bit byte offset per lane: inputs of 4 GB and more
    // (bs = 32 tops out at 2.07 GB) take the one-tile kernel below, which carries 64-bit addresses
    const unsigned long long in_bytes = (unsigned long long)(a.M / a.HmWm) * a.in_Hp * a.in_Wp * a.in_C * 2ull;
    if (a.ksteps >= 4 && nbias <= CONV256_MAX_BIAS && tile_ctr && in_bytes < (1ull << 32)) {
        HaloTaps ht;
#ifdef C256_T_NOHALO9
        if (a.ntaps == 9) {} else          // (same-box A/B only: 3x3 layers on the generic persistent form)
#endif
        if (!a.res && (a.ntaps == 9 || a.ntaps == 4) && nbias <= 1024 && conv_mfma256_halo_supported(a, groups, &ht))
            return launch_conv_mfma256_halo(a, ht, groups, nbias, device_cu_count(), tile_ctr, stat_out, s);
        if (stat_out) return hipErrorInvalidValue;   // only the halo kernel writes softmax partials
        // (-DC256_T_NOLATTICE, same-box A/B only: the dilated head conv on the generic persistent form)
        if (conv_mfma256_uses_lattice(a, groups)) return launch_conv_mfma256_lattice(a, groups, nbias, device_cu_count(), tile_ctr, s);
        // One workgroup per CU even when there are fewer tiles (level 4: 240): with exactly as many workgroups as tiles ONE CU held
        // by another stream's workgroup costs the launch a second round; the spare workgroups find the list empty and leave.
        // Launches of at most one round take one list for the whole chip (see the kernel).  Round 6, same box, pipelined bs=32
        // ms/step (the previous batch's 3D decode beside every forward): grid clamped to the tile count, lists per XCD 12.82-12.87 |
        // + second ticket behind the prologue 12.79-12.86 | + 256 workgroups 12.79-12.84 | + ONE list 12.60-12.63.
        const int per_xcd = device_cu_count() / 8;
        const int one_list = (long long)a.MT * a.NT * groups <= device_cu_count() ? 1 : 0;
        dim3 grid(per_xcd * 8, 1, 1);
        const bool x_once = !a.res && a.ntaps == 1 && a.NT == 1 && a.in_stride == 1;
        if (a.res) hipLaunchKernelGGL((conv_mfma256_persistent_kernel<1, 0>), grid, block, 0, s, a, groups, nbias, tile_ctr, one_list);
        else if (x_once) hipLaunchKernelGGL((conv_mfma256_persistent_kernel<0, 1>), grid, block, 0, s, a, groups, nbias, tile_ctr, one_list);
        else hipLaunchKernelGGL((conv_mfma256_persistent_kernel<0, 0>), grid, block, 0, s, a, groups, nbias, tile_ctr, one_list);
        return hipGetLastError();
    }
    if (stat_out) return hipErrorInvalidValue;       // only the halo kernel writes softmax partials
    const int mt8 = (a.MT + 7) / 8 * 8;
    dim3 grid(mt8 * a.NT, groups, 1);
    if (a.res) hipLaunchKernelGGL((conv_mfma256_kernel<1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_mfma256_kernel<0>), grid, block, 0, s, a);
    return hipGetLastError();
}
