// Wave-cooperative L-BFGS-B with the PUBLISHED subspace step (formk / subsm / formt: the form SciPy's L-BFGS-B 3.0 runs, and the
// form utils/model_utils.py:295-296 therefore runs in the reference): ONE 64-lane wavefront solves ONE object.
//
// History: the round-2 product solver (git fdc3025), replaced as the default by the two-loop direction of lbfgsb_wave.h at the end of
// round 2, back on the product path in round 5 (rtm3d_decode3d_slots form = 1) and THE DEFAULT of every entry since round 6
// (rtm3d_decode3d / rtm3d_decode3d_slots form = RTM3D_SOLVER_PUBLISHED): same algorithm and the same fp64 arithmetic in the same
// order as lb_minimize(direct = 0), bit-identical to it (tests/test_gpu_parity.py::test_decode3d_large_fixture, ..._wave_kernel_equals_scalar_kernel).
//   - all limited-memory matrices live in LDS (8.4 KB per object) instead of per-lane scratch,
//   - independent matrix entries / right-hand sides / vector components are spread over lanes
//     (formk rows and columns, the 55 entries of T and of the (2,2) block, the col right-hand sides
//     of the triangular solve, the 8x8 corner x component terms of the gradient),
//   - triangular solves and Cholesky factorisations advance one pivot per step with the lane-owned
//     partial sums updated in pivot order (identical rounding to the dot-product form),
//   - scalar control (line search state, convergence tests) is computed redundantly by every lane
//     from LDS broadcasts, so control flow stays wave-uniform without any cross-lane traffic.
// Costs (round 6, tools/prof_lbw.sh on the 64 golden objects): 34.3k cycles per iteration - formk 15.4k (Cholesky of WN(1,1) + T 4.7k and of
// the (2,2) block 2.5k, one entry per lane in registers: lbw_potrf_lanes; 4.3k + 4.6k through LDS before; the ten right-hand-side
// solves 2.9k, WN from WN1 1.6k), subsm 7.2k (two 20-step substitutions 2.2k + 2.2k), line search 6.0k, matupd 1.6k, formt 1.5k -
// against 14.7k for the direct form; 135 VGPRs (239 with the unrolled through-LDS factorisations), 67 KB of LDS per 8-object
// workgroup; the bench batch's decode kernel alone 2.62 ms (2.93 before, 1.64 for the direct form); per pipelined bs=32 step at
// ~470 objects + 0.33 ms over the direct form before this round's changes.
// Everything sits in namespace lbw_pub: the type and function names are those of lbfgsb_wave.h.
#pragma once
#include "lbfgsb.h"
#include "lbfgsb_wave.h"       // WSYNC / PTB / PTE / VWS_ / VWY_

#if defined(__HIPCC__)
#pragma clang fp contract(off)

namespace lbw_pub {

struct LbWaveMem {
    double ws[LB_N * LB_M], wy[LB_N * LB_M];
    double sy[LB_M * LB_M], ss[LB_M * LB_M], wt[LB_M * LB_M];
    // WN (upper triangle incl. diagonal: the matrix formk factorises in place) and WN1 (lower triangles of its (1,1) and
    // (2,2) blocks + the full (2,1) block: the running sums formk updates incrementally) share ONE 20 x 20 array; only
    // WN1's diagonal needs a home of its own.  8.4 KB per object instead of 11.4: 16 objects per workgroup.
    double wn[LB_M2 * LB_M2], wn1d[LB_M2];
    double x[LB_N], z[LB_N], r[LB_N], d[LB_N], t[LB_N], g[LB_N], wv[LB_M2];
    double terms[64], fterms[16], bc[2];
    double uv[16];
#ifdef LBW_PROF
    long long prof[24];
#endif
};
#define VSY_(i, j) w->sy[((j)-1) * LB_M + (i)-1]
#define VSS_(i, j) w->ss[((j)-1) * LB_M + (i)-1]
#define VWT_(i, j) w->wt[((j)-1) * LB_M + (i)-1]
#define VWN_(i, j) w->wn[((j)-1) * LB_M2 + (i)-1]
#define VWN1_(i, j) (*((i) == (j) ? &w->wn1d[(i)-1] : &w->wn[((j)-1) * LB_M2 + (i)-1]))      /* i >= j */

struct LbWaveK { double k00, k02, k11, k12; };

// The lane index as a value the optimiser cannot see through: index arithmetic derived from it is then redone where it is
// used (a few VALU instructions) instead of being hoisted out of the iteration loop and kept live across it.  The loop
// carries ~250 VGPRs of state; hoisted lane-derived indices were spilled to scratch and re-loaded eight times per
// iteration (a scratch load is a global-memory round trip in a kernel that is nothing but a latency chain).
__device__ static inline int lbw_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// column j of entry `e` of a packed upper triangle (e = j (j + 1) / 2 + i, 0 <= i <= j): closed form + one correction
// step instead of a search loop (the callers run once per L-BFGS-B iteration in a latency-bound wave)
__device__ static inline int lbw_tri_col(int e) {
    int j = (int)((__fsqrt_rn(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    if ((j + 1) * (j + 2) / 2 <= e) ++j;
    if (j * (j + 1) / 2 > e) --j;
    return j;
}

// value of `v` in lane `src` (src wave-uniform) -> every lane, through v_readlane (no LDS round trip)
__device__ static inline double lbw_bcast(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// A wave-uniform double (every lane computed the same bits from LDS broadcasts) handed to the compiler AS uniform: it may then live
// in an SGPR pair - or, under pressure, in two lanes of a spill VGPR (v_writelane / v_readlane: a few cycles) - instead of two
// VGPRs of every lane.  The loop-carried scalars of lbw_minimize (f, theta, the camera constants) are live across formk / subsm, whose
// unrolled factorisations want every vector register: at the 168-register budget of twelve waves per workgroup they were the values
// the allocator sent to scratch memory (a global-memory round trip inside a latency chain).  Same bits in, same bits out.
__device__ static inline double lbw_uni(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}

// f (returned, identical in every lane) and g (-> w->g) at w->x.
__device__ static inline double lbw_fg(LbWaveMem* w, const LbWaveK& K, int lane) {
    const int c = lane >> 3, i = lane & 7;
    double c0, c1, c2;
    lb_corner(c, &c0, &c1, &c2);
    const double x0 = w->x[0], x1 = w->x[1], x2 = w->x[2], x3 = w->x[3], x4 = w->x[4], x5 = w->x[5], x6 = w->x[6], x7 = w->x[7];
    const double xc = c0 * x2 * x1 + c2 * x4 * x0 + x5;
    const double yc = c1 * x3 + x6;
    const double zc = (-c0) * x2 * x0 + c2 * x4 * x1 + x7;
    const double u = w->uv[2 * c], v = w->uv[2 * c + 1];
    if (i == 0) {
        const double ex = xc * K.k00 / (zc + 1e-4) + K.k02 - u;
        const double ey = yc * K.k11 / (zc + 1e-4) + K.k12 - v;
        w->fterms[2 * c] = ex * ex;
        w->fterms[2 * c + 1] = ey * ey;
    }
    const double dex = (xc * K.k00 / (zc + 1e-6) + K.k02 - u) * 2;
    const double dey = (yc * K.k11 / (zc + 1e-6) + K.k12 - v) * 2;
    double dx, dy, dz;
    switch (i) {
        case 0: dx = c2 * x4; dy = 0; dz = (-c0) * x2; break;
        case 1: dx = c0 * x2; dy = 0; dz = c2 * x4; break;
        case 2: dx = c0 * x1; dy = 0; dz = (-c0) * x0; break;
        case 3: dx = 0; dy = c1; dz = 0; break;
        case 4: dx = c2 * x0; dy = 0; dz = c2 * x1; break;
        case 5: dx = 1; dy = 0; dz = 0; break;
        case 6: dx = 0; dy = 1; dz = 0; break;
        default: dx = 0; dy = 0; dz = 1; break;
    }
    const double den = zc * zc + 1e-6;
    const double gx = K.k00 * (dx * zc - dz * xc) / den;
    const double gy = K.k11 * (dy * zc - dz * yc) / den;
    w->terms[lane] = dex * gx + dey * gy;
    WSYNC();
    if (lane < 8) {
        double s = 0.0;
        for (int k = 0; k < 8; ++k) s += w->terms[k * 8 + lane];
        w->g[lane] = s;
    }
    double f = 0.0;
    for (int k = 0; k < 16; ++k) f += w->fterms[k];
    WSYNC();
    return f;
}

__device__ static inline double lbw_dot8(const double* a, const double* b) {
    double s = 0.0;
    for (int i = 0; i < LB_N; ++i) s += a[i] * b[i];
    return s;
}

// Cholesky A = U'U (upper, column-major, leading dimension lda) of an n x n block (n <= LB_M), in place; the diagonal of the result
// holds 1 / u_jj (lb_potrf) - of ONE block (b == nullptr) or of TWO independent blocks at once (WN's (1,1) block together with T:
// formt's factorisation of T is only ever consulted for its positive-definiteness verdict - without bounds nothing calls bmv - and it
// is the last thing an iteration does before the next formk, so it rides along there).  Returns 0, or 1 if `a` failed, or 2 if `b`
// failed (the caller resets the limited-memory matrices in both cases, exactly as lb_minimize does after either failure).
//
// Round 6: ONE ENTRY PER LANE, in registers.  Lane e < 55 owns entry (row r, column c), r <= c, e = c (c + 1) / 2 + r, of each block,
// and the running dot product sum_{k < r} u_kr u_kc that the dot-product form subtracts from it - accumulated term by term in k order
// from 0.0, one term per pivot, so every value is rounded exactly as in lb_potrf (the scalar form) and in the round-2 form of this
// function (one pivot per step through LDS: two LDS round trips, a dot product of length J and two wave fences per pivot, ~440
// cycles).  Per pivot k here: the diagonal entry's a_kk - dot goes to every lane by v_readlane and every lane takes the reciprocal
// square root of it (the same bits everywhere: no second broadcast); the lanes of row k scale their entries; the lanes below gather
// u_kr and u_kc from the two row-k lanes above them with ds_bpermute (no memory behind it) and add one product to their dot.
// The blocks are read from LDS once and written back once.
template <bool TWO>
__device__ static inline int lbw_potrf_lanes(double* a, int lda, double* b, int ldb, int n, int lane) {
    const int e = lane < (LB_M * (LB_M + 1)) / 2 ? lane : (LB_M * (LB_M + 1)) / 2 - 1;
    const int c = lbw_tri_col(e), r = e - c * (c + 1) / 2;
    const bool act = lane < (LB_M * (LB_M + 1)) / 2 && c < n;
    constexpr bool two = TWO;              // (a template parameter: as a run-time flag it put uniform branches between the two chains)
    double va = act ? a[c * lda + r] : 0.0, vb = (act && two) ? b[c * ldb + r] : 0.0;
    double da = 0.0, db = 0.0;
    const int src_r = (r * (r + 1) / 2) << 2, src_c = (c * (c + 1) / 2) << 2;        // byte index of lane (0, r) / (0, c) for ds_bpermute
    auto gather = [&](double v, int byte_idx) -> double {
        const int lo = __builtin_amdgcn_ds_bpermute(byte_idx, __double2loint(v));
        const int hi = __builtin_amdgcn_ds_bpermute(byte_idx, __double2hiint(v));
        return __hiloint2double(hi, lo);
    };
    // A pivot that is not positive ends the factorisation in the scalar form.  Here the verdict is collected and returned behind the
    // loop (what the entries hold after a failed pivot is never read: every caller resets the matrices on a non-zero return): a
    // return inside the loop would put a branch between the two blocks' reciprocal-square-root chains and run them one after the
    // other (measured: 600 cycles per pivot for two blocks against 350 for one).
    int fail_a = LB_M, fail_b = LB_M;                                    // first failing pivot of each block
#pragma unroll
    for (int k = 0; k < LB_M; ++k) {
        if (k < n) {
            const int dk = k * (k + 1) / 2 + k;                          // lane of the diagonal entry (k, k)
            const double pa = va - da, pb = vb - db;                     // (row-k lanes: a_kc - dot; the diagonal lane: a_kk - s)
            const double ajj_a = lbw_bcast(pa, dk), ajj_b = two ? lbw_bcast(pb, dk) : 1.0;
            if (!(ajj_a > 0.0) && fail_a == LB_M) fail_a = k;            // uniform
            if (!(ajj_b > 0.0) && fail_b == LB_M) fail_b = k;
            const double ra = lb_rsqrt(ajj_a), rb = two ? lb_rsqrt(ajj_b) : 0.0;       // the diagonal keeps 1 / u_kk (lb_potrf)
            const double ua = pa * ra, ub = pb * rb;
            if (act && r == k) { va = c == k ? ra : ua; vb = c == k ? rb : ub; }
            // lanes below row k: one more term of their dot products, u_kr * u_kc (row k's values sit in lanes src + k)
            const double ukr_a = gather(ua, src_r + (k << 2)), ukc_a = gather(ua, src_c + (k << 2));
            double ukr_b = 0.0, ukc_b = 0.0;
            if (two) { ukr_b = gather(ub, src_r + (k << 2)); ukc_b = gather(ub, src_c + (k << 2)); }
            if (act && r > k) { da += ukr_a * ukc_a; if (two) db += ukr_b * ukc_b; }
        }
    }
    if (fail_a != LB_M || fail_b != LB_M) return fail_a <= fail_b ? 1 : 2;      // (at one pivot the first block's verdict comes first)
    if (act) { a[c * lda + r] = va; if (two) b[c * ldb + r] = vb; }
    WSYNC();
    return 0;
}

// U' x = b, single right-hand side in LDS (n <= LB_M2): lane j owns b[j], its partial sum and COLUMN j of U in registers
// (all operands of the solve are fetched by one batch of LDS reads: a substitution step that waits for its own LDS
// read costs ~330 cycles, one that only passes the pivot value by v_readlane ~90); lane k also holds the k-th
// (reciprocal) diagonal entry.  The diagonal is never zero after a successful lb_potrf; a zero anywhere is reported before
// the first step (the callers only test for != 0).
__device__ static inline int lbw_trsv_ut(const double* a, int lda, int n, double* b, double* bc, int lane) {
    (void)bc;
    const bool act = lane < n;
    const int col = (act ? lane : 0) * lda;
    double cv[LB_M2];
#pragma unroll
    for (int k = 0; k < LB_M2; ++k) cv[k] = a[col + (k < n ? k : 0)];
    const double dg = a[col + (act ? lane : 0)];
    double bj = act ? b[lane] : 0.0;
    if (__any(act && dg == 0.0)) return 1;
    double dot = 0.0;
#pragma unroll
    for (int k = 0; k < LB_M2; ++k) {
        if (k < n) {
            // every lane forms (b - dot) / u_ll with its OWN diagonal entry; lane k's is x_k (its dot is complete at step k)
            const double mine = (bj - dot) * dg;
            const double bk = lbw_bcast(mine, k);
            if (lane == k) bj = mine;
            if (lane > k && act) dot += cv[k] * bk;
        }
    }
    if (act) b[lane] = bj;
    WSYNC();
    return 0;
}
// U x = b (column-oriented back substitution, same update order as the scalar version): lane i owns b[i] and ROW i of U.
// The row is fetched in two halves - entries 10 .. 19 before the chain starts, entries 0 .. 9 once the first five steps have
// retired theirs (their LDS latency then runs under steps 14 .. 10) - so that at most 15 + 10 entries are live at once: with all
// twenty up front (40 registers beside the iteration's state) the allocator sent four of them to scratch memory at the
// 168-register budget of twelve waves per workgroup, and reloaded each inside the dependent chain.
__device__ static inline int lbw_trsv_un(const double* a, int lda, int n, double* b, double* bc, int lane) {
    (void)bc;
    const bool act = lane < n;
    const int row = act ? lane : 0;
    constexpr int H = LB_M2 / 2;
    double rh[H], rl[H];
#pragma unroll
    for (int j = 0; j < H; ++j) rh[j] = a[((H + j) < n ? (H + j) : 0) * lda + row];
    const double dg = a[row * lda + row];
    double bk = act ? b[lane] : 0.0;
    if (__any(act && dg == 0.0)) return 1;
#pragma unroll
    for (int j = LB_M2 - 1; j >= 0; --j) {
        if (j == LB_M2 - 6) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < H; ++q) rl[q] = a[(q < n ? q : 0) * lda + row];
            asm volatile("" ::: "memory");
        }
        if (j < n) {
            const double mine = bk * dg;                     // own (reciprocal) diagonal entry: lane j's is x_j
            const double tmp = -lbw_bcast(mine, j);
            if (lane == j) bk = mine;
            if (lane < j) bk += tmp * (j >= H ? rh[j - H] : rl[j]);
        }
    }
    if (act) b[lane] = bk;
    WSYNC();
    return 0;
}

// forward substitution U' x = b for ONE right-hand side owned by this lane (n <= LB_M), unrolled.  Column-oriented: as
// soon as x_J exists its term goes into the running sums of all later rows (independent multiply-adds the SIMD pipelines),
// so the dependent chain is 4 operations per step instead of 2 J + 2; each running sum still receives its terms in the
// order k = 0, 1, ... from 0.0, i.e. the values are those of the row-oriented lb_trsv_ut.
// (Round 6, same box: all 55 entries of U loaded up front - 55 broadcast reads in one batch instead of a row per step - 2.93k -> 2.27k
// cycles for the solves, but 217 instead of 135 VGPRs for the kernel and 12.49 against 12.44 ms per pipelined step: at 135 registers
// two stem waves fit on a SIMD beside the decode's two, at 217 none.  Not kept: the register count of this kernel is part of its cost.)
// FULL: n == LB_M (every iteration once the memory is full: 16 of ~26) compiles without the per-index guards - the guarded
// form keeps ~40 uniform predicates alive and spills SGPRs into VGPR lanes around every step.
template <bool FULL>
__device__ static inline int lbw_rhs_solve_t(const double* u, double* b, int n) {
    double bv[LB_M], acc[LB_M];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < LB_M; ++j) {
        bv[j] = b[(FULL || j < n) ? j : 0];
        acc[j] = 0.0;
        if ((FULL || j < n) && u[j * LB_M2 + j] == 0.0) bad = 1;
    }
    if (bad) return 1;
#pragma unroll
    for (int J = 0; J < LB_M; ++J) {
        if (FULL || J < n) {
            // this step's row of U only (a compiler that hoists all 45 entries runs out of registers)
            asm volatile("" ::: "memory");
            double ur[LB_M];
#pragma unroll
            for (int Jp = J; Jp < LB_M; ++Jp) ur[Jp] = u[((FULL || Jp < n) ? Jp : J) * LB_M2 + J];
            const double xJ = (bv[J] - acc[J]) * ur[J];                    // (reciprocal diagonal)
            bv[J] = xJ;
#pragma unroll
            for (int Jp = J + 1; Jp < LB_M; ++Jp)
                if (FULL || Jp < n) acc[Jp] += ur[Jp] * xJ;
        }
    }
#pragma unroll
    for (int j = 0; j < LB_M; ++j) if (FULL || j < n) b[j] = bv[j];
    return 0;
}
__device__ static inline int lbw_rhs_solve(const double* u, double* b, int n) {
    return n == LB_M ? lbw_rhs_solve_t<true>(u, b, n) : lbw_rhs_solve_t<false>(u, b, n);
}

__device__ static inline int lbw_formk(LbWaveMem* w, int iupdat, double theta, int col, int head, int lane) {
    lane = lbw_opaque(lane);
    const int m = LB_M, n = LB_N;
    PTB(f0_);
    if (iupdat > m) {
        // shift the old part of WN1 one step up-left: three 9x9 index grids (lower triangles of the (1,1)
        // and (2,2) blocks, the full (2,1) block); two-phase (read all, then write all); slots are
        // statically indexed so they stay in registers (a scan with a running slot counter went to scratch
        // and cost 14k cycles per iteration)
        double v[6]; int dr[6], dc[6]; bool ok[6];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = lane + 64 * r;
            const bool in = e < (m - 1) * (m - 1);
            const int ee = in ? e : 0;
            const int c = ee / (m - 1) + 1, q = ee % (m - 1) + 1;     // column 1..m-1, row index 1..m-1
            // (q >= c keeps the first two on or below the diagonal; rows of the third are >= m + 1 > c)
            const int qs = q >= c ? q : c;
            ok[3 * r + 0] = in && q >= c; v[3 * r + 0] = VWN1_(qs + 1, c + 1);         dr[3 * r + 0] = qs;     dc[3 * r + 0] = c;
            ok[3 * r + 1] = in && q >= c; v[3 * r + 1] = VWN1_(m + qs + 1, m + c + 1); dr[3 * r + 1] = m + qs; dc[3 * r + 1] = m + c;
            ok[3 * r + 2] = in;           v[3 * r + 2] = VWN1_(m + q + 1, c + 1);      dr[3 * r + 2] = m + q;  dc[3 * r + 2] = c;
        }
        WSYNC();
#pragma unroll
        for (int q = 0; q < 6; ++q) if (ok[q]) VWN1_(dr[q], dc[q]) = v[q];
        WSYNC();
    }
    PTE(f0_, 8); PTB(f1_);
    {
        int ipntr = head + col - 1;
        if (ipntr > m) ipntr -= m;
        const int jy = lane + 1;
        if (jy <= col) {
            const int jpntr = (head + jy - 2) % m + 1;
            double temp1 = 0.0;
            for (int k = 1; k <= n; ++k) temp1 += VWY_(k, ipntr) * VWY_(k, jpntr);
            VWN1_(col, jy) = temp1;
            VWN1_(m + col, m + jy) = 0.0;
            VWN1_(m + col, jy) = 0.0;
        }
        WSYNC();
        const int i = lane + 1;
        if (i <= col) {
            const int ip = (head + i - 2) % m + 1;
            double temp3 = 0.0;
            for (int k = 1; k <= n; ++k) temp3 += VWS_(k, ip) * VWY_(k, ipntr);
            VWN1_(m + i, col) = temp3;
        }
        WSYNC();
    }
    PTE(f1_, 9); PTB(f2_);
    const float rcol = 1.0f / (float)col;
    for (int e = lane; e < col * col; e += 64) {
        const int q = (int)(((float)e + 0.5f) * rcol);       // e / col for e < 100 (exact: margins of 0.5 / col)
        const int iy = q + 1, jy = e - q * col + 1;
        const int is = col + iy, is1 = m + iy, js = col + jy, js1 = m + jy;
        if (jy <= iy) {
            double v = VWN1_(iy, jy) / theta;
            if (jy == iy) v = v + VSY_(iy, iy);
            VWN_(jy, iy) = v;
            VWN_(js, is) = VWN1_(is1, js1) * theta;
        }
        VWN_(jy, is) = (jy < iy) ? -VWN1_(is1, jy) : VWN1_(is1, jy);
    }
    WSYNC();
    PTE(f2_, 10); PTB(f3_);
    {   // WN (1,1) block, and T left unfactorised by lbw_formt at the end of the previous iteration
        const int r2 = lbw_potrf_lanes<true>(w->wn, LB_M2, w->wt, LB_M, col, lane);
        if (r2 != 0) return r2 == 2 ? -3 : -1;
    }
    PTE(f3_, 11); PTB(f4_);
    const int col2 = 2 * col;
    {   // L^-1 (-L_a' + R_z'): one right-hand side (column) per lane
        int bad = 0;
        if (lane < col) bad = lbw_rhs_solve(w->wn, &VWN_(1, col + 1 + lane), col);
        if (__any(bad)) return -1;
        WSYNC();
    }
    PTE(f4_, 12); PTB(f5_);
    {   // (2,2) block += (L^-1 ...)'(L^-1 ...), upper triangle: one entry per lane (<= 55)
        const int j = lbw_tri_col(lane);
        const int i = lane - j * (j + 1) / 2;
        if (j < col) {
            const int is = col + 1 + i, js = col + 1 + j;
            double pa[LB_M], pb[LB_M];
#pragma unroll
            for (int k = 1; k <= LB_M; ++k) { const int kk = k <= col ? k : 1; pa[k - 1] = VWN_(kk, is); pb[k - 1] = VWN_(kk, js); }
            double dot = 0.0;
#pragma unroll
            for (int k = 1; k <= LB_M; ++k) if (k <= col) dot += pa[k - 1] * pb[k - 1];
            VWN_(is, js) = VWN_(is, js) + dot;
        }
        WSYNC();
    }
    PTE(f5_, 13); PTB(f6_);
    if (lbw_potrf_lanes<false>(&VWN_(col + 1, col + 1), LB_M2, nullptr, 0, col, lane) != 0) return -2;
    PTE(f6_, 14);
    (void)col2;
    return 0;
}

__device__ static inline int lbw_subsm(LbWaveMem* w, double theta, int col, int head, int lane) {
    lane = lbw_opaque(lane);
    const int m = LB_M, n = LB_N;
    const int col2 = 2 * col;
    if (lane < col2) {
        const int i = (lane < col ? lane : lane - col) + 1;
        const int pointr = (head + i - 2) % m + 1;
        double tmp = 0.0;
        if (lane < col) { for (int j = 1; j <= n; ++j) tmp += VWY_(j, pointr) * w->r[j - 1]; w->wv[lane] = tmp; }
        else { for (int j = 1; j <= n; ++j) tmp += VWS_(j, pointr) * w->r[j - 1]; w->wv[lane] = theta * tmp; }
    }
    WSYNC();
    PTB(s0_);
    if (lbw_trsv_ut(w->wn, LB_M2, col2, w->wv, w->bc, lane) != 0) return 1;
    PTE(s0_, 16);
    if (lane < col) w->wv[lane] = -w->wv[lane];
    WSYNC();
    PTB(s1_);
    if (lbw_trsv_un(w->wn, LB_M2, col2, w->wv, w->bc, lane) != 0) return 1;
    PTE(s1_, 17);
    if (lane < n) {
        double di = w->r[lane];
        int pointr = head;
        const double rt = 1.0 / theta;
        // two batches of LDS reads (five terms each: 40 registers), each followed by its part of the sum in index order
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double ay[LB_M / 2], as[LB_M / 2], vy[LB_M / 2], vs[LB_M / 2];
#pragma unroll
            for (int q = 0; q < LB_M / 2; ++q) {
                const int jy = h * (LB_M / 2) + q + 1;
                const int jj = jy <= col ? jy : 1;
                const int pp = (head + jj - 2) % m + 1;
                ay[q] = VWY_(lane + 1, pp); as[q] = VWS_(lane + 1, pp);
                vy[q] = w->wv[jj - 1]; vs[q] = w->wv[col + jj - 1];
            }
#pragma unroll
            for (int q = 0; q < LB_M / 2; ++q)
                if (h * (LB_M / 2) + q + 1 <= col) di = di + ay[q] * vy[q] * rt + as[q] * vs[q];
        }
        (void)pointr;
        di = rt * di;
        w->r[lane] = di;
        w->z[lane] = w->z[lane] + di;
    }
    WSYNC();
    return 0;
}

__device__ static inline void lbw_matupd(LbWaveMem* w, int* itail, int iupdat, int* col, int* head, double* theta,
                                         double rr, double dr, double stp, double dtd, int lane) {
    lane = lbw_opaque(lane);
    const int m = LB_M, n = LB_N;
    {   // (value selection, not stores through col / head per branch: see lb_dcstep)
        const bool grow = iupdat <= m;
        const int col0 = *col, head0 = *head, itail0 = *itail;
        *col = grow ? iupdat : col0;
        *itail = grow ? (head0 + iupdat - 2) % m + 1 : itail0 % m + 1;
        *head = grow ? head0 : head0 % m + 1;
    }
    if (lane < n) { VWS_(lane + 1, *itail) = w->d[lane]; VWY_(lane + 1, *itail) = w->r[lane]; }
    *theta = rr / dr;
    if (iupdat > m) {
        // move the old information: SS upper triangle and SY lower triangle one step up-left (9x9 grids)
        double vs[2], vy[2]; int dst[2]; bool oks[2], oky[2];
        const int cm = *col - 1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = lane + 64 * r;
            const bool in0 = e < (m - 1) * (m - 1);
            const int ee = in0 ? e : 0;
            const int j = ee / (m - 1) + 1, q = ee % (m - 1) + 1;     // column j, row q in 1..m-1
            const bool in = in0 && j <= cm && q <= cm;
            oks[r] = in && q <= j; oky[r] = in && q >= j;
            vs[r] = VSS_(q + 1, j + 1); vy[r] = VSY_(q + 1, j + 1);
            dst[r] = (j - 1) * LB_M + (q - 1);
        }
        WSYNC();
#pragma unroll
        for (int r = 0; r < 2; ++r) { if (oks[r]) w->ss[dst[r]] = vs[r]; if (oky[r]) w->sy[dst[r]] = vy[r]; }
    }
    WSYNC();
    const int j = lane + 1;
    if (j <= *col - 1) {
        const int pointr = (*head + j - 2) % m + 1;
        VSY_(*col, j) = lbw_dot8(w->d, &VWY_(1, pointr));
        VSS_(j, *col) = lbw_dot8(&VWS_(1, pointr), w->d);
    }
    if (lane == 0) {
        VSS_(*col, *col) = (stp == 1.0) ? dtd : stp * stp * dtd;
        VSY_(*col, *col) = dr;
    }
    WSYNC();
}

__device__ static inline int lbw_formt(LbWaveMem* w, int col, double theta, int lane) {
    lane = lbw_opaque(lane);
    if (lane < col) w->wv[lane] = 1.0 / VSY_(lane + 1, lane + 1);      // 1 / SY(k, k) (wv is free between subsm calls)
    WSYNC();
    const int j0 = lbw_tri_col(lane);
    const int i = lane - j0 * (j0 + 1) / 2 + 1, j = j0 + 1;      // 1 <= i <= j
    if (j <= col) {
        if (i == 1) VWT_(1, j) = theta * VSS_(1, j);
        else {
            double ddum = 0.0;
#pragma unroll
            for (int h = 0; h < 1; ++h) {                  // one batch of reads, then the sum in index order
                double pa[9], pb[9], pr[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const int k = h * 9 + q + 1;
                    const int kk = k <= i - 1 ? k : 1;
                    pa[q] = VSY_(i, kk); pb[q] = VSY_(j, kk); pr[q] = w->wv[kk - 1];
                }
#pragma unroll
                for (int q = 0; q < 9; ++q) if (h * 9 + q + 1 <= i - 1) ddum = ddum + pa[q] * pb[q] * pr[q];
            }
            VWT_(i, j) = ddum + theta * VSS_(i, j);
        }
    }
    WSYNC();
    return 0;            // T's factorisation (a pass / fail verdict only) runs inside the next lbw_formk: lbw_potrf_lanes<true>
}

// Driver: identical control flow to lb_minimize (lbfgsb.h).  w->x holds x0 on entry, the result on exit.
__device__ static inline int lbw_minimize(LbWaveMem* w, const LbWaveK& K, double* f_out, int* nit_out, int lane,
                                          int maxiter, int maxfun) {
    const int n = LB_N, maxls = 20;
    const double epsmch = 2.220446049250313e-16, factr = 1e7, pgtol = 1e-5;
    const double ftol = 1e-3, gtol = 0.9, xtol = 0.1, big = 1e10;
    const double tol = factr * epsmch;
    int col = 0, head = 1, itail = 0, iupdat = 0, updatd = 0, iter = 0, nfgv = 0, info;
    double theta = 1.0, f, fold = 0.0, gd = 0.0, gdold = 0.0, stp = 0.0, dnorm = 0.0, dtd = 0.0, sbgnrm;
    LbSearch S;

#ifdef LBW_PROF
    if (lane < 24) w->prof[lane] = 0;
    WSYNC();
    const long long tstart_ = __builtin_readcyclecounter();
#endif
    f = lbw_uni(lbw_fg(w, K, lane)); nfgv = 1;
    {   // non-finite key points: x0, fun = NaN / Inf, 0 iterations, own status (see lb_minimize)
        bool finite = lb_isfinite(f);
        for (int i = 0; i < n; ++i) finite = finite && lb_isfinite(w->g[i]);
        if (!finite) { *f_out = f; *nit_out = 0; return LB_STATUS_NONFINITE; }
    }
    sbgnrm = 0.0;
    for (int i = 0; i < n; ++i) sbgnrm = fmax(sbgnrm, fabs(w->g[i]));
    if (sbgnrm <= pgtol) { *f_out = f; *nit_out = 0; return 0; }

    for (;;) {
        if (col == 0) {
            if (lane < n) w->z[lane] = w->x[lane] + 1.0 * (-w->g[lane]);
            WSYNC();
        } else {
            if (lane < n) { w->z[lane] = w->x[lane]; w->r[lane] = -w->g[lane]; }
            WSYNC();
            info = 0;
            PTB(tk_);
            if (updatd) info = lbw_formk(w, iupdat, theta, col, head, lane);
            PTE(tk_, 0); PTB(ts_);
            if (info == 0) info = lbw_subsm(w, theta, col, head, lane);
            PTE(ts_, 1);
            if (info != 0) {
                col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0;
                WSYNC();
                continue;
            }
        }
        if (lane < n) { w->d[lane] = w->z[lane] - w->x[lane]; w->t[lane] = w->x[lane]; w->r[lane] = w->g[lane]; }
        WSYNC();
        dtd = lbw_dot8(w->d, w->d);
        dnorm = sqrt(dtd);
        const double stpmx = big;
        stp = (iter == 0) ? fmin(1.0 / dnorm, stpmx) : 1.0;
        fold = f;
        int ifun = 0, iback = 0, ls_fail = 0, start = 1;
        info = 0;
        PTB(tl_);
        for (;;) {
            gd = lbw_dot8(w->g, w->d);
            if (ifun == 0) {
                gdold = gd;
                if (gd >= 0.0) { info = -4; break; }
            }
            const int task = lb_dcsrch(f, gd, &stp, ftol, gtol, xtol, 0.0, stpmx, start, &S);
            start = 0;
            if (task == LS_ERROR) { info = -4; break; }
            if (task == LS_CONV || task == LS_WARN) break;
            ifun += 1; nfgv += 1; iback = ifun - 1;
            WSYNC();
            if (lane < n) w->x[lane] = (stp == 1.0) ? w->z[lane] : stp * w->d[lane] + w->t[lane];
            WSYNC();
            if (iback >= maxls) { ls_fail = 1; break; }
            f = lbw_uni(lbw_fg(w, K, lane));
        }
        PTE(tl_, 2);
        if (info != 0 || ls_fail) {
            WSYNC();
            if (lane < n) { w->x[lane] = w->t[lane]; w->g[lane] = w->r[lane]; }
            WSYNC();
            f = fold;
            if (col == 0) { *f_out = f; *nit_out = iter; return 2; }
            col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0;
            continue;
        }
        iter += 1;
        sbgnrm = 0.0;
        for (int i = 0; i < n; ++i) sbgnrm = fmax(sbgnrm, fabs(w->g[i]));
        if (iter >= maxiter || nfgv > maxfun) { *f_out = f; *nit_out = iter; return 1; }
        if (sbgnrm <= pgtol) break;
        const double ddum0 = lb_max3(fabs(fold), fabs(f), 1.0);
        if ((fold - f) <= tol * ddum0) break;
        WSYNC();
        if (lane < n) w->r[lane] = w->g[lane] - w->r[lane];
        WSYNC();
        const double rr = lbw_dot8(w->r, w->r);
        double dr, ddum;
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else {
            dr = (gd - gdold) * stp;
            WSYNC();
            if (lane < n) w->d[lane] = stp * w->d[lane];
            WSYNC();
            ddum = -gdold * stp;
        }
        if (dr <= epsmch * ddum) { updatd = 0; continue; }
        updatd = 1; iupdat += 1;
        PTB(tm_);
        lbw_matupd(w, &itail, iupdat, &col, &head, &theta, rr, dr, stp, dtd, lane);
        theta = lbw_uni(theta);
        PTE(tm_, 3); PTB(tt_);
        if (lbw_formt(w, col, theta, lane) != 0) { col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0; }
        PTE(tt_, 4);
    }
    *f_out = f; *nit_out = iter;
#ifdef LBW_PROF
    WSYNC();
    if (lane == 0) w->prof[7] = __builtin_readcyclecounter() - tstart_;
    WSYNC();
    if (lane < 8) w->x[lane] = (double)w->prof[(LBW_PROF - 1) * 8 + lane];      // LBW_PROF = 1, 2, 3: which eight counters
    WSYNC();
#endif
    return 0;
}

}  // namespace lbw_pub
#endif  // __HIPCC__
