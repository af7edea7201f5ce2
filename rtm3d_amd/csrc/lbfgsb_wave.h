// Wave-cooperative form of lbfgsb.h in its DIRECT form (lb_minimize with direct = 1; the product's default in rounds 3-5, opt-in
// since round 6: lbfgsb_wave_pub.h is the default): ONE 64-lane wavefront solves ONE object.
//
// Same algorithm and the same fp64 arithmetic in the same order as the scalar version - every individual sum runs
// sequentially in index order - so the two are bit-identical (tests: rtm3d_decode3d vs rtm3d_decode3d_scalar):
//   - the 64 (corner, unknown) terms of the gradient and the 16 terms of the objective are spread over the lanes;
//   - the search direction comes from the two-loop recursion over the stored pairs (lb_two_loop): every lane runs the whole
//     recursion redundantly on the same LDS operands (broadcast reads), so its 2 * col dependent steps need no cross-lane
//     traffic at all; only lanes 0..7 store the result;
//   - scalar control (line search state, convergence tests) is computed redundantly by every lane, so control flow stays
//     wave-uniform.
// History: up to round 2 this file carried the published subspace step (formk / subsm / formt: 20 x 20 LEL' factorisation
// spread over the lanes, 8.4 KB of LDS and 178 VGPRs per object, 35.7k cycles per iteration of which 24k in those three).
// Without bounds that step is  -B^-1 g, which the two-loop recursion delivers from the same pairs in ~3k cycles; measured
// against the reference's SciPy results the direct form is as close as the published one (lbfgsb.h, header).  The
// published form lives on in lbfgsb.h (direct = 0) as the cross-check.
// A wave's LDS instructions execute in order, so WSYNC() is only a compiler/LDS ordering fence at
// wavefront scope (no s_barrier): a workgroup may hold several independent objects, one per wave.
#pragma once
#include "lbfgsb.h"

#if defined(__HIPCC__)
#pragma clang fp contract(off)

#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// -DLBW_PROF=1 (diagnostic build only, tools/prof_lbw.sh): per-phase cycle sums (s_memtime) in LDS; at the end the first
// eight replace the solution in w->x, where tools/prof_lbw.py reads them
#ifdef LBW_PROF
#define PTB(name) const long long name = __builtin_readcyclecounter()
#define PTE(name, k) do { if (lane == 0) w->prof[k] += __builtin_readcyclecounter() - name; } while (0)
#else
#define PTB(name)
#define PTE(name, k)
#endif

struct LbWaveMem {
    double ws[LB_N * LB_M], wy[LB_N * LB_M];      // S and Y, one pair per ring slot (column)
    double rho[LB_M];                             // 1 / (s'y) per ring slot
    double x[LB_N], z[LB_N], r[LB_N], d[LB_N], t[LB_N], g[LB_N];
    double terms[64], fterms[16];
    double uv[16];
#ifdef LBW_PROF
    long long prof[8];
#endif
};
#define VWS_(i, j) w->ws[((j)-1) * LB_N + (i)-1]
#define VWY_(i, j) w->wy[((j)-1) * LB_N + (i)-1]

struct LbWaveK { double k00, k02, k11, k12; };

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

// z = x - H g by the two-loop recursion (lb_two_loop, same operations in the same order), run redundantly by every lane:
// q and alpha live in registers, the pairs are read from LDS at wave-uniform addresses.  The recursion is ONE dependent
// chain of 2 * col steps (eight products, three tree additions, scale, multiply, subtract), ~360 cycles each: 7.2k cycles
// per iteration.  (Measured and not kept: fetching the operands of step k + 1 during step k from two alternating register
// buffers - 7.5k cycles: the chain waits for arithmetic, not for LDS; with the next operands copied into place 9.0k.)
__device__ static inline void lbw_two_loop(LbWaveMem* w, double theta, int col, int head, int lane) {
    const int m = LB_M, n = LB_N;
    double q[LB_N], alpha[LB_M];
#pragma unroll
    for (int i = 0; i < n; ++i) q[i] = -w->g[i];
    int p = head + col - 1;
    if (p > m) p -= m;
#pragma unroll
    for (int jj = 0; jj < LB_M; ++jj) {          // newest pair first (j = col - jj)
        if (jj < col) {
            double sv[LB_N], yv[LB_N];
#pragma unroll
            for (int i = 0; i < n; ++i) { sv[i] = VWS_(i + 1, p); yv[i] = VWY_(i + 1, p); }
            const double a = lb_dot8t(sv, q) * w->rho[p - 1];
            alpha[jj] = a;
#pragma unroll
            for (int i = 0; i < n; ++i) q[i] = q[i] - a * yv[i];
            p = p - 1;
            if (p < 1) p += m;
        }
    }
    const double rt = 1.0 / theta;
#pragma unroll
    for (int i = 0; i < n; ++i) q[i] = rt * q[i];
    p = head;
#pragma unroll
    for (int jj = LB_M - 1; jj >= 0; --jj) {     // oldest pair first (j = col - jj)
        if (jj < col) {
            double sv[LB_N], yv[LB_N];
#pragma unroll
            for (int i = 0; i < n; ++i) { sv[i] = VWS_(i + 1, p); yv[i] = VWY_(i + 1, p); }
            const double c = alpha[jj] - lb_dot8t(yv, q) * w->rho[p - 1];
#pragma unroll
            for (int i = 0; i < n; ++i) q[i] = q[i] + c * sv[i];
            p = p % m + 1;
        }
    }
    WSYNC();
    if (lane < n) {
        double qi = q[0];
#pragma unroll
        for (int i = 1; i < n; ++i) qi = lane == i ? q[i] : qi;       // value selection, not a dynamically indexed register array
        w->z[lane] = w->x[lane] + qi;
    }
    WSYNC();
}

// lb_matupd for the direct form: ring pointers, the new pair, rho of its slot, theta (S'S, S'Y are not needed)
__device__ static inline void lbw_matupd(LbWaveMem* w, int* itail, int iupdat, int* col, int* head, double* theta,
                                         double rr, double dr, int lane) {
    const int m = LB_M, n = LB_N;
    {   // (value selection, not stores through col / head per branch: see lb_dcstep)
        const bool grow = iupdat <= m;
        const int col0 = *col, head0 = *head, itail0 = *itail;
        *col = grow ? iupdat : col0;
        *itail = grow ? (head0 + iupdat - 2) % m + 1 : itail0 % m + 1;
        *head = grow ? head0 : head0 % m + 1;
    }
    if (lane < n) { VWS_(lane + 1, *itail) = w->d[lane]; VWY_(lane + 1, *itail) = w->r[lane]; }
    if (lane == 0) w->rho[*itail - 1] = 1.0 / dr;
    *theta = rr / dr;
    WSYNC();
}

// Driver: identical control flow to lb_minimize(direct = 1) (lbfgsb.h).  w->x holds x0 on entry, the result on exit.
__device__ static inline int lbw_minimize(LbWaveMem* w, const LbWaveK& K, double* f_out, int* nit_out, int lane,
                                          int maxiter, int maxfun) {
    const int n = LB_N, maxls = 20;
    const double epsmch = 2.220446049250313e-16, factr = 1e7, pgtol = 1e-5;
    const double ftol = 1e-3, gtol = 0.9, xtol = 0.1, big = 1e10;
    const double tol = factr * epsmch;
    int col = 0, head = 1, itail = 0, iupdat = 0, iter = 0, nfgv = 0, info;
    double theta = 1.0, f, fold = 0.0, gd = 0.0, gdold = 0.0, stp = 0.0, dnorm = 0.0, dtd = 0.0, sbgnrm;
    LbSearch S;

#ifdef LBW_PROF
    if (lane < 8) w->prof[lane] = 0;
    WSYNC();
    const long long tstart_ = __builtin_readcyclecounter();
#endif
    f = lbw_fg(w, K, lane); nfgv = 1;
    {   // non-finite key points: x0, fun = NaN / Inf, 0 iterations, own status (see lb_minimize)
        bool finite = lb_isfinite(f);
        for (int i = 0; i < n; ++i) finite = finite && lb_isfinite(w->g[i]);
        if (!finite) { *f_out = f; *nit_out = 0; return LB_STATUS_NONFINITE; }
    }
    sbgnrm = 0.0;
    for (int i = 0; i < n; ++i) sbgnrm = fmax(sbgnrm, fabs(w->g[i]));
    if (sbgnrm <= pgtol) { *f_out = f; *nit_out = 0; return 0; }

    for (;;) {
        PTB(td_);
        if (col == 0) {
            if (lane < n) w->z[lane] = w->x[lane] + 1.0 * (-w->g[lane]);
            WSYNC();
        } else {
            lbw_two_loop(w, theta, col, head, lane);
        }
        PTE(td_, 0);
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
            PTB(tc_);
            const int task = lb_dcsrch(f, gd, &stp, ftol, gtol, xtol, 0.0, stpmx, start, &S);
            PTE(tc_, 5);
            start = 0;
            if (task == LS_ERROR) { info = -4; break; }
            if (task == LS_CONV || task == LS_WARN) break;
            ifun += 1; nfgv += 1; iback = ifun - 1;
            WSYNC();
            if (lane < n) w->x[lane] = (stp == 1.0) ? w->z[lane] : stp * w->d[lane] + w->t[lane];
            WSYNC();
            if (iback >= maxls) { ls_fail = 1; break; }
            PTB(tf_);
            f = lbw_fg(w, K, lane);
            PTE(tf_, 4);
        }
        PTE(tl_, 2);
        if (info != 0 || ls_fail) {
            WSYNC();
            if (lane < n) { w->x[lane] = w->t[lane]; w->g[lane] = w->r[lane]; }
            WSYNC();
            f = fold;
            if (col == 0) { *f_out = f; *nit_out = iter; return 2; }
            col = 0; head = 1; theta = 1.0; iupdat = 0;
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
        if (dr <= epsmch * ddum) continue;
        iupdat += 1;
        PTB(tm_);
        lbw_matupd(w, &itail, iupdat, &col, &head, &theta, rr, dr, lane);
        PTE(tm_, 3);
    }
    *f_out = f; *nit_out = iter;
#ifdef LBW_PROF
    WSYNC();
    if (lane == 0) w->prof[7] = __builtin_readcyclecounter() - tstart_;
    WSYNC();
    if (lane < 8) w->x[lane] = (double)w->prof[lane];
    WSYNC();
#endif
    return 0;
}
#endif  // __HIPCC__
