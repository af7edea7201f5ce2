// fp64 L-BFGS-B (Byrd, Lu, Nocedal, Zhu; algorithm of L-BFGS-B 3.0 as SciPy 1.15 runs it) for the
// 8-unknown RTM3D box fit, written as straight-line host/device code: one thread = one object.
//
// The reference calls scipy.optimize.minimize(method='L-BFGS-B') (utils/model_utils.py:295-296)
// with m=10, factr=ftol/eps=1e7, pgtol=1e-5, maxls=20 and NO bounds (the `constraints=` it passes
// are ignored by L-BFGS-B).  The minimiser of the reprojection error is not unique,
// so the returned box depends on the optimiser's path; this file therefore follows the published
// algorithm step by step for the unbounded case:
//   mainlb  : iteration driver, convergence tests, BFGS skip rule, memory refresh on failure
//   cauchy  : with an empty memory the generalised Cauchy point is x - g (theta = 1)
//   formk / cmprlb / subsm : subspace (here: full-space) Newton step through the compact
//             representation  B = theta*I - W M W'  (LEL' factorisation of the 2col x 2col matrix)
//   lnsrlb / dcsrch / dcstep : More'-Thuente line search, ftol=1e-3, gtol=0.9, xtol=0.1
//   matupd / formt : limited-memory update of S, Y, S'Y, S'S and the Cholesky factor of T
// All sums run in index order in fp64 with fp contraction off, like the scalar CPU code.
//
// Two forms of the search direction (argument `direct` of lb_minimize):
//   direct = 0  the published subspace step (formk / subsm, and formt as the positive-definiteness check): the form
//               SciPy runs and, since round 6, the product's DEFAULT (wave kernel: lbfgsb_wave_pub.h; this scalar form is its
//               cross-check, rtm3d_decode3d_reference_form, tests/host_lbfgsb.cpp);
//   direct = 1  the opt-in form: the same vector  -B^-1 g  from the two-loop recursion over the same stored pairs
//               (lb_two_loop).  Without bounds the two are equal in exact arithmetic; in fp64 the iterates differ in the
//               last bits, and now and then one form stops an iteration before the other.  Against the reference's SciPy
//               results (keep / reject identical for both forms on every fixture; objects the reference rejects wander
//               equally in both), largest difference of a KEPT box: the 350 small golden objects 2.4e-7 for both; the
//               1536-object fixture (876 kept) direct 2.7e-5, published 1.3e-5; the bench's 111 planted boxes direct 1.6e-4
//               (one object's yaw: 99.1 % within north_star's 1e-4), published 4.7e-7; three times 1500 synthetic cuboids on
//               the host (profiles/r03_solver_forms_vs_scipy.txt): BOTH end further than 1e-4 from SciPy on 0.1-0.3 % of the
//               kept objects.  The direct form costs about 40 % of the dependent fp64 operations per iteration; it was the
//               default of the product path in rounds 3-5 and is selectable since (form = RTM3D_SOLVER_DIRECT).
//
// The objective / gradient restate aimFun (utils/model_utils.py:155-177, cost=1e-4) and jac
// (:206-234, cost=1e-6) in the reference's operation order.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define LB_HD __host__ __device__
#else
#define LB_HD
#endif

#define LB_N 8
#define LB_M 10
#define LB_M2 20

struct LbProblem {
    double k00, k02, k11, k12;   // intrinsics fx, cx, fy, cy
    double uv[16];               // 8 x (u, v), corner order of utils/model_utils.py:275-281
};

// corner signs (x, y, z) * 0.5, i/j/k nested loops over {1,-1}
LB_HD static inline void lb_corner(int c, double* cx, double* cy, double* cz) {
    *cx = (c & 4) ? -0.5 : 0.5;
    *cy = (c & 2) ? -0.5 : 0.5;
    *cz = (c & 1) ? -0.5 : 0.5;
}

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

LB_HD static inline double lb_fun(const LbProblem* p, const double* x) {
    const double cost = 1e-4;
    double obj = 0.0;
    for (int c = 0; c < 8; ++c) {
        double c0, c1, c2;
        lb_corner(c, &c0, &c1, &c2);
        const double xc = c0 * x[2] * x[1] + c2 * x[4] * x[0] + x[5];
        const double yc = c1 * x[3] + x[6];
        const double zc = (-c0) * x[2] * x[0] + c2 * x[4] * x[1] + x[7];
        const double ex = xc * p->k00 / (zc + cost) + p->k02 - p->uv[2 * c];
        const double ey = yc * p->k11 / (zc + cost) + p->k12 - p->uv[2 * c + 1];
        obj += ex * ex;
        obj += ey * ey;
    }
    return obj;
}

LB_HD static inline void lb_grad(const LbProblem* p, const double* x, double* g) {
    const double cost = 1e-6;
    for (int i = 0; i < 8; ++i) g[i] = 0.0;
    for (int c = 0; c < 8; ++c) {
        double c0, c1, c2;
        lb_corner(c, &c0, &c1, &c2);
        const double xc = c0 * x[2] * x[1] + c2 * x[4] * x[0] + x[5];
        const double yc = c1 * x[3] + x[6];
        const double zc = (-c0) * x[2] * x[0] + c2 * x[4] * x[1] + x[7];
        const double dex = (xc * p->k00 / (zc + cost) + p->k02 - p->uv[2 * c]) * 2;
        const double dey = (yc * p->k11 / (zc + cost) + p->k12 - p->uv[2 * c + 1]) * 2;
        const double dx[8] = {c2 * x[4], c0 * x[2], c0 * x[1], 0, c2 * x[0], 1, 0, 0};
        const double dy[8] = {0, 0, 0, c1, 0, 0, 1, 0};
        const double dz[8] = {(-c0) * x[2], c2 * x[4], (-c0) * x[0], 0, c2 * x[1], 0, 0, 1};
        const double den = zc * zc + cost;
        for (int i = 0; i < 8; ++i) {
            const double gx = p->k00 * (dx[i] * zc - dz[i] * xc) / den;
            const double gy = p->k11 * (dy[i] * zc - dz[i] * yc) / den;
            g[i] += dex * gx + dey * gy;
        }
    }
}

// ------------------------------------------------------------------------------------------
// workspace (1-based (row, col) accessors over column-major storage, like the published code)
struct LbWork {
    double ws[LB_N * LB_M], wy[LB_N * LB_M];
    double sy[LB_M * LB_M], ss[LB_M * LB_M], wt[LB_M * LB_M];
    double wn[LB_M2 * LB_M2], wn1[LB_M2 * LB_M2];
    double z[LB_N], r[LB_N], d[LB_N], t[LB_N], g[LB_N], wv[LB_M2];
    double rho[LB_M];            // 1 / (s'y) of the pair in ring slot p (direct form of the step, lb_two_loop)
};
#define WS_(i, j) w->ws[((j)-1) * LB_N + (i)-1]
#define WY_(i, j) w->wy[((j)-1) * LB_N + (i)-1]
#define SY_(i, j) w->sy[((j)-1) * LB_M + (i)-1]
#define SS_(i, j) w->ss[((j)-1) * LB_M + (i)-1]
#define WT_(i, j) w->wt[((j)-1) * LB_M + (i)-1]
#define WN_(i, j) w->wn[((j)-1) * LB_M2 + (i)-1]
#define WN1_(i, j) w->wn1[((j)-1) * LB_M2 + (i)-1]

// Reciprocal square root to ~1 ulp: hardware estimate (v_rsq_f64) on the device, 1/sqrt on the host, then two Newton
// steps y += y * (1/2 - (x/2) y^2) in fused arithmetic.  ONE long dependent fp64 operation per Cholesky pivot instead of a
// square root followed by a division (the solver is a serial chain of such pivots: DESIGN.md section 3, decode3d).
LB_HD static inline double lb_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
#else
    double y = 1.0 / sqrt(x);
#endif
    const double h = 0.5 * x;
    double e = fma(-(h * y), y, 0.5);
    y = fma(y, e, y);
    e = fma(-(h * y), y, 0.5);
    y = fma(y, e, y);
    return y;
}

// Cholesky A = U'U of the leading n x n block of a column-major matrix with leading dimension
// lda (upper triangle in/out).  Unblocked left-looking form: u_jj = sqrt(a_jj - u_j'u_j),
// row j scaled by the reciprocal.  THE DIAGONAL OF THE RESULT HOLDS 1 / u_jj: its only readers are the triangular solves
// below, which multiply by it (a division per substitution step was 30 % of an L-BFGS-B iteration on the GPU).
// Returns 0 or the failing 1-based column.
LB_HD static inline int lb_potrf(double* a, int lda, int n) {
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int k = 0; k < j; ++k) s += a[j * lda + k] * a[j * lda + k];
        const double ajj = a[j * lda + j] - s;
        if (!(ajj > 0.0)) { a[j * lda + j] = ajj; return j + 1; }
        const double rinv = lb_rsqrt(ajj);
        a[j * lda + j] = rinv;
        for (int i = j + 1; i < n; ++i) {
            double dot = 0.0;
            for (int k = 0; k < j; ++k) dot += a[j * lda + k] * a[i * lda + k];
            a[i * lda + j] = (a[i * lda + j] - dot) * rinv;
        }
    }
    return 0;
}
// solve U' x = b (forward) and U x = b (backward), U upper triangular with RECIPROCAL diagonal (lb_potrf), in place.
LB_HD static inline int lb_trsv_ut(const double* a, int lda, int n, double* b) {
    for (int j = 0; j < n; ++j) {
        if (a[j * lda + j] == 0.0) return j + 1;
        double dot = 0.0;
        for (int k = 0; k < j; ++k) dot += a[j * lda + k] * b[k];
        b[j] = (b[j] - dot) * a[j * lda + j];
    }
    return 0;
}
LB_HD static inline int lb_trsv_un(const double* a, int lda, int n, double* b) {
    for (int j = n - 1; j >= 0; --j) {
        if (a[j * lda + j] == 0.0) return j + 1;
        b[j] = b[j] * a[j * lda + j];
        const double tmp = -b[j];
        for (int k = 0; k < j; ++k) b[k] += tmp * a[j * lda + k];
    }
    return 0;
}

// ---- dcstep / dcsrch (MINPACK-2) ---------------------------------------------------------
struct LbSearch {
    int brackt, stage;
    double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
};
enum { LS_FG = 0, LS_CONV = 1, LS_WARN = 2, LS_ERROR = 3 };

LB_HD static inline double lb_max3(double a, double b, double c) { return fmax(fmax(a, b), c); }
// status of a start point whose objective or gradient is NaN / Inf (0 converged, 1 iteration limit, 2 abnormal line search)
#define LB_STATUS_NONFINITE 3
LB_HD static inline bool lb_isfinite(double v) { return (v - v) == 0.0; }      // false for NaN and +-Inf, no libm call

LB_HD static inline void lb_dcstep(double* stx, double* fx, double* dx, double* sty, double* fy, double* dy,
                                   double* stp, double fp, double dp, int* brackt, double stpmin, double stpmax) {
    const double sgnd = dp * (*dx / fabs(*dx));
    double theta, s, gamma, p, q, r, stpc, stpq, stpf;
    if (fp > *fx) {
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = lb_max3(fabs(theta), fabs(*dx), fabs(dp));
        gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp < *stx) gamma = -gamma;
        p = (gamma - *dx) + theta;
        q = ((gamma - *dx) + gamma) + dp;
        r = p / q;
        stpc = *stx + r * (*stp - *stx);
        stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.0) * (*stp - *stx);
        if (fabs(stpc - *stx) < fabs(stpq - *stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2.0;
        *brackt = 1;
    } else if (sgnd < 0.0) {
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = lb_max3(fabs(theta), fabs(*dx), fabs(dp));
        gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = ((gamma - dp) + gamma) + *dx;
        r = p / q;
        stpc = *stp + r * (*stx - *stp);
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc;
        else stpf = stpq;
        *brackt = 1;
    } else if (fabs(dp) < fabs(*dx)) {
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = lb_max3(fabs(theta), fabs(*dx), fabs(dp));
        gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = (gamma + (*dx - dp)) + gamma;
        r = p / q;
        if (r < 0.0 && gamma != 0.0) stpc = *stp + r * (*stx - *stp);
        else if (*stp > *stx) stpc = stpmax;
        else stpc = stpmin;
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (*brackt) {
            if (fabs(stpc - *stp) < fabs(stpq - *stp)) stpf = stpc;
            else stpf = stpq;
            if (*stp > *stx) stpf = fmin(*stp + 0.66 * (*sty - *stp), stpf);
            else stpf = fmax(*stp + 0.66 * (*sty - *stp), stpf);
        } else {
            if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc;
            else stpf = stpq;
            stpf = fmin(stpmax, stpf);
            stpf = fmax(stpmin, stpf);
        }
    } else {
        if (*brackt) {
            theta = 3.0 * (fp - *fy) / (*sty - *stp) + *dy + dp;
            s = lb_max3(fabs(theta), fabs(*dy), fabs(dp));
            gamma = s * sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
            if (*stp > *sty) gamma = -gamma;
            p = (gamma - dp) + theta;
            q = ((gamma - dp) + gamma) + *dy;
            r = p / q;
            stpc = *stp + r * (*sty - *stp);
            stpf = stpc;
        } else if (*stp > *stx) stpf = stpmax;
        else stpf = stpmin;
    }
    // interval update by value selection (with stores through either the x or the y pointers per branch the compiler
    // selects the POINTER at run time and the six doubles end up in scratch memory on the GPU)
    const bool hi = fp > *fx, flip = !hi && sgnd < 0.0;
    const double stx0 = *stx, fx0 = *fx, dx0 = *dx, stp0 = *stp;
    const double nsty = hi ? stp0 : (flip ? stx0 : *sty), nfy = hi ? fp : (flip ? fx0 : *fy), ndy = hi ? dp : (flip ? dx0 : *dy);
    const double nstx = hi ? stx0 : stp0, nfx = hi ? fx0 : fp, ndx = hi ? dx0 : dp;
    *sty = nsty; *fy = nfy; *dy = ndy;
    *stx = nstx; *fx = nfx; *dx = ndx;
    *stp = stpf;
}

// one call of dcsrch; `start` != 0 is task='START'.  Returns LS_*.
LB_HD static inline int lb_dcsrch(double f, double g, double* stp, double ftol, double gtol, double xtol,
                                  double stpmin, double stpmax, int start, LbSearch* S) {
    const double xtrapl = 1.1, xtrapu = 4.0, p5 = 0.5, p66 = 0.66;
    if (start) {
        if (*stp < stpmin || *stp > stpmax || g >= 0.0) return LS_ERROR;
        S->brackt = 0; S->stage = 1;
        S->finit = f; S->ginit = g; S->gtest = ftol * S->ginit;
        S->width = stpmax - stpmin; S->width1 = S->width / p5;
        S->stx = 0.0; S->fx = S->finit; S->gx = S->ginit;
        S->sty = 0.0; S->fy = S->finit; S->gy = S->ginit;
        S->stmin = 0.0; S->stmax = *stp + xtrapu * *stp;
        return LS_FG;
    }
    const double ftest = S->finit + *stp * S->gtest;
    if (S->stage == 1 && f <= ftest && g >= 0.0) S->stage = 2;
    int task = LS_FG;
    if (S->brackt && (*stp <= S->stmin || *stp >= S->stmax)) task = LS_WARN;
    if (S->brackt && S->stmax - S->stmin <= xtol * S->stmax) task = LS_WARN;
    if (*stp == stpmax && f <= ftest && g <= S->gtest) task = LS_WARN;
    if (*stp == stpmin && (f > ftest || g >= S->gtest)) task = LS_WARN;
    if (f <= ftest && fabs(g) <= gtol * (-S->ginit)) task = LS_CONV;
    if (task != LS_FG) return task;
    // One call of dcstep on local copies (selected VALUES, not selected pointers: with `&fxm` / `&S->fx` chosen per
    // branch the compiler merged the two inlined call sites and kept the bracket state in scratch memory behind a
    // run-time pointer - 2 x 40 B of private memory per lane on the GPU, touched in every line-search step).
    const bool modified = S->stage == 1 && f <= S->fx && f > ftest;
    double fx_l = S->fx, fy_l = S->fy, gx_l = S->gx, gy_l = S->gy, fp_l = f, gp_l = g;
    if (modified) {
        fp_l = f - *stp * S->gtest;
        fx_l = S->fx - S->stx * S->gtest; fy_l = S->fy - S->sty * S->gtest;
        gp_l = g - S->gtest;
        gx_l = S->gx - S->gtest; gy_l = S->gy - S->gtest;
    }
    lb_dcstep(&S->stx, &fx_l, &gx_l, &S->sty, &fy_l, &gy_l, stp, fp_l, gp_l, &S->brackt, S->stmin, S->stmax);
    if (modified) {
        S->fx = fx_l + S->stx * S->gtest; S->fy = fy_l + S->sty * S->gtest;
        S->gx = gx_l + S->gtest; S->gy = gy_l + S->gtest;
    } else {
        S->fx = fx_l; S->fy = fy_l; S->gx = gx_l; S->gy = gy_l;
    }
    if (S->brackt) {
        if (fabs(S->sty - S->stx) >= p66 * S->width1) *stp = S->stx + p5 * (S->sty - S->stx);
        S->width1 = S->width;
        S->width = fabs(S->sty - S->stx);
    }
    if (S->brackt) {
        S->stmin = fmin(S->stx, S->sty); S->stmax = fmax(S->stx, S->sty);
    } else {
        S->stmin = *stp + xtrapl * (*stp - S->stx); S->stmax = *stp + xtrapu * (*stp - S->stx);
    }
    *stp = fmax(*stp, stpmin);
    *stp = fmin(*stp, stpmax);
    if ((S->brackt && (*stp <= S->stmin || *stp >= S->stmax)) || (S->brackt && S->stmax - S->stmin <= xtol * S->stmax))
        *stp = S->stx;
    return LS_FG;
}

LB_HD static inline double lb_dot8(const double* a, const double* b) {
    double s = 0.0;
    for (int i = 0; i < LB_N; ++i) s += a[i] * b[i];
    return s;
}

// ---- formk (all variables free, no bound changes) ------------------------------------------
LB_HD static inline int lb_formk(LbWork* w, int iupdat, int updatd, double theta, int col, int head) {
    const int m = LB_M, n = LB_N;
    if (updatd) {
        if (iupdat > m) {   // shift the old part of WN1
            for (int jy = 1; jy <= m - 1; ++jy) {
                const int js = m + jy;
                for (int k = 0; k < m - jy; ++k) WN1_(jy + k, jy) = WN1_(jy + 1 + k, jy + 1);
                for (int k = 0; k < m - jy; ++k) WN1_(js + k, js) = WN1_(js + 1 + k, js + 1);
                for (int k = 0; k < m - 1; ++k) WN1_(m + 1 + k, jy) = WN1_(m + 2 + k, jy + 1);
            }
        }
        int ipntr = head + col - 1;
        if (ipntr > m) ipntr -= m;
        const int iy = col, is = m + col;
        int jpntr = head;
        for (int jy = 1; jy <= col; ++jy) {
            const int js = m + jy;
            double temp1 = 0.0;
            for (int k = 1; k <= n; ++k) temp1 += WY_(k, ipntr) * WY_(k, jpntr);
            WN1_(iy, jy) = temp1;       // row `col` of Y'Y
            WN1_(is, js) = 0.0;         // S'AA'S: no active variables
            WN1_(is, jy) = 0.0;         // L_a
            jpntr = jpntr % m + 1;
        }
        const int jy = col;
        jpntr = head + col - 1;
        if (jpntr > m) jpntr -= m;
        ipntr = head;
        for (int i = 1; i <= col; ++i) {
            double temp3 = 0.0;
            for (int k = 1; k <= n; ++k) temp3 += WS_(k, ipntr) * WY_(k, jpntr);
            ipntr = ipntr % m + 1;
            WN1_(m + i, jy) = temp3;    // column `col` of R_z
        }
    }
    // the "modify old parts" loops add exact zeros when no variable enters/leaves: omitted.
    for (int iy = 1; iy <= col; ++iy) {
        const int is = col + iy, is1 = m + iy;
        for (int jy = 1; jy <= iy; ++jy) {
            const int js = col + jy, js1 = m + jy;
            WN_(jy, iy) = WN1_(iy, jy) / theta;
            WN_(js, is) = WN1_(is1, js1) * theta;
        }
        for (int jy = 1; jy <= iy - 1; ++jy) WN_(jy, is) = -WN1_(is1, jy);
        for (int jy = iy; jy <= col; ++jy) WN_(jy, is) = WN1_(is1, jy);
        WN_(iy, iy) = WN_(iy, iy) + SY_(iy, iy);
    }
    if (lb_potrf(w->wn, LB_M2, col) != 0) return -1;
    const int col2 = 2 * col;
    for (int js = col + 1; js <= col2; ++js)
        if (lb_trsv_ut(w->wn, LB_M2, col, &WN_(1, js)) != 0) return -1;
    for (int is = col + 1; is <= col2; ++is)
        for (int js = is; js <= col2; ++js) {
            double dot = 0.0;
            for (int k = 1; k <= col; ++k) dot += WN_(k, is) * WN_(k, js);
            WN_(is, js) = WN_(is, js) + dot;
        }
    if (lb_potrf(&WN_(col + 1, col + 1), LB_M2, col) != 0) return -2;
    return 0;
}

// ---- subsm: z <- x + Newton direction of the quadratic model (r holds -g on entry) ---------
LB_HD static inline int lb_subsm(LbWork* w, double theta, int col, int head) {
    const int m = LB_M, n = LB_N;
    double* d = w->r;
    double* wv = w->wv;
    int pointr = head;
    for (int i = 1; i <= col; ++i) {
        double temp1 = 0.0, temp2 = 0.0;
        for (int j = 1; j <= n; ++j) {
            temp1 += WY_(j, pointr) * d[j - 1];
            temp2 += WS_(j, pointr) * d[j - 1];
        }
        wv[i - 1] = temp1;
        wv[col + i - 1] = theta * temp2;
        pointr = pointr % m + 1;
    }
    const int col2 = 2 * col;
    if (lb_trsv_ut(w->wn, LB_M2, col2, wv) != 0) return 1;
    for (int i = 0; i < col; ++i) wv[i] = -wv[i];
    if (lb_trsv_un(w->wn, LB_M2, col2, wv) != 0) return 1;
    pointr = head;
    const double rt = 1.0 / theta;       // (multiplied in below instead of col divisions per component)
    for (int jy = 1; jy <= col; ++jy) {
        const int js = col + jy;
        for (int i = 1; i <= n; ++i)
            d[i - 1] = d[i - 1] + WY_(i, pointr) * wv[jy - 1] * rt + WS_(i, pointr) * wv[js - 1];
        pointr = pointr % m + 1;
    }
    for (int i = 0; i < n; ++i) d[i] = rt * d[i];
    for (int i = 0; i < n; ++i) w->z[i] = w->z[i] + d[i];
    return 0;
}

// ---- matupd + formt ---------------------------------------------------------------------
LB_HD static inline void lb_matupd(LbWork* w, int* itail, int iupdat, int* col, int* head, double* theta,
                                   double rr, double dr, double stp, double dtd) {
    const int m = LB_M, n = LB_N;
    if (iupdat <= m) {
        *col = iupdat;
        *itail = (*head + iupdat - 2) % m + 1;
    } else {
        *itail = *itail % m + 1;
        *head = *head % m + 1;
    }
    for (int i = 1; i <= n; ++i) { WS_(i, *itail) = w->d[i - 1]; WY_(i, *itail) = w->r[i - 1]; }
    w->rho[*itail - 1] = 1.0 / dr;
    *theta = rr / dr;
    if (iupdat > m) {
        for (int j = 1; j <= *col - 1; ++j) {
            for (int k = 0; k < j; ++k) SS_(1 + k, j) = SS_(2 + k, j + 1);
            for (int k = 0; k < *col - j; ++k) SY_(j + k, j) = SY_(j + 1 + k, j + 1);
        }
    }
    int pointr = *head;
    for (int j = 1; j <= *col - 1; ++j) {
        SY_(*col, j) = lb_dot8(w->d, &WY_(1, pointr));
        SS_(j, *col) = lb_dot8(&WS_(1, pointr), w->d);
        pointr = pointr % m + 1;
    }
    if (stp == 1.0) SS_(*col, *col) = dtd;
    else SS_(*col, *col) = stp * stp * dtd;
    SY_(*col, *col) = dr;
}

LB_HD static inline int lb_formt(LbWork* w, int col, double theta) {
    double rsy[LB_M];                    // 1 / SY(k, k): one division per column instead of one per term
    for (int k = 1; k <= col; ++k) rsy[k - 1] = 1.0 / SY_(k, k);
    for (int j = 1; j <= col; ++j) WT_(1, j) = theta * SS_(1, j);
    for (int i = 2; i <= col; ++i)
        for (int j = i; j <= col; ++j) {
            const int k1 = (i < j ? i : j) - 1;
            double ddum = 0.0;
            for (int k = 1; k <= k1; ++k) ddum = ddum + SY_(i, k) * SY_(j, k) * rsy[k - 1];
            WT_(i, j) = ddum + theta * SS_(i, j);
        }
    return lb_potrf(w->wt, LB_M, col) != 0 ? -3 : 0;
}

// ---- direct form of the same step ---------------------------------------------------------
// With no bounds the subspace step of formk / subsm is z = x - B^-1 g for the limited-memory BFGS matrix
// B = theta*I - W M W' of the stored pairs.  H = B^-1 applied to -g by the two-loop recursion over the same pairs with
// H0 = I / theta is the same vector in exact arithmetic (Nocedal 1980; Byrd, Nocedal, Schnabel 1994, section 3) at
// ~2 * col dot products of length 8 instead of forming and factorising the 2col x 2col matrix K.
// dot product of the two-loop recursion: pairwise (tree) order, three dependent additions instead of eight (the recursion is
// one chain of 2 * col such dots; lbfgsb_wave.h evaluates the same expression)
LB_HD static inline double lb_dot8t(const double* a, const double* b) {
    const double p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2], p3 = a[3] * b[3];
    const double p4 = a[4] * b[4], p5 = a[5] * b[5], p6 = a[6] * b[6], p7 = a[7] * b[7];
    return ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));
}

LB_HD static inline void lb_two_loop(LbWork* w, const double* x, const double* g, double theta, int col, int head) {
    const int m = LB_M, n = LB_N;
    double q[LB_N], alpha[LB_M];
    for (int i = 0; i < n; ++i) q[i] = -g[i];
    int p = head + col - 1;
    if (p > m) p -= m;
    for (int j = col; j >= 1; --j) {            // newest pair first
        const double a = lb_dot8t(&WS_(1, p), q) * w->rho[p - 1];
        alpha[j - 1] = a;
        for (int i = 0; i < n; ++i) q[i] = q[i] - a * WY_(i + 1, p);
        p = p - 1;
        if (p < 1) p += m;
    }
    const double rt = 1.0 / theta;
    for (int i = 0; i < n; ++i) q[i] = rt * q[i];
    p = head;
    for (int j = 1; j <= col; ++j) {            // oldest pair first
        const double b = lb_dot8t(&WY_(1, p), q) * w->rho[p - 1];
        const double c = alpha[j - 1] - b;
        for (int i = 0; i < n; ++i) q[i] = q[i] + c * WS_(i + 1, p);
        p = p % m + 1;
    }
    for (int i = 0; i < n; ++i) w->z[i] = x[i] + q[i];
}

// ---- driver ------------------------------------------------------------------------------
// status: 0 converged (pgtol or factr test), 1 iteration/evaluation limit, 2 abnormal line search
// direct != 0: the search direction comes from lb_two_loop instead of formk / subsm (and formt is skipped: without bounds
// its factor is only a positive-definiteness verdict).
LB_HD static inline int lb_minimize(const LbProblem* prob, double* x, double* f_out, int* nit_out, LbWork* w,
                                    int maxiter, int maxfun, int direct = 0) {
    const int n = LB_N, maxls = 20;
    const double epsmch = 2.220446049250313e-16, factr = 1e7, pgtol = 1e-5;
    const double ftol = 1e-3, gtol = 0.9, xtol = 0.1, big = 1e10;
    const double tol = factr * epsmch;
    int col = 0, head = 1, itail = 0, iupdat = 0, updatd = 0, iter = 0, nfgv = 0, info;
    double theta = 1.0, f, fold = 0.0, gd = 0.0, gdold = 0.0, stp = 0.0, dnorm = 0.0, dtd = 0.0, sbgnrm;
    double* g = w->g;
    LbSearch S;

    f = lb_fun(prob, x); lb_grad(prob, x, g); nfgv = 1;
    // Non-finite key points (NaN / Inf logits upstream): the objective is not a number at the start point.  SciPy returns
    // x0 with fun = nan after 0 iterations there (the oracle run on such vertices); so does this, with its own status, instead
    // of feeding NaNs to the line search (fmax below would drop them and compare garbage).
    {
        bool finite = lb_isfinite(f);
        for (int i = 0; i < n; ++i) finite = finite && lb_isfinite(g[i]);
        if (!finite) { *f_out = f; *nit_out = 0; return LB_STATUS_NONFINITE; }
    }
    sbgnrm = 0.0;
    for (int i = 0; i < n; ++i) sbgnrm = fmax(sbgnrm, fabs(g[i]));
    if (sbgnrm <= pgtol) { *f_out = f; *nit_out = 0; return 0; }

    for (;;) {
        // ---- search direction: z = minimiser of the quadratic model (Cauchy point if memory empty)
        if (col == 0) {
            for (int i = 0; i < n; ++i) w->z[i] = x[i] + 1.0 * (-g[i]);
        } else if (direct) {
            lb_two_loop(w, x, g, theta, col, head);
        } else {
            for (int i = 0; i < n; ++i) w->z[i] = x[i];
            info = 0;
            if (updatd) info = lb_formk(w, iupdat, updatd, theta, col, head);
            if (info == 0) {
                for (int i = 0; i < n; ++i) w->r[i] = -g[i];
                info = lb_subsm(w, theta, col, head);
            }
            if (info != 0) {   // refresh the memory and restart the iteration
                col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0;
                continue;
            }
        }
        for (int i = 0; i < n; ++i) w->d[i] = w->z[i] - x[i];
        // ---- line search (lnsrlb)
        dtd = lb_dot8(w->d, w->d);
        dnorm = sqrt(dtd);
        const double stpmx = big;
        stp = (iter == 0) ? fmin(1.0 / dnorm, stpmx) : 1.0;
        for (int i = 0; i < n; ++i) { w->t[i] = x[i]; w->r[i] = g[i]; }
        fold = f;
        int ifun = 0, iback = 0, ls_fail = 0, start = 1;
        info = 0;
        for (;;) {
            gd = lb_dot8(g, w->d);
            if (ifun == 0) {
                gdold = gd;
                if (gd >= 0.0) { info = -4; break; }
            }
            const int task = lb_dcsrch(f, gd, &stp, ftol, gtol, xtol, 0.0, stpmx, start, &S);
            start = 0;
            if (task == LS_ERROR) { info = -4; break; }
            if (task == LS_CONV || task == LS_WARN) break;
            ifun += 1; nfgv += 1; iback = ifun - 1;
            if (stp == 1.0) { for (int i = 0; i < n; ++i) x[i] = w->z[i]; }
            else { for (int i = 0; i < n; ++i) x[i] = stp * w->d[i] + w->t[i]; }
            if (iback >= maxls) { ls_fail = 1; break; }
            f = lb_fun(prob, x); lb_grad(prob, x, g);
        }
        if (info != 0 || ls_fail) {
            for (int i = 0; i < n; ++i) { x[i] = w->t[i]; g[i] = w->r[i]; }
            f = fold;
            if (col == 0) { *f_out = f; *nit_out = iter; return 2; }
            col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0;
            continue;
        }
        // ---- new iterate
        iter += 1;
        sbgnrm = 0.0;
        for (int i = 0; i < n; ++i) sbgnrm = fmax(sbgnrm, fabs(g[i]));
        if (iter >= maxiter || nfgv > maxfun) { *f_out = f; *nit_out = iter; return 1; }
        if (sbgnrm <= pgtol) break;
        const double ddum0 = lb_max3(fabs(fold), fabs(f), 1.0);
        if ((fold - f) <= tol * ddum0) break;
        // ---- BFGS update
        for (int i = 0; i < n; ++i) w->r[i] = g[i] - w->r[i];
        const double rr = lb_dot8(w->r, w->r);
        double dr, ddum;
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else {
            dr = (gd - gdold) * stp;
            for (int i = 0; i < n; ++i) w->d[i] = stp * w->d[i];
            ddum = -gdold * stp;
        }
        if (dr <= epsmch * ddum) { updatd = 0; continue; }
        updatd = 1; iupdat += 1;
        lb_matupd(w, &itail, iupdat, &col, &head, &theta, rr, dr, stp, dtd);
        if (!direct && lb_formt(w, col, theta) != 0) { col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0; }
    }
    *f_out = f; *nit_out = iter;
    return 0;
}
