// Host build of the product's L-BFGS-B (rtm3d_amd/csrc/lbfgsb.h) so its logic can be tested on a
// machine without a GPU (tests/test_lbfgsb_host.py).  Not part of the product.
#include <stdint.h>
#include "../rtm3d_amd/csrc/lbfgsb.h"
static int solve(int N, const int64_t* cls, const float* verts, const double* K, const double* dim_ref,
                 const double* ref_loc, double* x_out, double* f_out, int* nit, int* status, int direct);
extern "C" int lb_solve_batch_direct(int N, const int64_t* cls, const float* verts, const double* K, const double* dim_ref,
                                     const double* ref_loc, double* x_out, double* f_out, int* nit, int* status) {
    return solve(N, cls, verts, K, dim_ref, ref_loc, x_out, f_out, nit, status, 1);
}
extern "C" int lb_solve_batch(int N, const int64_t* cls, const float* verts, const double* K, const double* dim_ref,
                              const double* ref_loc, double* x_out, double* f_out, int* nit, int* status) {
    return solve(N, cls, verts, K, dim_ref, ref_loc, x_out, f_out, nit, status, 0);
}
static int solve(int N, const int64_t* cls, const float* verts, const double* K, const double* dim_ref,
                 const double* ref_loc, double* x_out, double* f_out, int* nit, int* status, int direct) {
    for (int i = 0; i < N; ++i) {
        LbProblem p;
        p.k00 = K[i * 9 + 0]; p.k02 = K[i * 9 + 2]; p.k11 = K[i * 9 + 4]; p.k12 = K[i * 9 + 5];
        for (int j = 0; j < 16; ++j) p.uv[j] = (double)verts[i * 16 + j];
        const double* dim = dim_ref + cls[i] * 3;
        double x[8] = {0, 1, dim[2], dim[0], dim[1], ref_loc[0], ref_loc[1], ref_loc[2]};
        LbWork w;
        status[i] = lb_minimize(&p, x, &f_out[i], &nit[i], &w, 15000, 15000, direct);
        for (int j = 0; j < 8; ++j) x_out[i * 8 + j] = x[j];
    }
    return 0;
}
