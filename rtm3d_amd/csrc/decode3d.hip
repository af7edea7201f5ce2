// 3D box decode on device: the fp64 L-BFGS-B of lbfgsb.h, one wavefront per object - lbfgsb_wave_pub.h, the published subspace
// step SciPy runs (the default since round 6), or lbfgsb_wave.h, the direct two-loop search direction (opt-in) - or one lane per
// object (the cross-check kernels of the two forms, bit-identical to the wave kernels).
// Replaces optim_decode_bbox3d (utils/model_utils.py:264-312) + scipy L-BFGS-B.
// The work is latency-bound fp64 (a few hundred kFLOP per object, <= topk objects per image), so the
// kernel only needs enough lanes in flight: 64-lane workgroups, objects spread over the CUs.
#include <type_traits>
#include "common.h"
#include "lbfgsb.h"
#include "lbfgsb_wave.h"
#include "lbfgsb_wave_pub.h"
#include "../../include/rtm3d_hip.h"

// slot mode (n_per_image != nullptr): object i lives in slot (image = i / topk, rank = i % topk) of
// the decode2d outputs, is valid iff rank < n_per_image[image], and uses K[image].
template <int DIRECT>
__global__ __launch_bounds__(64) void decode3d_kernel(int N, const int64_t* __restrict__ cls,
                                                      const float* __restrict__ verts, const double* __restrict__ K,
                                                      const double* __restrict__ dim_ref, int ncls,
                                                      const double* __restrict__ ref_loc, double* __restrict__ x_out,
                                                      double* __restrict__ f_out, int32_t* __restrict__ nit,
                                                      int32_t* __restrict__ status,
                                                      const int32_t* __restrict__ n_per_image, int topk) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= N) return;
    int ki = i;
    if (n_per_image) {
        ki = i / topk;
        if (i - ki * topk >= n_per_image[ki]) { status[i] = -1; return; }
    }
    LbProblem p;
    p.k00 = K[ki * 9 + 0]; p.k02 = K[ki * 9 + 2]; p.k11 = K[ki * 9 + 4]; p.k12 = K[ki * 9 + 5];
    for (int j = 0; j < 16; ++j) p.uv[j] = (double)verts[(size_t)i * 16 + j];
    int c = (int)cls[i];
    c = c < 0 ? 0 : (c >= ncls ? ncls - 1 : c);
    const double* dim = dim_ref + c * 3;            // (h, w, l)  -> x0 = [0, 1, l, h, w, ref_loc]
    double x[8] = {0.0, 1.0, dim[2], dim[0], dim[1], ref_loc[0], ref_loc[1], ref_loc[2]};
    LbWork w;
    double f;
    int it;
    const int st = lb_minimize(&p, x, &f, &it, &w, 15000, 15000, DIRECT);
    for (int j = 0; j < 8; ++j) x_out[(size_t)i * 8 + j] = x[j];
    f_out[i] = f;
    nit[i] = it;
    status[i] = st;
}

// One wavefront per object: the production kernel.  A workgroup packs D3_WPB objects (8 waves; direct form 20 KB of
// LDS, 116 VGPRs; published form 67 KB, 135): the ~15 objects of an image then sit on two CUs instead of fifteen, which matters when this kernel runs
// beside the forward pass of the next batch - a CU that holds these waves cannot take a workgroup of the persistent conv
// kernels (160 KB of LDS) until they retire.  Objects per workgroup, pipelined ms/step at bs=1 serial / bs=32 with 473 /
// with 3200 objects, pairs measured on one box each (profiles/r02_decode3d_load.json): 8 vs 16: 1.57 / 14.04 / 14.97 vs
// 1.65 / 14.18 / 15.16; 8 vs 4: 1.57 / 14.06 / 15.01 vs 1.52 / 14.05 / 15.01; 8 vs 2: 1.57 / 14.01 / 15.01 vs 1.53 / 14.04 / 15.06:
// flat between 2 and 8, 8 kept.  (With the published subspace step - 178 VGPRs, 67 KB per 8 objects - 12 and 16 objects per
// workgroup needed VGPR caps that spilled: 15.62 / 16.02 / 16.12 ms per step at 473 objects.)
// Round 6, with the published form as the default (then 239 registers x 8 waves - 135 since its Cholesky factorisations run in
// registers - and 67 KB of LDS per workgroup: a CU that holds one takes no workgroup of the next batch's persistent conv kernels),
// pipelined bs=32 ms/step, interleaved on one box each (all measured with the 239-register kernel):
//   objects per workgroup 2 / 4 / 8 / 10 / 12 (12 spill-free at 168 registers after the register diet of lbfgsb_wave_pub.h):
//     12.98 / 12.95 / 12.69 / 12.90 / 13.19 - 8 stays;
//   a WORK QUEUE (G persistent workgroups whose waves draw the valid slots from a ticket counter; bit-identical results): the
//     bench workload's 474 objects need 34 iterations on average, p99 82, max 207 (tools/gpu_solver_nit.py), so the 82 workgroups
//     with work of this kernel hold their CUs for twice the work they do - but G = 8 / 16 / 24 / 32 / 48 / 64 / 96 / 400 ran
//     13.1 / 12.95 / 12.85 / 12.83 / 12.85 / 12.93 / 12.95 / 13.11 against 12.70-12.77 for this kernel.  The kernel traces say why
//     (tools/gpu_trace_variants.sh): what the decode costs the next forward is mostly the SECOND ROUND it forces on layers that
//     run one round of 240 tiles on 256 CUs (DLA level 4: 65 -> 100-110 us per conv while more than two CUs of an XCD are held),
//     and a queue starts its last objects late - at + 1.5 ms most of its workgroups still hold one running object and all eight
//     level-4 convs pay, where this launch (everything starts at once) has 6 of 82 workgroups left by then and only the first
//     three pay (the stem and level 2 beside the full 82: + 0.13 / + 0.10 ms);
//   the valid slots packed densely into 60 full workgroups (wave t takes the t-th valid slot, prefix sums over the images by a
//     wave scan) instead of 82 partly filled ones: 13.05-13.12 against 12.80-12.86.
// None kept.
#ifndef D3_WPB_DEFAULT
#define D3_WPB_DEFAULT 8
#endif
#ifndef D3_WPB_PUB
#define D3_WPB_PUB 8
#endif
// FORM 0: the direct form (two-loop search direction, lbfgsb_wave.h), opt-in;
// FORM 1: the published subspace step (lbfgsb_wave_pub.h: what SciPy runs), the default of every entry since round 6:
// 8.4 KB of LDS per object, i.e. 67 KB per workgroup of eight.
template <int D3_WPB, int FORM>
__global__ __launch_bounds__(64 * D3_WPB) void decode3d_wave_kernel(int N, const int64_t* __restrict__ cls,
                                                           const float* __restrict__ verts, const double* __restrict__ K,
                                                           const double* __restrict__ dim_ref, int ncls,
                                                           const double* __restrict__ ref_loc, double* __restrict__ x_out,
                                                           double* __restrict__ f_out, int32_t* __restrict__ nit,
                                                           int32_t* __restrict__ status,
                                                           const int32_t* __restrict__ n_per_image, int topk) {
    using Mem = typename std::conditional<FORM == 1, lbw_pub::LbWaveMem, LbWaveMem>::type;
    using KK = typename std::conditional<FORM == 1, lbw_pub::LbWaveK, LbWaveK>::type;
    __shared__ Mem mems[D3_WPB];
    // These waves are latency chains that issue an instruction every few hundred cycles; beside the MFMA / DMA
    // waves of the next batch's convolutions they lose every arbitration at equal priority and the kernel stretches
    // from 1.3 ms to ~5 ms, into the persistent conv kernels that need whole CUs.  Highest wave priority: the
    // chains run at their own pace, the co-resident conv waves give up a few issue slots.
    __builtin_amdgcn_s_setprio(3);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    Mem& mem = mems[wave];
    const int i = blockIdx.x * D3_WPB + wave;
    if (i >= N) return;
    int ki = i;
    if (n_per_image) {
        ki = i / topk;
        if (i - ki * topk >= n_per_image[ki]) { if (lane == 0) status[i] = -1; return; }
    }
    KK Kk;
    Kk.k00 = K[ki * 9 + 0]; Kk.k02 = K[ki * 9 + 2]; Kk.k11 = K[ki * 9 + 4]; Kk.k12 = K[ki * 9 + 5];
    int c = (int)cls[i];
    c = c < 0 ? 0 : (c >= ncls ? ncls - 1 : c);
    const double* dim = dim_ref + c * 3;
    if (lane < 16) mem.uv[lane] = (double)verts[(size_t)i * 16 + lane];
    if (lane == 0) {
        mem.x[0] = 0.0; mem.x[1] = 1.0; mem.x[2] = dim[2]; mem.x[3] = dim[0]; mem.x[4] = dim[1];
        mem.x[5] = ref_loc[0]; mem.x[6] = ref_loc[1]; mem.x[7] = ref_loc[2];
    }
    WSYNC();
    double f;
    int it;
    int st;
    if constexpr (FORM == 1) st = lbw_pub::lbw_minimize(&mem, Kk, &f, &it, lane, 15000, 15000);
    else st = lbw_minimize(&mem, Kk, &f, &it, lane, 15000, 15000);
    WSYNC();
    if (lane < 8) x_out[(size_t)i * 8 + lane] = mem.x[lane];
    if (lane == 0) { f_out[i] = f; nit[i] = it; status[i] = st; }
}

extern void rt_set_error(const char* fmt, ...);

#define D3_LAUNCH(...) hipLaunchKernelGGL((decode3d_wave_kernel<D3_WPB_DEFAULT, 0>), dim3((N + D3_WPB_DEFAULT - 1) / D3_WPB_DEFAULT), \
                                          dim3(64 * D3_WPB_DEFAULT), 0, (hipStream_t)stream, __VA_ARGS__)
#define D3_LAUNCH_PUB(...) hipLaunchKernelGGL((decode3d_wave_kernel<D3_WPB_PUB, 1>), dim3((N + D3_WPB_PUB - 1) / D3_WPB_PUB), \
                                              dim3(64 * D3_WPB_PUB), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" int rtm3d_decode3d(void* stream, int N, const int64_t* d_cls, const float* d_verts, const double* d_K,
                              const double* d_dim_ref, int ncls, const double* d_ref_loc, double* d_x,
                              double* d_fun, int32_t* d_nit, int32_t* d_status, int form) {
    if (N < 0 || ncls <= 0) { rt_set_error("decode3d: bad sizes"); return 1; }
    if (form != RTM3D_SOLVER_DIRECT && form != RTM3D_SOLVER_PUBLISHED) { rt_set_error("decode3d: unknown solver form %d (0 = direct two-loop direction, 1 = published subspace step)", form); return 1; }
    if (N == 0) return 0;
    if (!d_cls || !d_verts || !d_K || !d_dim_ref || !d_ref_loc || !d_x || !d_fun || !d_nit || !d_status) {
        rt_set_error("decode3d: null pointer"); return 1;
    }
    if (form == RTM3D_SOLVER_PUBLISHED) D3_LAUNCH_PUB(N, d_cls, d_verts, d_K, d_dim_ref, ncls, d_ref_loc, d_x, d_fun, d_nit, d_status, (const int32_t*)nullptr, 0);
    else D3_LAUNCH(N, d_cls, d_verts, d_K, d_dim_ref, ncls, d_ref_loc, d_x, d_fun, d_nit, d_status, (const int32_t*)nullptr, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("decode3d launch: %s", hipGetErrorString(e)); return 1; }
    return 0;
}

extern "C" int rtm3d_decode3d_slots(void* stream, int B, int topk, const int32_t* d_n, const int64_t* d_cls,
                                    const float* d_verts, const double* d_K_per_image, const double* d_dim_ref, int ncls,
                                    const double* d_ref_loc, double* d_x, double* d_fun, int32_t* d_nit,
                                    int32_t* d_status, int form) {
    if (B <= 0 || topk <= 0 || ncls <= 0) { rt_set_error("decode3d_slots: bad sizes"); return 1; }
    if (form != RTM3D_SOLVER_DIRECT && form != RTM3D_SOLVER_PUBLISHED) { rt_set_error("decode3d_slots: unknown solver form %d (0 = direct two-loop direction, 1 = published subspace step)", form); return 1; }
    if (!d_n || !d_cls || !d_verts || !d_K_per_image || !d_dim_ref || !d_ref_loc || !d_x || !d_fun || !d_nit || !d_status) {
        rt_set_error("decode3d_slots: null pointer"); return 1;
    }
    const int N = B * topk;
    if (form == RTM3D_SOLVER_PUBLISHED) D3_LAUNCH_PUB(N, d_cls, d_verts, d_K_per_image, d_dim_ref, ncls, d_ref_loc, d_x, d_fun, d_nit, d_status, d_n, topk);
    else D3_LAUNCH(N, d_cls, d_verts, d_K_per_image, d_dim_ref, ncls, d_ref_loc, d_x, d_fun, d_nit, d_status, d_n, topk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("decode3d_slots launch: %s", hipGetErrorString(e)); return 1; }
    return 0;
}

// Same problem on the one-lane-per-object kernels (scalar lbfgsb.h): cross-checks of the wave kernel.
static int launch_scalar(int direct, void* stream, int N, const int64_t* d_cls, const float* d_verts, const double* d_K,
                         const double* d_dim_ref, int ncls, const double* d_ref_loc, double* d_x, double* d_fun, int32_t* d_nit,
                         int32_t* d_status, const char* what) {
    if (N <= 0 || ncls <= 0) { rt_set_error("%s: bad sizes", what); return 1; }
    if (!d_cls || !d_verts || !d_K || !d_dim_ref || !d_ref_loc || !d_x || !d_fun || !d_nit || !d_status) { rt_set_error("%s: null pointer", what); return 1; }
    if (direct)
        hipLaunchKernelGGL(decode3d_kernel<1>, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, N, d_cls, d_verts,
                           d_K, d_dim_ref, ncls, d_ref_loc, d_x, d_fun, d_nit, d_status, (const int32_t*)nullptr, 0);
    else
        hipLaunchKernelGGL(decode3d_kernel<0>, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, N, d_cls, d_verts,
                           d_K, d_dim_ref, ncls, d_ref_loc, d_x, d_fun, d_nit, d_status, (const int32_t*)nullptr, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("%s launch: %s", what, hipGetErrorString(e)); return 1; }
    return 0;
}

// the direct form's arithmetic (two-loop direction), one lane per object: bit-identical to rtm3d_decode3d(form = RTM3D_SOLVER_DIRECT)
extern "C" int rtm3d_decode3d_scalar(void* stream, int N, const int64_t* d_cls, const float* d_verts, const double* d_K,
                                     const double* d_dim_ref, int ncls, const double* d_ref_loc, double* d_x,
                                     double* d_fun, int32_t* d_nit, int32_t* d_status) {
    return launch_scalar(1, stream, N, d_cls, d_verts, d_K, d_dim_ref, ncls, d_ref_loc, d_x, d_fun, d_nit, d_status, "decode3d_scalar");
}

// the published subspace step (formk / subsm / formt), i.e. L-BFGS-B 3.0 as SciPy runs it, one lane per object
extern "C" int rtm3d_decode3d_reference_form(void* stream, int N, const int64_t* d_cls, const float* d_verts, const double* d_K,
                                             const double* d_dim_ref, int ncls, const double* d_ref_loc, double* d_x,
                                             double* d_fun, int32_t* d_nit, int32_t* d_status) {
    return launch_scalar(0, stream, N, d_cls, d_verts, d_K, d_dim_ref, ncls, d_ref_loc, d_x, d_fun, d_nit, d_status, "decode3d_reference_form");
}

// ------------------------------------------------------------------------------------------------------------
// Detection records for the multi-GPU all-gather: one 32-float record per (image, rank) slot, see
// rtm3d_amd/distributed.py for the layout.  One thread per float: stores are fully coalesced, the loads are a
// few scattered dwords per slot.  Arithmetic = the field definitions of utils/model_utils.py:300-303 (Ry = atan2,
// dimension = (h, w, l) = x[3], x[4], x[2], location = x[5:8]) rounded to fp32; empty slots are all zero.
__global__ __launch_bounds__(256) void pack_records_kernel(int total, int topk, const int32_t* __restrict__ n,
                                                          const int64_t* __restrict__ cls, const float* __restrict__ score,
                                                          const float* __restrict__ mproj, const float* __restrict__ verts,
                                                          const float* __restrict__ bbox, const double* __restrict__ x,
                                                          const double* __restrict__ fun, const int32_t* __restrict__ status,
                                                          double fun_accept, float* __restrict__ rec) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int slot = t >> 5, f = t & 31;
    const int img = slot / topk;
    float v = 0.0f;
    if (slot - img * topk < n[img]) {
        const bool kept = x && status[slot] >= 0 && fun[slot] < fun_accept;
        if (f == 0) v = (float)cls[slot];
        else if (f == 1) v = score[slot];
        else if (f < 4) v = mproj[(size_t)slot * 2 + (f - 2)];
        else if (f < 20) v = verts[(size_t)slot * 16 + (f - 4)];
        else if (f < 24) v = bbox[(size_t)slot * 4 + (f - 20)];
        else if (f == 31) v = kept ? 2.0f : 1.0f;
        else if (x) {
            const double* xs = x + (size_t)slot * 8;
            if (f < 27) v = (float)xs[f == 26 ? 2 : f - 21];           // (h, w, l) = x[3], x[4], x[2]
            else if (f < 30) v = (float)xs[f - 22];                    // X, Y, Z = x[5..7]
            else v = (float)atan2(xs[0], xs[1]);
        }
    }
    rec[t] = v;
}

extern "C" int rtm3d_pack_records(void* stream, int B, int topk, const int32_t* d_n, const int64_t* d_cls, const float* d_score,
                                  const float* d_mproj, const float* d_verts, const float* d_bbox, const double* d_x,
                                  const double* d_fun, const int32_t* d_status, double fun_accept, float* d_rec) {
    if (B <= 0 || topk <= 0) { rt_set_error("pack_records: bad sizes"); return 1; }
    if (!d_n || !d_cls || !d_score || !d_mproj || !d_verts || !d_bbox || !d_rec || (d_x && (!d_fun || !d_status))) {
        rt_set_error("pack_records: null pointer"); return 1;
    }
    const int total = B * topk * 32;
    hipLaunchKernelGGL(pack_records_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, total, topk, d_n, d_cls,
                       d_score, d_mproj, d_verts, d_bbox, d_x, d_fun, d_status, fun_accept, d_rec);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("pack_records launch: %s", hipGetErrorString(e)); return 1; }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Box post-processing on the device: the eight corners + the centre of every solved slot projected
// into the image, and the corners' bounding rectangle - calc_proj_corners / create_corners / rotation_matrix,
// utils/model_utils.py:66-152 (|sin|, |cos| < 1e-3 snap to 0; corner order i, j, k in {1, -1} nested, then the centre;
// divide by z + 1e-6).  One thread per (slot, corner); fp64 like the reference.  Slots with status < 0 get zeros.
__global__ __launch_bounds__(256) void project_boxes_kernel(int N, int topk, const double* __restrict__ x, const int32_t* __restrict__ status,
                                                            const double* __restrict__ K, double* __restrict__ proj, double* __restrict__ rect) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int slot = t / 9, c = t - slot * 9;
    if (slot >= N) return;
    double u = 0.0, v = 0.0;
    if (status[slot] >= 0) {
        const double* xs = x + (size_t)slot * 8;
        const double* k = K + (size_t)(topk > 0 ? slot / topk : slot) * 9;
        const double ry = atan2(xs[0], xs[1]);                      // :300
        double sn = sin(ry), cs = cos(ry);
        if (fabs(sn) < 1e-3) sn = 0.0;
        if (fabs(cs) < 1e-3) cs = 0.0;
        const double dx = xs[2] / 2, dy = xs[3] / 2, dz = xs[4] / 2;  // dimension = (h, w, l) = (x3, x4, x2): half extents (l, h, w) / 2
        const double sx = c == 8 ? 0.0 : ((c & 4) ? -1.0 : 1.0), sy = c == 8 ? 0.0 : ((c & 2) ? -1.0 : 1.0), sz = c == 8 ? 0.0 : ((c & 1) ? -1.0 : 1.0);
        // corners = (R diag(dx, dy, dz)) signs + location
        const double X = (cs * dx) * sx + (sn * dz) * sz + xs[5];
        const double Y = dy * sy + xs[6];
        const double Z = (-sn * dx) * sx + (cs * dz) * sz + xs[7];
        const double pu = k[0] * X + k[1] * Y + k[2] * Z, pv = k[3] * X + k[4] * Y + k[5] * Z, pw = k[6] * X + k[7] * Y + k[8] * Z;
        u = pu / (pw + 1e-6);
        v = pv / (pw + 1e-6);
    }
    proj[(size_t)t * 2] = u;
    proj[(size_t)t * 2 + 1] = v;
    // bounding rectangle of the eight corners: thread c == 0 of a slot recomputes them (the nine threads of a slot may
    // straddle two waves or blocks; 8 more projections are cheaper than a segmented reduction)
    if (c == 0) {
        double x1 = 0.0, y1 = 0.0, x2 = 0.0, y2 = 0.0;
        if (status[slot] >= 0) {
            const double* xs = x + (size_t)slot * 8;
            const double* k = K + (size_t)(topk > 0 ? slot / topk : slot) * 9;
            const double ry = atan2(xs[0], xs[1]);
            double sn = sin(ry), cs = cos(ry);
            if (fabs(sn) < 1e-3) sn = 0.0;
            if (fabs(cs) < 1e-3) cs = 0.0;
            const double dx = xs[2] / 2, dy = xs[3] / 2, dz = xs[4] / 2;
            x1 = y1 = 1e300; x2 = y2 = -1e300;
            for (int cc = 0; cc < 8; ++cc) {
                const double sx = (cc & 4) ? -1.0 : 1.0, sy = (cc & 2) ? -1.0 : 1.0, sz = (cc & 1) ? -1.0 : 1.0;
                const double X = (cs * dx) * sx + (sn * dz) * sz + xs[5], Y = dy * sy + xs[6], Z = (-sn * dx) * sx + (cs * dz) * sz + xs[7];
                const double pw = k[6] * X + k[7] * Y + k[8] * Z;
                const double uu = (k[0] * X + k[1] * Y + k[2] * Z) / (pw + 1e-6), vv = (k[3] * X + k[4] * Y + k[5] * Z) / (pw + 1e-6);
                x1 = fmin(x1, uu); y1 = fmin(y1, vv); x2 = fmax(x2, uu); y2 = fmax(y2, vv);
            }
        }
        rect[(size_t)slot * 4] = x1; rect[(size_t)slot * 4 + 1] = y1; rect[(size_t)slot * 4 + 2] = x2; rect[(size_t)slot * 4 + 3] = y2;
    }
}

extern "C" int rtm3d_project_boxes(void* stream, int N, int topk, const double* d_x, const int32_t* d_status, const double* d_K,
                                   double* d_proj, double* d_rect) {
    if (N <= 0 || topk < 0) { rt_set_error("project_boxes: bad sizes"); return 1; }
    if (!d_x || !d_status || !d_K || !d_proj || !d_rect) { rt_set_error("project_boxes: null pointer"); return 1; }
    hipLaunchKernelGGL(project_boxes_kernel, dim3((N * 9 + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, topk, d_x, d_status, d_K, d_proj, d_rect);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("project_boxes launch: %s", hipGetErrorString(e)); return 1; }
    return 0;
}
