// 2D decode: sigmoid -> 3x3 equality NMS -> top-k -> threshold -> key-point regression.
// Replaces Model.inference / _obtain_main_proj2d / _obtain_offset_fr_main
// (models/model.py:29-98,117-132) and nms_hm (utils/model_utils.py:17-26).
//
// Three launches:
//   A  (grid over all B*K*H*W elements) sigmoid of the heat-map logits -> workspace (L2 resident)
//   B  (same grid) 3x3 max with -inf border, keep s == max (all members of a plateau survive, like
//      the reference), append survivors with s > thresh as unique 64-bit keys
//      (score bits << 32 | ~flat_index) to the image's candidate list (global atomic counter)
//   C+D (one 1024-thread workgroup per image) exact radix select (8 passes x 8 bits) of the top-k-th
//      key, gather keys >= it, rank them => order: score descending, ties by ascending flat
//      (class-major) index; then one lane per detection: class/y/x from the flat index, 16-channel
//      offset gather, sub-pixel sigmoid, vertices and 2D box in the reference's fp32 operation order.
//
// Bit-exactness: the fp32 sigmoid reproduces what PyTorch-CPU computes: the vectorised path is
// Sleef's expf_u10 polynomial (fma form) followed by an IEEE divide; elements that fall in the
// scalar tail of ATen's vectorised loop (last n % 32 elements of a < 32768-element tensor on an
// AVX-512 host) use a correctly rounded expf instead.  Verified against torch.sigmoid on 16M values.
#include "common.h"
#include "../../include/rtm3d_hip.h"

#pragma clang fp contract(off)

__device__ __forceinline__ float ldexp2kf(float d, int e) {
    const float a = __int_as_float(((e >> 1) + 127) << 23);
    const float b = __int_as_float(((e - (e >> 1)) + 127) << 23);
    return d * a * b;
}

__device__ __forceinline__ float expf_sleef_u10(float d) {
    const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2Uf = 0.693145751953125f, L2Lf = 1.428606765330187045e-06f;
    const int q = (int)rintf(d * R_LN2f);
    float s = __fmaf_rn((float)q, -L2Uf, d);
    s = __fmaf_rn((float)q, -L2Lf, s);
    float u = 0.000198527617612853646278381f;
    u = __fmaf_rn(u, s, 0.00139304355252534151077271f);
    u = __fmaf_rn(u, s, 0.00833336077630519866943359f);
    u = __fmaf_rn(u, s, 0.0416664853692054748535156f);
    u = __fmaf_rn(u, s, 0.166666671633720397949219f);
    u = __fmaf_rn(u, s, 0.5f);
    u = 1.0f + __fmaf_rn(s * s, u, s);
    u = ldexp2kf(u, q);
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = INFINITY;
    return u;
}

// vector == true : ATen Vectorized<float> path;  false : scalar tail (1/(1+std::exp(-x)))
__device__ __forceinline__ float sigmoid_aten(float x, bool vector) {
    float e;
    if (vector) e = expf_sleef_u10(0.0f - x);
    else e = (float)exp((double)(-x));
    return __fdiv_rn(1.0f, 1.0f + e);
}

#define D2_THREADS 1024
#define D2_MAXK 256
#define ATEN_GRAIN 32768
#define ATEN_VSTEP 32

__global__ __launch_bounds__(256) void d2_sigmoid_kernel(const float* __restrict__ main_kf, float* __restrict__ sig, int total,
                                                         int B, int* __restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) cnt[i] = 0;
    if (i >= B * total) return;
    const int local = i % total;
    const int vec_end = (total < ATEN_GRAIN) ? (total / ATEN_VSTEP) * ATEN_VSTEP : total;
    sig[i] = sigmoid_aten(main_kf[i], local < vec_end);
}

__global__ __launch_bounds__(256) void d2_nms_kernel(const float* __restrict__ sig_all, unsigned long long* __restrict__ cand_all,
                                                     int* __restrict__ cnt, int B, int ncls, int H, int W, float thresh) {
    const int HW = H * W, total = ncls * HW;
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= B * total) return;                 // (tail lanes leave: ballots below only see the active lanes)
    const int b = gi / total, i = gi - b * total;
    const int c = i / HW, r = i - c * HW, y = r / W, x = r - y * W;
    const float* pl = sig_all + (size_t)b * total + c * HW;
    const float s = pl[r];
    float mx = s;
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            mx = fmaxf(mx, pl[yy * W + xx]);
        }
    }
    const bool keep = mx == s && s > thresh;
    // Append survivors to the image's candidate list.  One atomic per WAVE, not per survivor: 64 same-address atomics
    // serialise in L2 (with ~3000 peaks per image - a dense heat map - the per-lane form took 4.9 ms per batch).  The
    // wave ballots its survivors, its first survivor reserves the whole run, every survivor takes the slot given by
    // its rank among the lower lanes (v_mbcnt).  A wave that straddles two images falls back to per-lane atomics.
    const unsigned long long key = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    const int b0 = __builtin_amdgcn_readfirstlane(b);
    if (__builtin_amdgcn_ballot_w64(b != b0) == 0ull) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (m != 0ull) {
            const int leader = __builtin_ctzll(m);
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == leader) base = atomicAdd(&cnt[b0], __builtin_popcountll(m));
            base = __builtin_amdgcn_readlane(base, leader);
            if (keep) {
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                cand_all[(size_t)b0 * total + base + rank] = key;
            }
        }
    } else if (keep) {
        const int pos = atomicAdd(&cnt[b], 1);
        cand_all[(size_t)b * total + pos] = key;
    }
}

__global__ __launch_bounds__(D2_THREADS) void decode2d_kernel(
    const float* __restrict__ main_kf, const float* __restrict__ offs, const float* __restrict__ moff,
    int ncls, int H, int W, float thresh, int topk, float down, int mode,
    const int* __restrict__ cnt, unsigned long long* __restrict__ ws_cand,
    int32_t* __restrict__ out_n, int64_t* __restrict__ out_cls, float* __restrict__ out_score,
    float* __restrict__ out_mproj, float* __restrict__ out_verts, float* __restrict__ out_bbox) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int HW = H * W, total = ncls * HW;
    unsigned long long* cand = ws_cand + (size_t)b * total;

    __shared__ int hist[256];
    __shared__ unsigned long long sel[D2_MAXK];
    __shared__ int s_nsel, s_bin, s_cum;

    if (tid == 0) s_nsel = 0;
    __syncthreads();
    const int n = cnt[b];
    // ---- C: exact top-k by radix select on the unique 64-bit keys
    unsigned long long T = 0ull;
    if (n > topk) {
        unsigned long long prefix = 0ull, mask = 0ull;
        int remaining = topk;
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += D2_THREADS) {
                const unsigned long long k = cand[i];
                if ((k & mask) == prefix) atomicAdd(&hist[(int)((k >> shift) & 255ull)], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int cum = 0, bin = 255;
                for (; bin > 0; --bin) {
                    if (cum + hist[bin] >= remaining) break;
                    cum += hist[bin];
                }
                s_bin = bin; s_cum = cum;
            }
            __syncthreads();
            remaining -= s_cum;
            prefix |= (unsigned long long)s_bin << shift;
            mask |= 0xFFull << shift;
            __syncthreads();
        }
        T = prefix;
    }
    for (int i = tid; i < n; i += D2_THREADS) {
        const unsigned long long k = cand[i];
        if (k >= T) {
            const int pos = atomicAdd(&s_nsel, 1);
            if (pos < D2_MAXK) sel[pos] = k;
        }
    }
    __syncthreads();
    const int nsel = min(s_nsel, topk);
    if (tid == 0) out_n[b] = nsel;
    // ---- D: rank + gather (one lane per detection)
    if (tid < nsel) {
        const unsigned long long key = sel[tid];
        int rank = 0;
        for (int jx = 0; jx < nsel; ++jx) rank += (sel[jx] > key) ? 1 : 0;
        const float score = __uint_as_float((unsigned)(key >> 32));
        const int idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        const int c = idx / HW, r = idx - c * HW, y = r / W, x = r - y * W;
        if (mode == 1) {      // peaks only ("smoke" head table): class, score, integer key-point
            const size_t row = (size_t)b * topk + rank;
            out_cls[row] = (int64_t)c;
            out_score[row] = score;
            out_mproj[row * 2 + 0] = (float)x;
            out_mproj[row * 2 + 1] = (float)y;
            return;
        }
        // sub-pixel offset: sigmoid_ over the contiguous (2, N) gather result (models/model.py:48)
        const int n2 = 2 * nsel;
        const int vend = (n2 / ATEN_VSTEP) * ATEN_VSTEP;
        const float* mo = moff + (size_t)b * 2 * HW;
        const float sx = sigmoid_aten(mo[r], rank < vend);
        const float sy = sigmoid_aten(mo[HW + r], (nsel + rank) < vend);
        const float xf = (float)x + sx, yf = (float)y + sy;
        const size_t row = (size_t)b * topk + rank;
        out_cls[row] = (int64_t)c;
        out_score[row] = score;
        out_mproj[row * 2 + 0] = down * xf;
        out_mproj[row * 2 + 1] = down * yf;
        const float* of = offs + (size_t)b * 16 * HW + r;
        float minx = INFINITY, miny = INFINITY, maxx = -INFINITY, maxy = -INFINITY;
        for (int k = 0; k < 8; ++k) {
            const float vx = down * (of[(size_t)(2 * k) * HW] + xf);
            const float vy = down * (of[(size_t)(2 * k + 1) * HW] + yf);
            out_verts[row * 16 + 2 * k] = vx;
            out_verts[row * 16 + 2 * k + 1] = vy;
            minx = fminf(minx, vx); miny = fminf(miny, vy);
            maxx = fmaxf(maxx, vx); maxy = fmaxf(maxy, vy);
        }
        out_bbox[row * 4 + 0] = minx; out_bbox[row * 4 + 1] = miny;
        out_bbox[row * 4 + 2] = maxx; out_bbox[row * 4 + 3] = maxy;
    }
}

extern void rt_set_error(const char* fmt, ...);

extern "C" size_t rtm3d_decode2d_workspace_bytes(int B, int ncls, int H, int W) {
    return (size_t)B * ncls * H * W * (sizeof(float) + sizeof(unsigned long long)) + (size_t)B * sizeof(int) + 256;
}

extern "C" int rtm3d_decode2d(void* stream, const float* d_main_kf, const float* d_offset_fr_main,
                              const float* d_main_offset, int B, int ncls, int H, int W, float score_thresh,
                              int topk, float down_sample, void* d_workspace, int32_t* d_n, int64_t* d_cls,
                              float* d_score, float* d_mproj, float* d_verts, float* d_bbox) {
    if (B <= 0 || ncls <= 0 || H <= 0 || W <= 0) { rt_set_error("decode2d: bad shape"); return 1; }
    if (topk < 1 || topk > D2_MAXK) { rt_set_error("decode2d: topk must be in [1,%d]", D2_MAXK); return 1; }
    const int mode = (d_offset_fr_main == nullptr && d_main_offset == nullptr) ? 1 : 0;     // peaks only
    if (!d_workspace || !d_main_kf || (mode == 0 && (!d_offset_fr_main || !d_main_offset))) { rt_set_error("decode2d: null pointer"); return 1; }
    const size_t total = (size_t)B * ncls * H * W;
    // candidate keys first (8-byte aligned), then the sigmoid plane
    unsigned long long* cand = (unsigned long long*)(((uintptr_t)d_workspace + 7) & ~(uintptr_t)7);
    float* sig = (float*)(cand + total);
    int* cnt = (int*)(sig + total);
    const int per_image = ncls * H * W;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(d2_sigmoid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_main_kf, sig, per_image, B, cnt);
    hipLaunchKernelGGL(d2_nms_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sig, cand, cnt, B, ncls, H, W, score_thresh);
    hipLaunchKernelGGL(decode2d_kernel, dim3(B), dim3(D2_THREADS), 0, (hipStream_t)stream, d_main_kf, d_offset_fr_main,
                       d_main_offset, ncls, H, W, score_thresh, topk, down_sample, mode, cnt, cand, d_n, d_cls, d_score,
                       d_mproj, d_verts, d_bbox);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("decode2d launch: %s", hipGetErrorString(e)); return 1; }
    return 0;
}

// Second half of the 2D decode for the peaks-only regression heads (sparse_heads.hip): rtm3d_decode2d ran in its peaks-only
// mode (class, score, integer key point per slot); the regression logits of every slot - [B*topk][16] offset_fr_main and
// [B*topk][2] main_offset, evaluated at the peak - arrive here.  Same fp32 operation order as decode2d_kernel, incl. the
// position-dependent ATen sigmoid of the contiguous (2, N) gather.
__global__ __launch_bounds__(256) void decode2d_finish_kernel(int B, int topk, const int32_t* __restrict__ n_per_image,
                                                              const float* __restrict__ reg_offs, const float* __restrict__ reg_moff,
                                                              float down, float* __restrict__ mproj, float* __restrict__ verts,
                                                              float* __restrict__ bbox) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * topk) return;
    const int b = i / topk, rank = i - b * topk, nsel = n_per_image[b];
    if (rank >= nsel) return;
    const float x = mproj[(size_t)i * 2], y = mproj[(size_t)i * 2 + 1];          // integer peak, as the peaks-only decode left it
    const int n2 = 2 * nsel;
    const int vend = (n2 / ATEN_VSTEP) * ATEN_VSTEP;
    const float sx = sigmoid_aten(reg_moff[(size_t)i * 2], rank < vend);
    const float sy = sigmoid_aten(reg_moff[(size_t)i * 2 + 1], (nsel + rank) < vend);
    const float xf = x + sx, yf = y + sy;
    mproj[(size_t)i * 2] = down * xf;
    mproj[(size_t)i * 2 + 1] = down * yf;
    const float* of = reg_offs + (size_t)i * 16;
    float minx = INFINITY, miny = INFINITY, maxx = -INFINITY, maxy = -INFINITY;
    for (int k = 0; k < 8; ++k) {
        const float vx = down * (of[2 * k] + xf);
        const float vy = down * (of[2 * k + 1] + yf);
        verts[(size_t)i * 16 + 2 * k] = vx;
        verts[(size_t)i * 16 + 2 * k + 1] = vy;
        minx = fminf(minx, vx); miny = fminf(miny, vy);
        maxx = fmaxf(maxx, vx); maxy = fmaxf(maxy, vy);
    }
    bbox[(size_t)i * 4 + 0] = minx; bbox[(size_t)i * 4 + 1] = miny;
    bbox[(size_t)i * 4 + 2] = maxx; bbox[(size_t)i * 4 + 3] = maxy;
}

extern "C" int rtm3d_decode2d_finish(void* stream, int B, int topk, const int32_t* d_n, const float* d_reg_offset_fr_main,
                                     const float* d_reg_main_offset, float down_sample, float* d_mproj, float* d_verts, float* d_bbox) {
    if (B <= 0 || topk <= 0) { rt_set_error("decode2d_finish: bad sizes"); return 1; }
    if (!d_n || !d_reg_offset_fr_main || !d_reg_main_offset || !d_mproj || !d_verts || !d_bbox) { rt_set_error("decode2d_finish: null pointer"); return 1; }
    hipLaunchKernelGGL(decode2d_finish_kernel, dim3((B * topk + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, topk, d_n,
                       d_reg_offset_fr_main, d_reg_main_offset, down_sample, d_mproj, d_verts, d_bbox);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("decode2d_finish launch: %s", hipGetErrorString(e)); return 1; }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// "smoke" head-table variant.  The branch's source is not part of the
// reference snapshot, so this decode follows the published SMOKE formulation (Liu et al. 2020) and its
// parity is UNPINNED: 8 regression channels [dz, dxc, dyc, dh, dw, dl, sin a, cos a] at the key point,
//   z = 28.01 + 16.32 dz;  (u, v) = down * (x + dxc, y + dyc);  (X, Y, Z) = K^-1 (u z, v z, z)
//   (h, w, l) = dim_ref[cls] * exp(dh, dw, dl)
//   alpha = atan(sin / (cos + 1e-7)) -+ pi/2 (cos >= 0: -, else +);  ry = alpha + atan2(X, Z) wrapped to (-pi, pi]
// One lane per slot; results in the layout of the L-BFGS-B solver: x = [sin ry, cos ry, l, h, w, X, Y, Z].
__global__ __launch_bounds__(256) void smoke_decode_kernel(int B, int topk, const int32_t* __restrict__ n_per_image,
                                                           const int64_t* __restrict__ cls, const float* __restrict__ peak,
                                                           const float* __restrict__ reg, int H, int W, float down,
                                                           const double* __restrict__ K, const double* __restrict__ dim_ref,
                                                           int ncls, double* __restrict__ x_out, double* __restrict__ f_out,
                                                           int32_t* __restrict__ nit, int32_t* __restrict__ status) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * topk) return;
    const int b = i / topk;
    if (i - b * topk >= n_per_image[b]) { status[i] = -1; return; }
    const int px = (int)peak[(size_t)i * 2], py = (int)peak[(size_t)i * 2 + 1];
    const float* rp = reg + (size_t)b * 8 * H * W + (size_t)py * W + px;
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = (double)rp[(size_t)k * H * W];
    int c = (int)cls[i];
    c = c < 0 ? 0 : (c >= ncls ? ncls - 1 : c);
    const double* Kb = K + (size_t)b * 9;
    const double z = 28.01 + 16.32 * r[0];
    const double u = (double)down * ((double)px + r[1]), v = (double)down * ((double)py + r[2]);
    const double X = (u - Kb[2]) * z / Kb[0], Y = (v - Kb[5]) * z / Kb[4];
    const double h = dim_ref[c * 3 + 0] * exp(r[3]), w = dim_ref[c * 3 + 1] * exp(r[4]), l = dim_ref[c * 3 + 2] * exp(r[5]);
    const double PI = 3.14159265358979323846;
    double alpha = atan(r[6] / (r[7] + 1e-7));
    alpha += (r[7] >= 0.0) ? -0.5 * PI : 0.5 * PI;
    double ry = alpha + atan2(X, z);
    if (ry > PI) ry -= 2.0 * PI;
    if (ry < -PI) ry += 2.0 * PI;
    double* xo = x_out + (size_t)i * 8;
    xo[0] = sin(ry); xo[1] = cos(ry); xo[2] = l; xo[3] = h; xo[4] = w; xo[5] = X; xo[6] = Y; xo[7] = z;
    f_out[i] = 0.0; nit[i] = 0; status[i] = 0;
}

extern "C" int rtm3d_decode_smoke(void* stream, int B, int topk, const int32_t* d_n, const int64_t* d_cls, const float* d_peak_xy,
                                  const float* d_reg, int H, int W, float down_sample, const double* d_K_per_image,
                                  const double* d_dim_ref, int ncls, double* d_x, double* d_fun, int32_t* d_nit, int32_t* d_status) {
    if (B <= 0 || topk <= 0 || H <= 0 || W <= 0 || ncls <= 0) { rt_set_error("decode_smoke: bad sizes"); return 1; }
    if (!d_n || !d_cls || !d_peak_xy || !d_reg || !d_K_per_image || !d_dim_ref || !d_x || !d_fun || !d_nit || !d_status) {
        rt_set_error("decode_smoke: null pointer"); return 1;
    }
    hipLaunchKernelGGL(smoke_decode_kernel, dim3((B * topk + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, topk, d_n, d_cls,
                       d_peak_xy, d_reg, H, W, down_sample, d_K_per_image, d_dim_ref, ncls, d_x, d_fun, d_nit, d_status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("decode_smoke launch: %s", hipGetErrorString(e)); return 1; }
    return 0;
}
