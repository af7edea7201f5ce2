// HBM-bound element-wise / reduction kernels of the neck and backbones (padded NHWC fp16).
//   maxpool        : models/nets/dla.py:170-172,190-193 (2x2 s2) and models/nets/resnet.py:128 (3x3 s2 p1)
//   softmax fusion : models/nets/keypoint_fpn_fusion.py:60-69
//                    z += u * softmax(u over H*W) per (image, channel), for up to three u.
// All accesses are 8- or 16-byte vectors along the contiguous channel axis.
#include "common.h"

__global__ __launch_bounds__(256) void maxpool_kernel(const PoolKArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = a.B * a.Ho * a.Wo * a.C8;
    if (idx >= total) return;
    const int c8 = idx % a.C8;
    int p = idx / a.C8;
    const int ox = p % a.Wo; p /= a.Wo;
    const int oy = p % a.Ho;
    const int n = p / a.Ho;
    f16x8 m;
    bool first = true;
    for (int ky = 0; ky < a.ksize; ++ky)
        for (int kx = 0; kx < a.ksize; ++kx) {
            const int iy = oy * a.stride - a.pad + ky + a.in_P, ix = ox * a.stride - a.pad + kx + a.in_P;
            const f16x8 v = *(const f16x8*)(a.in + ((size_t)(n * a.in_Hp + iy) * a.in_Wp + ix) * a.in_C + a.in_coff + c8 * 8);
            if (first) { m = v; first = false; }
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
    *(f16x8*)(a.out + ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + a.out_coff + c8 * 8) = m;
}

hipError_t launch_maxpool(const PoolKArgs& a, hipStream_t s) {
    const int total = a.B * a.Ho * a.Wo * a.C8;
    hipLaunchKernelGGL(maxpool_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------- spatial softmax fusion
// pass 1: per (u, image, row-chunk) partial (max, sum exp) for each of the 256 channels.
__global__ __launch_bounds__(256) void softmax_reduce_kernel(const SoftmaxKArgs a) {
    const int chunk = blockIdx.x, n = blockIdx.y, ui = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f16* u = a.u[ui];
    const int Hp = a.u_Hp[ui], Wp = a.u_Wp[ui], C = a.u_C[ui], P = a.u_P[ui];
    float m[4], s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { m[e] = -INFINITY; s[e] = 0.f; }
    const int y0 = chunk * a.rows_per_chunk;
    const int y1 = min(y0 + a.rows_per_chunk, a.H);
    for (int y = y0; y < y1; ++y) {
        const f16* row = u + ((size_t)(n * Hp + y + P) * Wp + P) * C + lane * 4;
        for (int x = wave; x < a.W; x += 4) {
            const f16x4 v = *(const f16x4*)(row + (size_t)x * C);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = (float)v[e];
                const float mn = fmaxf(m[e], f);
                s[e] = s[e] * __expf(m[e] - mn) + __expf(f - mn);
                m[e] = mn;
            }
        }
    }
    __shared__ float sm[4][256], ss[4][256];
#pragma unroll
    for (int e = 0; e < 4; ++e) { sm[wave][lane * 4 + e] = m[e]; ss[wave][lane * 4 + e] = s[e]; }
    __syncthreads();
    const int c = threadIdx.x;   // 256 threads <-> 256 channels
    float M = fmaxf(fmaxf(sm[0][c], sm[1][c]), fmaxf(sm[2][c], sm[3][c]));
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) S += (sm[w][c] == -INFINITY) ? 0.f : ss[w][c] * __expf(sm[w][c] - M);
    float* out = a.partial + ((((size_t)ui * a.B + n) * a.chunks + chunk) * a.C + c) * 2;
    out[0] = M; out[1] = S;
}

// pass 2: combine the per-chunk partials once per (u, image, channel): stats[u][b][c] = (max, 1/sum)
__global__ __launch_bounds__(256) void softmax_combine_kernel(const SoftmaxKArgs a) {
    const int n = blockIdx.x, ui = blockIdx.y, c = threadIdx.x;
    const float* p = a.partial + (((size_t)ui * a.B + n) * a.chunks * a.C + c) * 2;
    float mm = -INFINITY;
    for (int k = 0; k < a.chunks; ++k) mm = fmaxf(mm, p[(size_t)k * a.C * 2]);
    float ssum = 0.f;
    for (int k = 0; k < a.chunks; ++k) {
        const float mk = p[(size_t)k * a.C * 2];
        if (mk != -INFINITY) ssum += p[(size_t)k * a.C * 2 + 1] * __expf(mk - mm);
    }
    float* o = a.stats + (((size_t)ui * a.B + n) * a.C + c) * 2;
    o[0] = mm; o[1] = 1.f / ssum;
}

// pass 3: z_out = z_in + sum_i u_i * exp(u_i - M_i) / S_i
__global__ __launch_bounds__(256) void softmax_apply_kernel(const SoftmaxKArgs a) {
    const int chunk = blockIdx.x, n = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float M[3][4], invS[3][4];
    for (int ui = 0; ui < a.n_u; ++ui) {
        const float* st = a.stats + (((size_t)ui * a.B + n) * a.C + lane * 4) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { M[ui][e] = st[2 * e]; invS[ui][e] = st[2 * e + 1]; }
    }
    const int y0 = chunk * a.rows_per_chunk;
    const int y1 = min(y0 + a.rows_per_chunk, a.H);
    for (int y = y0; y < y1; ++y) {
        const size_t zrow = ((size_t)(n * a.z_Hp + y + a.z_P) * a.z_Wp + a.z_P) * a.z_C + lane * 4;
        const size_t zirow = ((size_t)(n * a.zi_Hp + y + a.zi_P) * a.zi_Wp + a.zi_P) * a.zi_C + lane * 4;
        for (int x = wave; x < a.W; x += 4) {
            const f16x4 zi = *(const f16x4*)(a.z_in + zirow + (size_t)x * a.zi_C);
            float acc[4] = {(float)zi[0], (float)zi[1], (float)zi[2], (float)zi[3]};
            for (int ui = 0; ui < a.n_u; ++ui) {
                const f16x4 v = *(const f16x4*)(a.u[ui] + ((size_t)(n * a.u_Hp[ui] + y + a.u_P[ui]) * a.u_Wp[ui] + a.u_P[ui] + x) * a.u_C[ui] + lane * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f = (float)v[e];
                    acc[e] += f * (__expf(f - M[ui][e]) * invS[ui][e]);
                }
            }
            f16x4 o = {(f16)acc[0], (f16)acc[1], (f16)acc[2], (f16)acc[3]};
            *(f16x4*)(a.z_out + zrow + (size_t)x * a.z_C) = o;
        }
    }
}

hipError_t launch_softmax_fuse(const SoftmaxKArgs& a, hipStream_t s) {
    if (a.C != 256 || a.n_u < 1 || a.n_u > 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(softmax_reduce_kernel, dim3(a.chunks, a.B, a.n_u), dim3(256), 0, s, a);
    hipLaunchKernelGGL(softmax_combine_kernel, dim3(a.B, a.n_u, 1), dim3(256), 0, s, a);
    hipLaunchKernelGGL(softmax_apply_kernel, dim3(a.chunks, a.B, 1), dim3(256), 0, s, a);
    return hipGetLastError();
}
