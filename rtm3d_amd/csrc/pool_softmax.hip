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
    if (a.ksize == 0) {
        // space-to-depth input (rtm3d_op_maxpool_s2d): the 2 x 2 window of full-resolution pixels = the four channel slices of this pixel
        const f16* ip = a.in + ((size_t)(n * a.in_Hp + oy + a.in_P) * a.in_Wp + ox + a.in_P) * a.in_C + a.in_coff + c8 * 8;
        m = *(const f16x8*)ip;
#pragma unroll
        for (int ph = 1; ph < 4; ++ph) {
            const f16x8 v = *(const f16x8*)(ip + ph * a.C8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
        }
        first = false;
    }
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
// A lane owns 8 channels (16-byte loads), a wave instruction covers two pixels, four pixels are in
// flight per lane; the running sum is rescaled once per batch of four values, not per value.
__global__ __launch_bounds__(256) void softmax_reduce_kernel(const SoftmaxKArgs a) {
    const int chunk = blockIdx.x, n = blockIdx.y, ui = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 31, slot = wave * 2 + (lane >> 5);        // 8 pixel slots per workgroup
    const f16* u = a.u[ui];
    const int Hp = a.u_Hp[ui], Wp = a.u_Wp[ui], C = a.u_C[ui], P = a.u_P[ui];
    float m[8], s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; s[e] = 0.f; }
    const int y0 = chunk * a.rows_per_chunk;
    const int y1 = min(y0 + a.rows_per_chunk, a.H);
    for (int y = y0; y < y1; ++y) {
        const f16* row = u + ((size_t)(n * Hp + y + P) * Wp + P) * C + cg * 8;
        for (int x = slot; x < a.W; x += 32) {
            f16x8 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = x + 8 * k < a.W ? x + 8 * k : x;          // tail: re-read a pixel already counted ...
                v[k] = *(const f16x8*)(row + (size_t)xx * C);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) f[k] = (x + 8 * k < a.W) ? (float)v[k][e] : -INFINITY;   // ... with weight exp(-inf) = 0
                const float mn = fmaxf(fmaxf(fmaxf(f[0], f[1]), fmaxf(f[2], f[3])), m[e]);
                s[e] = s[e] * __expf(m[e] - mn) + ((__expf(f[0] - mn) + __expf(f[1] - mn)) + (__expf(f[2] - mn) + __expf(f[3] - mn)));
                m[e] = mn;
            }
        }
    }
    __shared__ float sm[8][256], ss[8][256];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[slot][cg * 8 + e] = m[e]; ss[slot][cg * 8 + e] = s[e]; }
    __syncthreads();
    const int c = threadIdx.x;   // 256 threads <-> 256 channels
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 8; ++w) M = fmaxf(M, sm[w][c]);
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) S += (sm[w][c] == -INFINITY) ? 0.f : ss[w][c] * __expf(sm[w][c] - M);
    float* out = a.partial + ((((size_t)ui * a.B + n) * a.chunks + chunk) * a.C + c) * 2;
    out[0] = M; out[1] = S;
}

// pass 2: combine the per-chunk partials once per (u, image, channel): stats[u][b][c] = (max, 1/sum)
// Round 6: one workgroup per (image, u, QUARTER of the channels), 64 channels x 16 parts: thread (part, c) folds the chunks
// k = part (mod 16) of channel c, LDS folds the sixteen.  The former form - one workgroup per (image, u), 256 channels x 4 parts -
// put 96 workgroups on the chip for the bs=32 fusion and walked 60 dependent (load, two exponentials) steps per thread: 49 us in the
// kernel trace for 47 MB of partials; now 384 workgroups and 15 steps.
__global__ __launch_bounds__(1024) void softmax_combine_kernel(const SoftmaxKArgs a) {
    const int n = blockIdx.x, ui = blockIdx.y, cl = threadIdx.x & 63, c = blockIdx.z * 64 + cl, part = threadIdx.x >> 6;
    const int chunks = a.partial_chunks > 0 ? a.partial_chunks : a.chunks;
    const float2* p = (const float2*)a.partial + ((size_t)ui * a.B + n) * chunks * a.C + c;
    float m = -INFINITY, sum = 0.f;
    for (int k = part; k < chunks; k += 16) {
        const float2 v = p[(size_t)k * a.C];
        if (v.x != -INFINITY) {
            const float mn = fmaxf(m, v.x);
            sum = sum * __expf(m - mn) + v.y * __expf(v.x - mn);
            m = mn;
        }
    }
    __shared__ float sm[16][64], ss[16][64];
    sm[part][cl] = m; ss[part][cl] = sum;
    __syncthreads();
    if (part == 0) {
        float mm = sm[0][cl];
#pragma unroll
        for (int w = 1; w < 16; ++w) mm = fmaxf(mm, sm[w][cl]);
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += (sm[w][cl] == -INFINITY) ? 0.f : ss[w][cl] * __expf(sm[w][cl] - mm);
        float* o = a.stats + (((size_t)ui * a.B + n) * a.C + c) * 2;
        o[0] = mm; o[1] = 1.f / t;
    }
}

// pass 3: z_out = z_in + sum_i u_i * exp(u_i - M_i) / S_i      (16-byte lanes, two pixels in flight per lane)
// Round 4, same-box timing builds: without the exponentials 0.496 vs 0.497 ms (the pass is memory-bound: 2.0 GB read + 0.5 GB written at 5.2-5.4 TB/s,
// four read streams and one write stream); one pixel per lane and step with the next step's loads issued before the arithmetic
// 0.539 vs 0.528; 8-byte lanes (4 channels per lane, 124 registers, four waves per SIMD) 0.483 vs 0.479; smaller workgroup shares help
// a little (runtime.hip: one row cut into four segments per workgroup, -4 %).  A plain 4-reads-1-write kernel on 512 MB streams gets
// 4.65-5.3 TB/s on these boxes (profiles/r04_hbm_rates.txt).
template <int NU>
__global__ __launch_bounds__(256) void softmax_apply_kernel(const SoftmaxKArgs a) {
    const int chunk = blockIdx.x, n = blockIdx.y;
    const int rchunk = chunk / a.xsplit, xseg = chunk - rchunk * a.xsplit;
    const int x0 = xseg * a.seg_w, x1 = min(x0 + a.seg_w, a.W);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 31, slot = wave * 2 + (lane >> 5);
    float M[NU][8], invS[NU][8];
#pragma unroll
    for (int ui = 0; ui < NU; ++ui) {
        const float* st = a.stats + (((size_t)ui * a.B + n) * a.C + cg * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) { M[ui][e] = st[2 * e]; invS[ui][e] = st[2 * e + 1]; }
    }
    const int y0 = rchunk * a.apply_rows;
    const int y1 = min(y0 + a.apply_rows, a.H);
    for (int y = y0; y < y1; ++y) {
        const size_t zrow = ((size_t)(n * a.z_Hp + y + a.z_P) * a.z_Wp + a.z_P) * a.z_C + cg * 8;
        const size_t zirow = ((size_t)(n * a.zi_Hp + y + a.zi_P) * a.zi_Wp + a.zi_P) * a.zi_C + cg * 8;
        size_t urow[NU];
#pragma unroll
        for (int ui = 0; ui < NU; ++ui) urow[ui] = ((size_t)(n * a.u_Hp[ui] + y + a.u_P[ui]) * a.u_Wp[ui] + a.u_P[ui]) * a.u_C[ui] + cg * 8;
        for (int x = x0 + slot; x < x1; x += 16) {
            f16x8 zi[2], v[2][NU];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int xx = x + 8 * k < x1 ? x + 8 * k : x;
                // non-temporal: the 2 GB of z_in and u are read exactly once (same-box A/B, profiles/r03_ab_nt.txt: 0.550 -> 0.509 ms)
                zi[k] = __builtin_nontemporal_load((const f16x8*)(a.z_in + zirow + (size_t)xx * a.zi_C));
#pragma unroll
                for (int ui = 0; ui < NU; ++ui) v[k][ui] = __builtin_nontemporal_load((const f16x8*)(a.u[ui] + urow[ui] + (size_t)xx * a.u_C[ui]));
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (x + 8 * k >= x1) break;
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float acc = (float)zi[k][e];
#pragma unroll
                    for (int ui = 0; ui < NU; ++ui) {
                        const float f = (float)v[k][ui][e];
                        acc += f * (__expf(f - M[ui][e]) * invS[ui][e]);
                    }
                    o[e] = (f16)acc;
                }
                // (a non-temporal STORE here was measured too: the head conv that reads z next went 3.72 -> 4.06 ms)
                *(f16x8*)(a.z_out + zrow + (size_t)(x + 8 * k) * a.z_C) = o;
            }
        }
    }
}

hipError_t launch_softmax_fuse(const SoftmaxKArgs& a, hipStream_t s) {
    if (a.C != 256 || a.n_u < 1 || a.n_u > 3 || a.xsplit < 1 || a.seg_w < 1 || a.apply_rows < 1) return hipErrorInvalidValue;
    // pass 1 is skipped when the producing convolutions emitted the partials from their epilogues
    if (a.partial_chunks <= 0) hipLaunchKernelGGL(softmax_reduce_kernel, dim3(a.chunks, a.B, a.n_u), dim3(256), 0, s, a);
    hipLaunchKernelGGL(softmax_combine_kernel, dim3(a.B, a.n_u, a.C / 64), dim3(1024), 0, s, a);
    if (a.n_u == 3) hipLaunchKernelGGL(softmax_apply_kernel<3>, dim3(a.apply_chunks, a.B, 1), dim3(256), 0, s, a);
    else if (a.n_u == 2) hipLaunchKernelGGL(softmax_apply_kernel<2>, dim3(a.apply_chunks, a.B, 1), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(softmax_apply_kernel<1>, dim3(a.apply_chunks, a.B, 1), dim3(256), 0, s, a);
    return hipGetLastError();
}
