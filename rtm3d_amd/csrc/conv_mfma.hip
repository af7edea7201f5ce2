// Implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_16x16x32_f16).
//
// GEMM view:  D[cout][pixel] = sum_k  W[cout][k] * X[pixel][k],   k = (tap, 64-channel chunk)
//   - activations are padded NHWC fp16, so a tap is a constant element offset from the pixel's
//     base address and the zero border implements the convolution padding: no bounds checks;
//   - one k-step stages a BM x 64 pixel tile and a BN x 64 weight tile (128-byte rows) into LDS
//     with global_load_lds_dwordx4 (16 B/lane, direct to LDS).  The LDS image is lane-linear, so
//     the bank swizzle (16-B chunk ^= row & 7) is applied on the per-lane SOURCE address for
//     pixels and baked into the host-side weight packing; reads apply the same XOR -> conflict
//     free ds_read_b128 for the 16x16x32 operand map (lane -> row = lane & 15, k-chunk = lane >> 4);
//   - weights are the MFMA "A" operand, pixels the "B" operand, so each lane ends up with four
//     consecutive output channels of one pixel: 8-byte NHWC stores, fused bias/residual/ReLU.
//   - blockIdx is remapped so the NT channel tiles of one pixel tile run on one XCD (shared L2).
//
// Replaces the stock ATen conv2d / conv_transpose2d calls of models/nets/{dla,resnet,header,
// keypoint_fpn_fusion,module}.py.
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

template <int BM, int BN, int WGM, int WGN, int EPI>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvKArgs a) {
    constexpr int XP = BM * 8;                 // 16-byte pieces of the pixel tile per k-step
    constexpr int WP = BN * 8;                 // pieces of the weight tile
    constexpr int XI = XP / 256;
    constexpr int WI = (WP + 255) / 256;
    constexpr int TP = BM / WGM / 16;          // 16-pixel sub-tiles per wave
    constexpr int TC = BN / WGN / 16;          // 16-channel sub-tiles per wave
    constexpr int STAGE = (BM + BN) * 64;      // halves per LDS stage
    __shared__ __attribute__((aligned(16))) f16 lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave / WGN, wc = wave % WGN;

    // XCD-aware tile mapping: blocks b, b+8, b+16.. share an XCD -> give them the same pixel tile.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int chunk = (a.MT + 7) >> 3;           // each XCD walks a contiguous run of pixel tiles (shared halo rows)
    const int ntile = j % a.NT;
    const int mtile = xcd * chunk + j / a.NT;
    if (mtile >= a.MT) return;
    const ConvGroupArgs& g = a.g[blockIdx.y];

    // per-lane source offsets of the pixel rows this lane stages (fixed for the whole K loop)
    uint32_t xoff[XI];
    {
        const int rr = tid >> 3, cs = tid & 7;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int m = mtile * BM + i * 32 + rr;
            m = m < a.M ? m : a.M - 1;
            const int n = m / a.HmWm, rem = m - n * a.HmWm;
            const int y = rem / a.Wm, x = rem - y * a.Wm;
            const uint32_t pix = (uint32_t)((n * a.in_Hp + y * a.in_stride + a.in_P) * a.in_Wp + x * a.in_stride + a.in_P);
            xoff[i] = pix * (uint32_t)a.in_C + (uint32_t)g.in_coff + (uint32_t)((cs ^ (rr & 7)) * 8);
        }
    }
    const f16* wbase = a.wgt + g.w_off + (size_t)ntile * a.ksteps * (BN * 64);

    auto stage = [&](int buf, int ks) {
        const int tap = ks / a.cpt, q = ks - tap * a.cpt;
        const int koff = g.tap_off[tap] + q * 64;
        f16* xl = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const f16* src = a.in + (size_t)xoff[i] + (ptrdiff_t)koff;
            __builtin_amdgcn_global_load_lds((const GLB_AS void*)src, (LDS_AS void*)(xl + (i * 256 + wave * 64) * 8), 16, 0, 0);
        }
        const f16* ws = wbase + (size_t)ks * (BN * 64);
        f16* wl = xl + BM * 64;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            if (WP >= 256 || wave * 64 < WP) {   // wave-uniform guard for the 16-channel tile
                __builtin_amdgcn_global_load_lds((const GLB_AS void*)(ws + (i * 256 + tid) * 8),
                                                 (LDS_AS void*)(wl + (i * 256 + wave * 64) * 8), 16, 0, 0);
            }
        }
    };

    f32x4 acc[TC][TP];
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
        for (int p = 0; p < TP; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = lane >> 4;
    stage(0, 0);
    __syncthreads();
    for (int ks = 0; ks < a.ksteps; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < a.ksteps) stage(cur ^ 1, ks + 1);
        const f16* xl = lds + cur * STAGE;
        const f16* wl = xl + BM * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = ((kk * 4 + fk) ^ (frow & 7)) * 8;
            f16x8 xf[TP], wf[TC];
#pragma unroll
            for (int p = 0; p < TP; ++p)
                xf[p] = *(const f16x8*)(xl + (wp * (BM / WGM) + p * 16 + frow) * 64 + sw);
#pragma unroll
            for (int c = 0; c < TC; ++c)
                wf[c] = *(const f16x8*)(wl + (wc * (BN / WGN) + c * 16 + frow) * 64 + sw);
#pragma unroll
            for (int c = 0; c < TC; ++c)
#pragma unroll
                for (int p = 0; p < TP; ++p)
                    acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], xf[p], acc[c][p], 0, 0, 0);
        }
        __syncthreads();   // drains the in-flight global_load_lds (vmcnt(0)) and orders LDS reuse
    }

    // ---------------------------------------------------------------- epilogue
#pragma unroll
    for (int p = 0; p < TP; ++p) {
        const int m = mtile * BM + wp * (BM / WGM) + p * 16 + frow;
        if (m >= a.M) continue;
        const int n = m / a.HmWm, rem = m - n * a.HmWm;
        const int y = rem / a.Wm, x = rem - y * a.Wm;
        const int oy = y * a.out_scale + g.out_oy, ox = x * a.out_scale + g.out_ox;
#pragma unroll
        for (int c = 0; c < TC; ++c) {
            const int c0 = ntile * BN + wc * (BN / WGN) + c * 16 + fk * 4;
            if (c0 >= a.cout) continue;
            f32x4 v = acc[c][p];
            const float* bp = a.bias + g.bias_off + c0;
            if (EPI == 0) {
                const f32x4 b = *(const f32x4*)bp;
                v += b;
                if (a.res) {
                    const size_t ro = ((size_t)(n * a.res_Hp + oy + a.res_P) * a.res_Wp + ox + a.res_P) * a.res_C + g.res_coff + c0;
                    const f16x4 r = *(const f16x4*)(a.res + ro);
                    v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
                }
                if (a.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                const size_t oo = ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + g.out_coff + c0;
                f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *(f16x4*)((f16*)a.out + oo) = h;
            } else {
                float* o = (float*)a.out;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cc = c0 + e;
                    if (cc < a.cout) {
                        float r = v[e] + bp[e];
                        if (a.relu) r = fmaxf(r, 0.f);
                        o[((size_t)(n * a.out_C + g.out_coff + cc) * a.out_H + oy) * a.out_W + ox] = r;
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN>
static hipError_t launch_t(const ConvKArgs& a, int groups, int epi, hipStream_t s) {
    const int mt8 = (a.MT + 7) / 8 * 8;
    dim3 grid(mt8 * a.NT, groups, 1), block(256, 1, 1);
    if (epi)
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WGM, WGN, 1>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WGM, WGN, 0>), grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_conv_mfma(const ConvKArgs& a, int bn_tile, int groups, int epi_nchw, hipStream_t s) {
    switch (bn_tile) {
        case 128: return launch_t<128, 128, 2, 2>(a, groups, epi_nchw, s);
        case 64: return launch_t<128, 64, 2, 2>(a, groups, epi_nchw, s);
        case 32: return launch_t<128, 32, 4, 1>(a, groups, epi_nchw, s);
        case 16: return launch_t<128, 16, 4, 1>(a, groups, epi_nchw, s);
        default: return hipErrorInvalidValue;
    }
}
