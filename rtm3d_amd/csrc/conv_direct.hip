// Direct (VALU) convolutions for the 16/32-channel layers at full / half resolution (DLA level0, level1,
// level2 entry).  Kept as kernel variant 1: the register-direct MFMA kernels of conv_smallc.hip are the
// default for these layers; this form is the cross-check and the fallback for residual epilogues.
// These layers are HBM-bound (72-124 FLOP/B), so the design goal is
// coalesced NHWC traffic, not MFMA: one thread owns one output pixel and all COUT accumulators,
// weights are wave-uniform and come through the scalar cache (s_load), the inner product uses
// v_dot2_f32_f16 (two fp16 MACs, fp32 accumulate).
//
// Replaces models/nets/dla.py:259-279 (base_layer, level0, level1), the first convolution and the
// 1x1 projection of level2 (dla.py:56-100,175-184) and models/nets/resnet.py:124-126 (conv1).
#include "common.h"

// weights: uint32 (= two fp16: cin 2p, 2p+1) laid out [tap][cin/2][cout]
template <int CIN, int COUT, int NTAPS>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvKArgs a) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= a.M) return;
    const ConvGroupArgs& g = a.g[0];
    const int n = m / a.HmWm, rem = m - n * a.HmWm;
    const int y = rem / a.Wm, x = rem - y * a.Wm;
    const size_t pix = (size_t)(n * a.in_Hp + y * a.in_stride + a.in_P) * a.in_Wp + x * a.in_stride + a.in_P;
    const f16* ip = a.in + pix * a.in_C + g.in_coff;
    const uint32_t* __restrict__ w = (const uint32_t*)(a.wgt + g.w_off);

    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.f;

#pragma unroll 1
    for (int t = 0; t < NTAPS; ++t) {
        const f16* tp = ip + g.tap_off[t];
        f16x8 xv[CIN / 8];
#pragma unroll
        for (int i = 0; i < CIN / 8; ++i) xv[i] = *(const f16x8*)(tp + i * 8);
        const uint32_t* wt = w + t * (CIN / 2) * COUT;
#pragma unroll
        for (int p = 0; p < CIN / 2; ++p) {
            const f16x2 xp = {xv[p / 4][(p % 4) * 2], xv[p / 4][(p % 4) * 2 + 1]};
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                const uint32_t wb = wt[p * COUT + c];
                f16x2 wp;
                __builtin_memcpy(&wp, &wb, 4);
                acc[c] = __builtin_amdgcn_fdot2(xp, wp, acc[c], false);
            }
        }
    }

    const int oy = y * a.out_scale + g.out_oy, ox = x * a.out_scale + g.out_ox;
    const size_t oo = ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + g.out_coff;
    const f16* rp = nullptr;
    if (a.res) rp = a.res + ((size_t)(n * a.res_Hp + oy + a.res_P) * a.res_Wp + ox + a.res_P) * a.res_C + g.res_coff;
    const float* bp = a.bias + g.bias_off;
#pragma unroll
    for (int c8 = 0; c8 < COUT / 8; ++c8) {
        f16x8 r8;
        if (rp) r8 = *(const f16x8*)(rp + c8 * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[c8 * 8 + e] + bp[c8 * 8 + e];
            if (rp) v += (float)r8[e];
            if (a.relu) v = fmaxf(v, 0.f);
            o[e] = (f16)v;
        }
        *(f16x8*)((f16*)a.out + oo + c8 * 8) = o;
    }
}

bool conv_direct_supported(int cin, int cout, int ntaps) {
    return (cin == 16 && cout == 16 && ntaps == 9) || (cin == 16 && cout == 32 && ntaps == 9) ||
           (cin == 32 && cout == 64 && ntaps == 9) || (cin == 32 && cout == 64 && ntaps == 1);
}

hipError_t launch_conv_direct(const ConvKArgs& a, int ksize, int groups, hipStream_t s) {
    (void)ksize;
    if (groups != 1) return hipErrorInvalidValue;
    dim3 grid((a.M + 255) / 256), block(256);
    if (a.cin == 16 && a.cout == 16 && a.ntaps == 9)
        hipLaunchKernelGGL((conv_direct_kernel<16, 16, 9>), grid, block, 0, s, a);
    else if (a.cin == 16 && a.cout == 32 && a.ntaps == 9)
        hipLaunchKernelGGL((conv_direct_kernel<16, 32, 9>), grid, block, 0, s, a);
    else if (a.cin == 32 && a.cout == 64 && a.ntaps == 9)
        hipLaunchKernelGGL((conv_direct_kernel<32, 64, 9>), grid, block, 0, s, a);
    else if (a.cin == 32 && a.cout == 64 && a.ntaps == 1)
        hipLaunchKernelGGL((conv_direct_kernel<32, 64, 1>), grid, block, 0, s, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
