// DLA-34 stem, fused: base_layer (7x7, 3 -> 16, BN, ReLU) + level0 (3x3, 16 -> 16, BN, ReLU) in one kernel
// (models/nets/dla.py:259-273).  Both layers are HBM-bound on their own (0.22 + 0.25 ms at bs=32: the 16-channel
// full-resolution map is written once and read back once, 1.0 GB of the 1.6 GB they move); fused, the intermediate map
// lives in LDS only and the pair moves 0.13 GB in + 0.5 GB out.
//
// One 256-thread workgroup per 16 x 32 output tile (several per CU: 28 KB of LDS, few registers - the latency of the
// loads is covered by occupancy, no DMA ring needed):
//   1. the (16+8) x (32+10) window of the NHWC4 fp16 image (8 B per pixel) -> LDS;
//   2. base_layer on the (16+2) x (32+2) halo the 3x3 needs: register-direct MFMA, K-step = one filter row
//      (7 taps x 4 channels + one zero tap = 32: lane group fk holds the pixels x+2fk, x+2fk+1, which are adjacent in
//      the NHWC4 window -> one 16-byte operand), bias + ReLU, fp16, -> LDS as [row][column][16 channels];
//      pixels outside the image are written as ZERO: they are level0's zero padding, not a convolution of padded input;
//   3. level0 on the 16 x 32 tile from that LDS map: K-step = two taps x 16 channels, bias + ReLU -> global NHWC.
// Weight fragments (7 + 5 A operands) stay in registers.  Same packing as conv_smallc.hip (cin = 4 / cin = 16 layouts).
#include "common.h"
#include "../../include/rtm3d_hip.h"

#define SF_TH 16
#define SF_TW 32
#define SF_BH (SF_TH + 2)            // base rows needed
#define SF_BW (SF_TW + 2)
#define SF_XH (SF_TH + 8)            // image rows needed (7x7 around every base pixel)
#define SF_XW (SF_TW + 10)           // + 2: the zero-weighted 8th tap of a filter row still loads a pixel
#define SF_BPIX (SF_BH * SF_BW)      // 612
#define SF_BFRAGS ((SF_BPIX + 15) / 16)

__global__ __launch_bounds__(256) void stem_fused_kernel(const StemFusedArgs a) {
    __shared__ __attribute__((aligned(16))) f16 xt[SF_XH * SF_XW * 4];        // image window, 4 halves per pixel
    __shared__ __attribute__((aligned(16))) f16 bt[SF_BPIX * 16 + 16 * 16];   // base map (+ slack for the clamped last fragment)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fk = lane >> 4;
    const int tpi = a.tiles_x * a.tiles_y;
    const int n = blockIdx.x / tpi, r0 = blockIdx.x - n * tpi;
    const int ty = r0 / a.tiles_x, tx = r0 - ty * a.tiles_x;
    const int y0 = ty * SF_TH, x0 = tx * SF_TW;

    // ---- 1. image window: rows y0-4 .. y0+19, columns x0-4 .. x0+37 of the padded NHWC4 tensor (8-byte pixels)
    {
        const f16* src = a.x4 + ((size_t)(n * a.x_Hp + y0 - 4 + a.x_P) * a.x_Wp + (x0 - 4 + a.x_P)) * 4;
        for (int p = tid; p < SF_XH * SF_XW; p += 256) {
            const int r = p / SF_XW, c = p - r * SF_XW;
            *(f16x4*)(xt + p * 4) = *(const f16x4*)(src + ((size_t)r * a.x_Wp + c) * 4);
        }
    }
    f16x8 wb[7], wl[5];
#pragma unroll
    for (int s = 0; s < 7; ++s) wb[s] = *(const f16x8*)(a.w_base + (s * 64 + lane) * 8);
#pragma unroll
    for (int s = 0; s < 5; ++s) wl[s] = *(const f16x8*)(a.w_l0 + (s * 64 + lane) * 8);
    const f32x4 bb = *(const f32x4*)(a.b_base + fk * 4), bl = *(const f32x4*)(a.b_l0 + fk * 4);
    __syncthreads();

    // ---- 2. base_layer on the 18 x 34 halo (612 pixels = 39 fragments of 16, dealt round-robin to the 4 waves)
    for (int f = wave; f < SF_BFRAGS; f += 4) {
        int q = f * 16 + frow;
        q = q < SF_BPIX ? q : SF_BPIX - 1;                  // the last fragment's spare lanes redo pixel 611
        const int r = q / SF_BW, c = q - r * SF_BW;
        // operand of filter row ky: pixels (r + ky, c + 2fk), (r + ky, c + 2fk + 1) of the window, 4 channels each
        const f16* xp = xt + (r * SF_XW + c + 2 * fk) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const f16x4 lo = *(const f16x4*)(xp + ky * SF_XW * 4), hi = *(const f16x4*)(xp + ky * SF_XW * 4 + 4);
            const f16x8 xf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ky], xf, acc, 0, 0, 0);
        }
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        f16x4 h = {(f16)fmaxf(acc[0] + bb[0], 0.f), (f16)fmaxf(acc[1] + bb[1], 0.f), (f16)fmaxf(acc[2] + bb[2], 0.f),
                   (f16)fmaxf(acc[3] + bb[3], 0.f)};
        if (!inside) h = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        *(f16x4*)(bt + (f * 16 + frow) * 16 + fk * 4) = h;          // slot f*16+frow (= q except in the clamped tail)
    }
    __syncthreads();

    // ---- 3. level0 on the 16 x 32 tile: fragment f -> row f >> 1, columns (f & 1) * 16 + frow
    for (int f = wave; f < SF_TH * 2; f += 4) {
        const int row = f >> 1, col = (f & 1) * 16 + frow;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            int t = 2 * s + (fk >> 1);
            t = t < 9 ? t : 8;                              // the 10th tap has zero weights; any valid address will do
            const int dy = t / 3, dx = t - dy * 3;
            const f16x8 xf = *(const f16x8*)(bt + ((row + dy) * SF_BW + col + dx) * 16 + (fk & 1) * 8);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], xf, acc, 0, 0, 0);
        }
        const f16x4 h = {(f16)fmaxf(acc[0] + bl[0], 0.f), (f16)fmaxf(acc[1] + bl[1], 0.f), (f16)fmaxf(acc[2] + bl[2], 0.f),
                         (f16)fmaxf(acc[3] + bl[3], 0.f)};
        f16* op = a.out + ((size_t)(n * a.o_Hp + y0 + row + a.o_P) * a.o_Wp + x0 + col + a.o_P) * a.o_C + a.o_coff + fk * 4;
        *(f16x4*)op = h;
    }
}

hipError_t launch_stem_fused(const StemFusedArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(stem_fused_kernel, dim3(a.B * a.tiles_x * a.tiles_y), dim3(256), 0, s, a);
    return hipGetLastError();
}
