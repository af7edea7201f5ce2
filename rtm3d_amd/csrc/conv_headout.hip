// Final 3x3 convolution of the four detection heads: 256 -> {3,16,2,2} channels, fp32 NCHW logits
// (models/nets/header.py:17,27,32,37).  1.6 GMAC/image but 0.5 GB of input per head: HBM-bound.
//
// The generic implicit-GEMM kernel re-reads every input pixel once per tap through L2 (9x).  Here a
// workgroup stages an (8+2) x (32+2) pixel HALO TILE of one 64-channel chunk in LDS once
// (global_load_lds_dwordx4, double-buffered over the four chunks) and all nine taps read shifted rows of
// that tile, so each input byte crosses L2 ~1.4x.  Pixels are the MFMA B operand (16 consecutive x per
// 16x16x32 tile, swizzle chunk ^= pixel & 7 keeps ds_read_b128 conflict-free for shifted rows too);
// the 16-row weight operand comes straight from global/L1 in fragment order.
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

#define HO_TH 8
#define HO_TW 32
#define HO_HW (HO_TW + 2)                 // halo row pitch in pixels
#define HO_PIX 352                        // (8+2)*(32+2) = 340 halo pixels, padded to 11 x 32
#define HO_STAGE (HO_PIX * 64)            // halves per chunk stage

__global__ __launch_bounds__(256) void conv_headout_kernel(const HeadOutArgs a) {
    // single 44 KB stage: two to three workgroups per CU overlap each other's DMA / MFMA phases
    __shared__ __attribute__((aligned(16))) f16 lds[HO_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.y;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int y0 = ty * HO_TH, x0 = tx * HO_TW;

    // source offsets of the 11 pieces this thread stages per chunk: piece p = i*256 + tid -> halo pixel p/8
    uint32_t soff[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int p = i * 256 + tid;
        int q = p >> 3;
        const int cs = p & 7;
        q = q < (HO_TH + 2) * HO_HW ? q : (HO_TH + 2) * HO_HW - 1;
        int hy = q / HO_HW, hx = q - hy * HO_HW;
        int gy = y0 - 1 + hy, gx = x0 - 1 + hx;                 // unpadded coords, -1 .. H / W (border)
        gy = gy < a.H + a.in_P ? gy : a.H + a.in_P - 1;
        gx = gx < a.W + a.in_P ? gx : a.W + a.in_P - 1;
        const uint32_t pix = (uint32_t)((n * a.in_Hp + gy + a.in_P) * a.in_Wp + gx + a.in_P);
        soff[i] = pix * (uint32_t)a.in_C + (uint32_t)(head * 256) + (uint32_t)((cs ^ ((p >> 3) & 7)) * 8);
    }
    auto stage = [&](int buf, int chunk) {
        f16* dst = lds + buf * HO_STAGE;
#pragma unroll
        for (int i = 0; i < 11; ++i)
            __builtin_amdgcn_global_load_lds((const GLB_AS void*)(a.in + (size_t)soff[i] + chunk * 64),
                                             (LDS_AS void*)(dst + (i * 256 + wave * 64) * 8), 16, 0, 0);
    };

    const int frow = lane & 15, fk = lane >> 4;
    // this wave: tile rows 2*wave, 2*wave+1; pixel tiles p = (row r, x half h)
    f32x4 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16* wp = a.wgt + (size_t)head * 9 * 8 * 64 * 8 + lane * 8;

    const f32x4 bias4 = *(const f32x4*)(a.bias + head * 16 + fk * 4);
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        stage(0, c);
        // the chunk's 18 weight fragments (9 taps x 2 k-halves) are fetched beside the halo DMA: one wait
        // for both at the barrier instead of a vmcnt(0) round trip in front of every MFMA group
        f16x8 wfr[18];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wfr[tap * 2 + kk] = *(const f16x8*)(wp + (size_t)(tap * 8 + c * 2 + kk) * 64 * 8);
        __syncthreads();
        const f16* xl = lds;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f16x8 wf = wfr[tap * 2 + kk];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int q = (2 * wave + (p >> 1) + 1 + dy) * HO_HW + (p & 1) * 16 + frow + 1 + dx;
                    const f16x8 xf = *(const f16x8*)(xl + q * 64 + (((kk * 4 + fk) ^ (q & 7)) * 8));
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf, acc[p], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // epilogue: D[cout = fk*4 + e][pixel = frow] -> fp32 NCHW
    const int co = a.cout[head];
    float* o = a.out[head];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = y0 + 2 * wave + (p >> 1), x = x0 + (p & 1) * 16 + frow;
        if (y >= a.H || x >= a.W) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cc = fk * 4 + e;
            if (cc < co) o[((size_t)(n * co + cc) * a.H + y) * a.W + x] = acc[p][e] + bias4[e];
        }
    }
}

hipError_t launch_conv_headout(const HeadOutArgs& a, hipStream_t s) {
    dim3 grid(a.B * a.tiles_y * a.tiles_x, a.nheads, 1), block(256);
    hipLaunchKernelGGL(conv_headout_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}
