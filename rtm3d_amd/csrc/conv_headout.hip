// Final 3x3 convolution of the four detection heads: 256 -> {3,16,2,2} channels, fp32 NCHW logits
// (models/nets/header.py:17,27,32,37).  1.6 GMAC/image but 0.5 GB of input per head: HBM-bound.
//
// The generic implicit-GEMM kernel re-reads every input pixel once per tap through L2 (9x).  Here a
// workgroup stages an (R+2) x (32+2) pixel HALO TILE (R = 8 or 16 rows) of one 32-channel half chunk in LDS
// (LDS-DMA, double-buffered over the eight half chunks) and all nine taps read shifted rows of that tile.
// Pixels are the MFMA B operand (16 consecutive x per 16x16x32 tile); the 16-row weight operand comes
// straight from global/L1 in fragment order, one half chunk ahead.  Workgroups take tiles in XCD-contiguous
// order (common.h), so the halo rows / columns neighbouring tiles share come out of one L2.
// Where the time goes (round 4, bs=32, timing-only builds on one box): staging alone 0.443 ms, operand reads + MFMAs
// alone 0.222, both 0.495: the kernel runs at the rate its 2.7 GB (8-row tiles) arrive in LDS, 6.1 TB/s - and with the
// XCD order only 2.1 GB of them come from HBM (PMC FETCH_SIZE 2.94 -> 2.12 GB), which did not change the time: the
// wall is the delivery into the CUs.  16-row tiles stage 10 % less: 0.503 -> 0.476.
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

// timing-only builds: -DHO_TIMING_NO_DMA (no staging), -DHO_TIMING_NO_MM (no operand reads / MFMAs)
#ifdef HO_TIMING_NO_DMA
#define HO_T_DMA 0
#else
#define HO_T_DMA 1
#endif
#ifdef HO_TIMING_NO_MM
#define HO_T_MM 0
#else
#define HO_T_MM 1
#endif
#define HO_TW 32
#define HO_HW (HO_TW + 2)                 // halo row pitch in pixels
#define HO_DMA16 RT_DMA16                           // common.h: the one LDS-DMA definition

// Round 4: the 256 input channels of a head are walked in EIGHT 32-channel half chunks through two 24 KB buffers: the DMA of
// half chunk h + 1 is in flight while h is multiplied (the former form staged a 64-channel chunk into one 44 KB buffer, waited,
// multiplied: the DMA engine of a CU idled during every multiply phase and only other workgroups covered it; a 16-row tile
// in that form - 20 % fewer staged bytes, two workgroups per CU - ran 0.62 ms against 0.50).  LDS layout of a buffer:
// [halo pixel q][4 slots of 8 channels], slot = chunk ^ 3 * ((halo column >> 2) & 1): a ds_read_b128 serves eight consecutive
// lanes of two k-chunk groups per cycle; pixels 4 apart share their 64-byte quarter of the bank row, the key gives the four
// (pixel, chunk) pairs four different slots (PMC: SQ_LDS_BANK_CONFLICT 50 % of the LDS cycles with the key (column >> 2) & 3).
// A lane's address is one base per (x half, column shift) plus a compile-time row offset.
template <int HO_TH>                      // tile rows: 8 (three workgroups per CU) or 16 (two; 10 % fewer staged bytes: 0.503 -> 0.476 ms at bs=32)
__global__ __launch_bounds__(256, HO_TH == 8 ? 3 : 2) void conv_headout_kernel(const HeadOutArgs a) {
    constexpr int HO_RPW = HO_TH / 4;                       // tile rows per wave
    constexpr int HO_HPIX = (HO_TH + 2) * HO_HW;            // halo pixels (340 / 612)
    constexpr int HO_NP = (HO_HPIX * 4 + 255) / 256;        // 16-byte pieces per thread and 32-channel half chunk (6 / 10)
    constexpr int HO_BUF = HO_NP * 256 * 8;                 // halves per buffer (24 / 40 KB)
    __shared__ __attribute__((aligned(16))) f16 lds[2 * HO_BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.y;
    // consecutive workgroups go to the 8 XCDs in turn: XCD k takes the k-th eighth of the tiles, so that neighbouring tiles -
    // which share two halo rows / columns - are staged through ONE L2 (PMC before: 2.94 GB fetched for 2.0 GB of input)
    int t = xcd_contiguous_index(blockIdx.x, gridDim.x);
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int y0 = ty * HO_TH, x0 = tx * HO_TW;

    // source offsets of the pieces this thread stages per half chunk: piece p = i*256 + tid -> halo pixel p/4, slot p%4
    uint32_t soff[HO_NP];
#pragma unroll
    for (int i = 0; i < HO_NP; ++i) {
        const int p = i * 256 + tid;
        int q = p >> 2;
        const int cs = p & 3;
        q = q < HO_HPIX ? q : HO_HPIX - 1;
        int hy = q / HO_HW, hx = q - hy * HO_HW;
        int gy = y0 - 1 + hy, gx = x0 - 1 + hx;                 // unpadded coords, -1 .. H / W (border)
        gy = gy < a.H + a.in_P ? gy : a.H + a.in_P - 1;
        gx = gx < a.W + a.in_P ? gx : a.W + a.in_P - 1;
        const uint32_t pix = (uint32_t)((n * a.in_Hp + gy + a.in_P) * a.in_Wp + gx + a.in_P);
        soff[i] = pix * (uint32_t)a.in_C + (uint32_t)(head * 256) + (uint32_t)((cs ^ (((hx >> 2) & 1) * 3)) * 8);
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;
    auto stage = [&](int buf, int h) {
        const f16* src = a.in + h * 32;
#pragma unroll
        for (int i = 0; i < (HO_T_DMA ? HO_NP : 0); ++i)
            HO_DMA16(src + (size_t)soff[i], __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((buf * HO_BUF + (i * 256 + wave * 64) * 8) * 2)));
    };

    const int frow = lane & 15, fk = lane >> 4;
    // this wave: tile rows 2*wave, 2*wave+1; pixel tiles p = (row r, x half hx); LDS byte offsets of this lane's operand for
    // 2 halves x 3 column shifts (halo row 2*wave of buffer 0; rows and buffer 1 are compile-time offsets)
    uint32_t xoff[2][3];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int col = hx * 16 + frow + d;
            xoff[hx][d] = lds_base + (uint32_t)(((HO_RPW * wave * HO_HW + col) * 32 + ((fk ^ (((col >> 2) & 1) * 3)) * 8)) * 2);
        }
    f32x4 acc[2 * HO_RPW];
#pragma unroll
    for (int p = 0; p < 2 * HO_RPW; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16* wp = a.wgt + (size_t)head * 9 * 8 * 64 * 8 + lane * 8;
    const f32x4 bias4 = *(const f32x4*)(a.bias + head * 16 + fk * 4);

    // weight fragments of half chunk h: [tap][h] in the packed order (9 per half chunk), fetched one half chunk ahead
    f16x8 wfr[2][9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wfr[0][tap] = *(const f16x8*)(wp + (size_t)(tap * 8) * 64 * 8);
    stage(0, 0);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        if (h + 1 < 8) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wfr[(h + 1) & 1][tap] = *(const f16x8*)(wp + (size_t)(tap * 8 + h + 1) * 64 * 8);
            stage((h + 1) & 1, h + 1);
            // everything but the 9 + HO_NP newest loads (the next half chunk's weights and pieces) has landed
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(9 + (HO_T_DMA ? HO_NP : 0)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < (HO_T_MM ? 9 : 0); ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            const f16x8 wf = wfr[h & 1][tap];
#pragma unroll
            for (int p = 0; p < 2 * HO_RPW; ++p) {
                const f16x8 xf = *(const LDS_AS f16x8*)(uintptr_t)(xoff[p & 1][dx] + (uint32_t)(((p >> 1) + dy) * HO_HW * 64 + (h & 1) * HO_BUF * 2));
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf, acc[p], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // buffer h & 1 is free for half chunk h + 2
        __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: D[cout = fk*4 + e][pixel = frow] -> fp32 NCHW
    const int co = a.cout[head];
    float* o = a.out[head];
#pragma unroll
    for (int p = 0; p < 2 * HO_RPW; ++p) {
        const int y = y0 + HO_RPW * wave + (p >> 1), x = x0 + (p & 1) * 16 + frow;
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
    if (a.tile_rows == 16) hipLaunchKernelGGL(conv_headout_kernel<16>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(conv_headout_kernel<8>, grid, block, 0, s, a);
    return hipGetLastError();
}
