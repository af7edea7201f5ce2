// Entry of DLA-34 level2 (models/nets/dla.py:186-206 with Tree(levels=1, stride=2, 32 -> 64)), three ops on the same
// 32-channel half-resolution map in ONE launch:
//   bottom   = max_pool2d(x, 2, 2)                         (:190-193)   - never materialised
//   residual = BN(conv1x1(bottom))  32 -> 64, no ReLU      (:195-198, project)
//   mid      = ReLU(BN(conv3x3 stride 2 (x)))  32 -> 64    (tree1.conv1, :86-92)
// Separately they read x three times (252 MB each at bs=32) and ran at 0.066 + 0.045 + 0.18 ms; the stride-2 conv on the
// register-direct kernel was bound by neither HBM nor MFMA (1.0 TB/s, 200 TFLOP/s).  Here a persistent workgroup
// (8 waves) owns an 8 x 32 output tile: its (17 x 65)-pixel input halo (70.7 KB) is staged once by LDS-DMA, double
// buffered; both filter banks live in registers (9 + 1 taps x 2 channel tiles per wave: K = 32 is one MFMA deep);
// the 2x2 max is taken on the operand fragments in registers.  Tiles come from one atomic ticket counter per op.
// Halo layout: pixel pitch 64 B (4 chunks of 8 channels), chunk position = chunk ^ ((column >> 2) & 3): the stride-2
// operand reads of 16 lanes then fall on 8 distinct 16-byte bank slots (2-way conflict; the kernel is HBM-bound).
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
#define S2_HW 65                                   // halo columns (2 * 32 + 1)
#define S2_HH 17                                   // halo rows    (2 * 8 + 1)
#define S2_PIECES (S2_HH * S2_HW * 4)              // 4420 16-byte pieces
#define S2_NDMA 9                                  // DMA instructions per wave and tile (9 x 512 lanes >= 4420)
#define S2_BUF_PIECES (S2_NDMA * 512)
#define S2_DMA16 RT_DMA16                           // common.h: the one LDS-DMA definition
#define S2_LDS_F16X8(byte_addr) (*(const LDS_AS f16x8*)(uintptr_t)(byte_addr))

__global__ __launch_bounds__(512) void conv32s2_fused_kernel(const Conv32S2Args a, unsigned int* ticket_ctr, const int single) {
    __shared__ __attribute__((aligned(128))) f16 lds[2 * S2_BUF_PIECES * 8];
    __shared__ int tk[3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wp = wave >> 1;                  // 32-channel half of the 64 outputs, pixel-row pair of the tile
    const int frow = lane & 15, fk = lane >> 4;
    const int tiles_x = a.Wo >> 5, tpi = tiles_x * (a.Ho >> 3);
    const int total = a.B * tpi;

    f16x8 wreg[9][2], wproj[2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) wreg[t][c] = *(const f16x8*)(a.w_conv + ((size_t)((t * 4 + wc * 2 + c) * 64 + lane)) * 8);
#pragma unroll
    for (int c = 0; c < 2; ++c) wproj[c] = *(const f16x8*)(a.w_proj + ((size_t)((wc * 2 + c) * 64 + lane)) * 8);
    f32x4 bc[2], bp[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        bc[c] = *(const f32x4*)(a.b_conv + wc * 32 + c * 16 + fk * 4);
        bp[c] = *(const f32x4*)(a.b_proj + wc * 32 + c * 16 + fk * 4);
    }
    // per-thread source offsets (elements, relative to the halo origin) of its DMA pieces
    uint32_t poff[S2_NDMA];
#pragma unroll
    for (int i = 0; i < S2_NDMA; ++i) {
        int p = i * 512 + tid;
        p = p < S2_PIECES ? p : S2_PIECES - 1;
        const int hq = p >> 2, pos = p & 3;
        const int hy = hq / S2_HW, hx = hq - hy * S2_HW;
        poff[i] = (uint32_t)((hy * a.in_Wp + hx) * a.in_C + ((pos ^ ((hx >> 2) & 3)) * 8));
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;

    // (ONE draw of three: three separate returning atomics per workgroup - 768 on one word, three round trips in a row
    // before the first MFMA - cost 8-10 us per launch; the word serves ~88 draws per microsecond)
    // `single` (no more work items than CUs: small batches): workgroup b takes item b and nothing else - with three tickets per
    // draw only a third of the workgroups would work, three items each in a row (bs=1 at 384 x 1280: 40 of 120)
    int cur, nxt, nn;
    if (single) {
        cur = blockIdx.x; nxt = nn = total;
    } else {
        if (tid == 0) tk[0] = (int)atomicAdd(ticket_ctr, 3u);
        __syncthreads();
        const int tk0 = __builtin_amdgcn_readfirstlane(tk[0]);
        cur = tk0; nxt = tk0 + 1; nn = tk0 + 2;
    }
    if (cur >= total) return;
    __syncthreads();

    // halo origin of tile v: input pixel (2 * ty * 8 - 1, 2 * tx * 32 - 1) of image n (the input border supplies the padding)
    auto stage = [&](int v, int par) {
        const int n = v / tpi, r = v - n * tpi;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const f16* src = a.in + ((size_t)(n * a.in_Hp + ty * 16 - 1 + a.in_P) * a.in_Wp + tx * 64 - 1 + a.in_P) * a.in_C + a.in_coff;
#pragma unroll
        for (int i = 0; i < S2_NDMA; ++i)
            S2_DMA16(src + poff[i], __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((par * S2_BUF_PIECES + i * 512 + wave * 64) * 16)));
    };
    // this lane's operand (8 channels = chunk fk) of halo pixel (hy, hx)
    auto frag = [&](uint32_t hb, int hy, int hx) -> f16x8 {
        return S2_LDS_F16X8(hb + (uint32_t)((hy * S2_HW + hx) * 64 + ((fk ^ ((hx >> 2) & 3)) * 16)));
    };
    const int so = (fk & 1) * 16 + (fk >> 1) * 8;

    stage(cur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int par = 0, it = 0;
    for (;;) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();               // halo of `cur` has landed for every wave; buffer par ^ 1 is free
        __builtin_amdgcn_sched_barrier(0);
        if (it > 0) nn = __builtin_amdgcn_readfirstlane(tk[(it - 1) & 1]);
        const bool more = nxt < total;
        if (more) stage(nxt, par ^ 1);

        const uint32_t hb = lds_base + (uint32_t)(par * S2_BUF_PIECES * 16);
        f32x4 acc[2][4], accp[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p) { acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f}; accp[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        // fragment p: output row 2*wp + (p >> 1), output columns (p & 1) * 16 + frow;
        // output (oy, ox) reads halo pixels (2 oy + dy, 2 ox + dx), dy, dx in 0..2; its pool window is dy, dx in 1..2
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int oy = 2 * wp + (p >> 1), ox = (p & 1) * 16 + frow;
            f16x8 pool;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t % 3;
                const f16x8 xf = frag(hb, 2 * oy + dy, 2 * ox + dx);
                if (dy >= 1 && dx >= 1) pool = (dy == 1 && dx == 1) ? xf : __builtin_elementwise_max(pool, xf);
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[t][c], xf, acc[c][p], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) accp[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wproj[c], pool, accp[c][p], 0, 0, 0);
        }
        if (!single && tid == 0) tk[it & 1] = (int)atomicAdd(ticket_ctr, 1u);

        // ---- epilogue of `cur`: mid = ReLU(conv + bias) and residual = project + bias, 16-byte swapped stores
        {
            const int n = cur / tpi, r = cur - n * tpi;
            const int ty = r / tiles_x, tx = r - ty * tiles_x;
            const f16x4 z4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int oy = ty * 8 + 2 * wp + (p >> 1), ox = tx * 32 + (p & 1) * 16 + frow;
                uint32_t u[2][2], up[2][2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x4 v = acc[c][p] + bc[c], w = accp[c][p] + bp[c];
                    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    h = __builtin_elementwise_max(h, z4);
                    const f16x4 hp = {(f16)w[0], (f16)w[1], (f16)w[2], (f16)w[3]};
                    __builtin_memcpy(u[c], &h, 8);
                    __builtin_memcpy(up[c], &hp, 8);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
                const auto q0 = __builtin_amdgcn_permlane16_swap(up[0][0], up[1][0], false, false);
                const auto q1 = __builtin_amdgcn_permlane16_swap(up[0][1], up[1][1], false, false);
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]}, op = {q0[0], q1[0], q0[1], q1[1]};
                *(u32x4*)(a.out_conv + ((size_t)(n * a.oc_Hp + oy + a.oc_P) * a.oc_Wp + ox + a.oc_P) * a.oc_C + a.oc_coff + wc * 32 + so) = o;
                *(u32x4*)(a.out_proj + ((size_t)(n * a.op_Hp + oy + a.op_P) * a.op_Wp + ox + a.op_P) * a.op_C + a.op_coff + wc * 32 + so) = op;
            }
        }
        if (!more) break;
        // the next tile's halo (issued before this tile's 8 stores) must have landed; the stores may stay in flight
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        cur = nxt; nxt = nn;
        par ^= 1;
        ++it;
    }
}

hipError_t launch_conv32s2_fused(const Conv32S2Args& a, int cu_count, unsigned int* ticket_ctr, hipStream_t s) {
    const int total = a.B * (a.Wo >> 5) * (a.Ho >> 3);
    const int grid = cu_count < total ? cu_count : total;
    hipLaunchKernelGGL(conv32s2_fused_kernel, dim3(grid), dim3(512), 0, s, a, ticket_ctr, total <= cu_count ? 1 : 0);
    return hipGetLastError();
}
