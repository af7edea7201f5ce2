// Calibration: the bare v_mfma_f32_16x16x32_f16 rate of this chip on RANDOM fp16 data, and whether the LDS operand
// traffic beside the MFMAs changes it (is ~1.23 PFLOP/s the chip's ceiling for the head convs, or
// would a 128 x 128 wave tile with fewer LDS bytes per MFMA run faster?).
//
// Every variant issues the SAME number of MFMAs per launch (256 workgroups, one per CU; 64 (two waves per SIMD) or 128 (one wave per SIMD) MFMAs per wave and step on
// independent accumulators; no global-memory traffic inside the loop):
//   reg2    operands register-resident, 8 waves per workgroup (two per SIMD), the head conv's wave tile (128 px x 64 ch:
//           8 pixel + 4 channel fragments, 32 accumulator tiles, every accumulator hit twice per step)
//   reg1    operands register-resident, 4 waves per workgroup (one per SIMD), a 128 x 128 wave tile (8 + 8 fragments, 64 tiles)
//   lds384  the head conv's read pattern: per 64 MFMAs 24 ds_read_b128 of swizzled rows of a 2 x 64 KB LDS image
//           (384 LDS bytes per MFMA), two waves per SIMD, no barriers
//   lds256  one wave per SIMD, 128 x 128 wave tile: 16 ds_read_b128 per 64 MFMAs (256 B per MFMA), fragments of step k+1
//           read while step k multiplies
//   lds384b lds384 with the conv kernels' two raw barriers per 16-MFMA phase and waves 4-7 one barrier behind
//   dma_l2 / dma_mall / dma_hbm   lds384b + the conv kernels' 64 KB of LDS-DMA per K-tile: all from L2; the pixel half from a 1 MB
//           window per workgroup (Infinity Cache); from an 8 MB window (HBM)
//   1w_lds / 1w_dma / 1w_mall   one wave per SIMD on a 128 x 128 wave tile, fragment sets ping-ponging per 32-deep K half, reads
//           and DMA pieces written between groups of eight MFMAs (mfma_rate_1w_kernel): what that layout gives in HIP C++
// Measured (profiles/r05_mfma_rate_random.txt): reg2 1.91 PF at 1.88 GHz, lds384b 1.70, dma_l2 1.45 (pipe 0.85 busy at 1.68 GHz),
// dma_hbm 1.14 (1.36 GHz); 1w_lds 1.60 (pipe 0.74: the compiler shuffles fragments through AGPRs), 1w_dma 0.96 (pipe 0.40: the
// only wave of a SIMD pays every DMA issue and every wait itself) - the one-wave layout needs loader waves of its own and a
// hand-scheduled MFMA stream before it can be compared with the 8-wave loop.
// Reported per variant: ms per launch, TFLOP/s, the in-kernel clock (s_memtime / s_memrealtime around the loop, median over
// waves, MI355X_MICROARCH.md "DVFS give-back" item 6) and the matrix-pipe occupancy that clock implies (16 cycles per MFMA and
// SIMD).  >= 2.5 s of back-to-back launches run before the timed ones.  Under `rocprofv3 --pmc GRBM_GUI_ACTIVE
// SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace` the same binary gives the counter view (tools/gpu_mfma_rate.sh).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mfma_rate.out tools/microbench/mfma_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LDS_AS __attribute__((address_space(3)))
#define LDS_F16X8(byte_addr) (*(const LDS_AS f16x8*)(uintptr_t)(byte_addr))

#define LDS_BYTES (128 * 1024)
// LDS-DMA from inline asm, as rtm3d_amd/csrc/common.h RT_DMA16 (m0 = LDS byte address of the wave's 1 KB run)
#define DMA16(gptr, lds_byte_addr) \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_byte_addr) : "memory", "m0")

struct Stamp { unsigned long long c0, r0, c1, r1; };

// MODE 0: registers, 1: LDS re-read, 2: LDS re-read + the conv kernels' barrier schedule.  NX / NW: pixel / channel fragments
// per wave (k halves 0 and 1 each).  WAVES: waves per workgroup.
template <int MODE, int NX, int NW, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma_rate_kernel(const f16* __restrict__ rnd, float* __restrict__ out, Stamp* __restrict__ stamps, int steps,
                                                               const f16* __restrict__ dma_w, const f16* __restrict__ dma_x, unsigned x_window) {
    __shared__ __attribute__((aligned(16))) f16 lds[LDS_BYTES / 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the whole LDS image = random halves (every workgroup another 128 KB of the 32 MB source)
    {
        const f16x8* src = (const f16x8*)(rnd + ((size_t)blockIdx.x % 128) * (LDS_BYTES / 2));
        for (int i = tid; i < LDS_BYTES / 16; i += WAVES * 64) ((f16x8*)lds)[i] = src[i];
    }
    __syncthreads();
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;
    const int frow = lane & 15, fk = lane >> 4;
    const int sw0 = ((0 * 4 + fk) ^ (frow & 7)) * 8, sw1 = ((1 * 4 + fk) ^ (frow & 7)) * 8;
    // rows of 64 halves (128 B), the conv kernels' layout: 16-row fragment p of a wave starts at row wave_row + 16 p
    const int xw = WAVES == 8 ? (wave & 1) : (wave & 1), ww = WAVES == 8 ? (wave >> 1) : (wave >> 1);
    const uint32_t xrow0 = (uint32_t)((((xw * NX * 16) + frow) * 64 + sw0) * 2), xrow1 = (uint32_t)((((xw * NX * 16) + frow) * 64 + sw1) * 2);
    const uint32_t wrow0 = (uint32_t)(((256 + ww * NW * 16 + frow) * 64 + sw0) * 2), wrow1 = (uint32_t)(((256 + ww * NW * 16 + frow) * 64 + sw1) * 2);

    f16x8 xf[NX][2], wf[NW][2];
#pragma unroll
    for (int p = 0; p < NX; ++p) { xf[p][0] = LDS_F16X8(lds_base + xrow0 + p * 2048); xf[p][1] = LDS_F16X8(lds_base + xrow1 + p * 2048); }
#pragma unroll
    for (int c = 0; c < NW; ++c) { wf[c][0] = LDS_F16X8(lds_base + wrow0 + c * 2048); wf[c][1] = LDS_F16X8(lds_base + wrow1 + c * 2048); }

    constexpr int REP = 1;                        // a step = one 64-deep K-tile of the wave tile: NX * NW * 2 MFMAs (64 or 128)
    f32x4 acc[NW][NX];
#pragma unroll
    for (int c = 0; c < NW; ++c)
#pragma unroll
        for (int p = 0; p < NX; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (MODE >= 2 && wave >= WAVES / 2) __builtin_amdgcn_s_barrier();       // second half of the waves one barrier behind
    unsigned xpos = 0;

    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t bufb = lds_base;
    for (int s = 0; s < steps; ++s) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < REP; ++r)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int c = 0; c < NW; ++c)
#pragma unroll
                        for (int p = 0; p < NX; ++p)
                            acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c][kk], xf[p][kk], acc[c][p], 0, 0, 0);
        } else if (NX == 8 && NW == 4) {
            // the head conv's four phases: X half A (4 frags) + W half A (2) | W half B (2) | X half B (4) | -, 16 MFMAs each
            // (its 64 x 32 quadrants are modelled as acc[c][p] with c = 2 j + cc, p = 4 i + pp)
            f16x8 xa[4][2], wa[2][2], wb[2][2];
#define RD_X(dst, half) _Pragma("unroll") for (int p = 0; p < 4; ++p) { dst[p][0] = LDS_F16X8(bufb + xrow0 + ((half) * 4 + p) * 2048); dst[p][1] = LDS_F16X8(bufb + xrow1 + ((half) * 4 + p) * 2048); }
#define RD_W(dst, half) _Pragma("unroll") for (int c = 0; c < 2; ++c) { dst[c][0] = LDS_F16X8(bufb + wrow0 + ((half) * 2 + c) * 2048); dst[c][1] = LDS_F16X8(bufb + wrow1 + ((half) * 2 + c) * 2048); }
#define SYNC() if (MODE >= 2) { if (MODE == 3) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
            // MODE 3: the conv kernels' staging beside it: per phase ONE 16 KB half-tile = two 1 KB LDS-DMA pieces per wave, into the
            // half-tile slot of the other buffer the real schedule targets; weights (W slots) from a small region every workgroup
            // shares (L2 hits), pixels (X slots) from the workgroup's own window of `x_window` bytes walked cyclically
#define STAGE(slot, other) if (MODE == 3) { \
                const uint32_t dst = lds_base + (((bufb - lds_base) ^ ((other) ? 64 * 1024 : 0)) + (slot) * 16384); \
                const f16* src = (slot) >= 2 ? dma_w + ((size_t)((s * 2 + (slot) - 2) & 63) * 8192) \
                                             : dma_x + (size_t)blockIdx.x * (x_window / 2) + (size_t)(xpos % (x_window / 2)); \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) DMA16(src + (i * 512 + tid) * 8, __builtin_amdgcn_readfirstlane(dst + (uint32_t)((i * 512 + wave * 64) * 16))); \
                if ((slot) < 2) xpos += 8192; }
#define MM(i, j, wfrag) \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int c = 0; c < 2; ++c) _Pragma("unroll") for (int p = 0; p < 4; ++p) \
                acc[(j) * 2 + c][(i) * 4 + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag[c][kk], xa[p][kk], acc[(j) * 2 + c][(i) * 4 + p], 0, 0, 0); \
            if (MODE >= 2) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
            RD_X(xa, 0) RD_W(wa, 0) STAGE(3, 1) SYNC() MM(0, 0, wa)
            RD_W(wb, 1) STAGE(1, 1) SYNC() MM(0, 1, wb)
            RD_X(xa, 1) STAGE(0, 0) SYNC() MM(1, 1, wb)
            STAGE(2, 0) SYNC() MM(1, 0, wa)
            bufb = lds_base + ((bufb - lds_base) ^ (64 * 1024));
        } else {
            // one wave per SIMD, 128 x 128 wave tile: fragments of the next step are read while this one multiplies
            f16x8 xn[NX][2], wn[NW][2];
            const uint32_t nb = lds_base + ((bufb - lds_base) ^ (64 * 1024));
#pragma unroll
            for (int p = 0; p < NX; ++p) { xn[p][0] = LDS_F16X8(nb + xrow0 + p * 2048); xn[p][1] = LDS_F16X8(nb + xrow1 + p * 2048); }
#pragma unroll
            for (int c = 0; c < NW; ++c) { wn[c][0] = LDS_F16X8(nb + wrow0 + c * 2048); wn[c][1] = LDS_F16X8(nb + wrow1 + c * 2048); }
#pragma unroll
            for (int r = 0; r < REP; ++r)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int c = 0; c < NW; ++c)
#pragma unroll
                        for (int p = 0; p < NX; ++p)
                            acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c][kk], xf[p][kk], acc[c][p], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < NX; ++p) { xf[p][0] = xn[p][0]; xf[p][1] = xn[p][1]; }
#pragma unroll
            for (int c = 0; c < NW; ++c) { wf[c][0] = wn[c][0]; wf[c][1] = wn[c][1]; }
            bufb = nb;
        }
        asm volatile("" : "+s"(bufb));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE >= 2 && wave < WAVES / 2) __builtin_amdgcn_s_barrier();

    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NW; ++c)
#pragma unroll
        for (int p = 0; p < NX; ++p) sum += acc[c][p][0] + acc[c][p][1] + acc[c][p][2] + acc[c][p][3];
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = sum;
    if (lane == 0) stamps[blockIdx.x * WAVES + wave] = Stamp{c0, r0, c1, r1};
}

// One wave per SIMD, 128 x 128 wave tile, written the way a kernel on that tile would have to be (round 5, for DESIGN section 8:
// is the "256 LDS bytes per MFMA" layout worth a rewrite of the conv256 family?).  Two fragment register sets ping-pong per
// 32-deep K half: while the 64 MFMAs of one half run on set A, the 16 fragments of the next half are read into set B, two per
// group of eight MFMAs (written in that order: the MFMAs of a group do not depend on the reads beside them).  DMA = 1: the same
// 64 KB of LDS-DMA per K-tile as the 8-wave loop (16 pieces per wave and K-tile, one per group), one barrier per K-tile behind the
// counted wait that retires the next K-tile's pieces.  Accumulators: 64 tiles = 256 registers (AGPRs), fragments 128.
template <int DMA>
__global__ __launch_bounds__(256) void mfma_rate_1w_kernel(const f16* __restrict__ rnd, float* __restrict__ out, Stamp* __restrict__ stamps, int steps,
                                                           const f16* __restrict__ dma_w, const f16* __restrict__ dma_x, unsigned x_window) {
    __shared__ __attribute__((aligned(16))) f16 lds[LDS_BYTES / 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const f16x8* src = (const f16x8*)(rnd + ((size_t)blockIdx.x % 128) * (LDS_BYTES / 2));
        for (int i = tid; i < LDS_BYTES / 16; i += 256) ((f16x8*)lds)[i] = src[i];
    }
    __syncthreads();
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_AS f16*)lds;
    const int frow = lane & 15, fk = lane >> 4;
    const int xw = wave & 1, ww = wave >> 1;
    uint32_t xrow[2], wrow[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fk) ^ (frow & 7)) * 8;
        xrow[kk] = (uint32_t)((((xw * 128) + frow) * 64 + sw) * 2);
        wrow[kk] = (uint32_t)(((256 + ww * 128 + frow) * 64 + sw) * 2);
    }
    f32x4 acc[8][8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 xa[8], wa[8], xb[8], wb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { xa[i] = LDS_F16X8(lds_base + xrow[0] + i * 2048); wa[i] = LDS_F16X8(lds_base + wrow[0] + i * 2048); }
    unsigned xpos = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t bufb = lds_base;
    for (int s = 0; s < steps; ++s) {
        const uint32_t nb = lds_base + ((bufb - lds_base) ^ (64 * 1024));
        // piece g (0..15) of this wave's share of the NEXT-BUT-ONE K-tile's 64 KB -> the buffer the first half just finished reading
#define DMA_PIECE(g, dstbuf) if (DMA) { \
            const int q_ = wave * 16 + (g); \
            const f16* src_ = q_ < 32 ? dma_x + (size_t)blockIdx.x * (x_window / 2) + (size_t)((xpos + q_ * 512) % (x_window / 2)) \
                                      : dma_w + (size_t)(((s & 15) * 32 + (q_ - 32)) * 512); \
            DMA16(src_ + lane * 8, __builtin_amdgcn_readfirstlane((dstbuf) + (uint32_t)(q_ * 1024))); }
        // ---- K half 0: MFMAs on set A (xa, wa), reads of this K-tile's half 1 into set B; DMA pieces 8..15 of the K-tile that goes into nb ... see below
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            DMA_PIECE(8 + c, nb)          // second half of the pieces of K-tile s + 1 (nb was released by the barrier of step s - 1)
            xb[c] = LDS_F16X8(bufb + xrow[1] + c * 2048);
            wb[c] = LDS_F16X8(bufb + wrow[1] + c * 2048);
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c], xa[p], acc[c][p], 0, 0, 0);
        }
        // every piece of K-tile s + 1 has landed (this wave's 16), every wave is done reading bufb's half-1 fragments... the barrier publishes both
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- K half 1: MFMAs on set B, reads of the NEXT K-tile's half 0 (buffer nb) into set A; DMA pieces 0..7 of K-tile s + 2 into bufb
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            DMA_PIECE(c, bufb)
            xa[c] = LDS_F16X8(nb + xrow[0] + c * 2048);
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[c], xb[p], acc[c][p], 0, 0, 0);
            wa[c] = LDS_F16X8(nb + wrow[0] + c * 2048);
        }
        if (DMA) xpos += 32 * 512;
        bufb = nb;
        asm volatile("" : "+s"(bufb));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int p = 0; p < 8; ++p) sum += acc[c][p][0] + acc[c][p][1] + acc[c][p][2] + acc[c][p][3];
    out[(size_t)blockIdx.x * 256 + tid] = sum;
    if (lane == 0) stamps[blockIdx.x * 4 + wave] = Stamp{c0, r0, c1, r1};
}
template <int DMA>
static void launch_1w(const f16* rnd, float* out, Stamp* st, int steps, hipStream_t s, const f16* dw, const f16* dx, unsigned xw) {
    hipLaunchKernelGGL((mfma_rate_1w_kernel<DMA>), dim3(256), dim3(256), 0, s, rnd, out, st, steps, dw, dx, xw);
}

struct Variant { const char* name; int waves; int lds_bytes_per_mfma; void (*launch)(const f16*, float*, Stamp*, int, hipStream_t, const f16*, const f16*, unsigned); unsigned x_window; };

template <int MODE, int NX, int NW, int WAVES>
static void launch(const f16* rnd, float* out, Stamp* st, int steps, hipStream_t s, const f16* dw, const f16* dx, unsigned xw) {
    hipLaunchKernelGGL((mfma_rate_kernel<MODE, NX, NW, WAVES>), dim3(256), dim3(WAVES * 64), 0, s, rnd, out, st, steps, dw, dx, xw);
}

int main(int argc, char** argv) {
    const double warm_s = argc > 1 ? atof(argv[1]) : 2.5;
    const int base_steps = argc > 2 ? atoi(argv[2]) : 4000;    // 64-deep K-steps per wave
    const int zero = argc > 3 ? atoi(argv[3]) : 0;             // 1: all-zero operands (the clock without operand energy)
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s, %d CUs; %s operands; %.1f s of back-to-back launches before each timed batch\n", prop.name, prop.multiProcessorCount,
           zero ? "ALL-ZERO" : "random fp16 in [-1, 1)", warm_s);
    const size_t n_rnd = (size_t)128 * (LDS_BYTES / 2);
    std::vector<f16> h(n_rnd);
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (auto& v : h) { x = x * 6364136223846793005ull + 1442695040888963407ull; v = zero ? (f16)0.f : (f16)(((int)(x >> 40) & 0xFFFF) / 32768.0f - 1.0f); }
    f16* d_rnd; float* d_out; Stamp* d_st;
    CK(hipMalloc(&d_rnd, n_rnd * 2));
    CK(hipMalloc(&d_out, (size_t)256 * 512 * 4));
    CK(hipMalloc(&d_st, (size_t)256 * 8 * sizeof(Stamp)));
    CK(hipMemcpy(d_rnd, h.data(), n_rnd * 2, hipMemcpyHostToDevice));
    // pixel source of the dma_* variants: 256 windows of up to 8 MB (random halves, replicated from the 32 MB image)
    f16* d_x;
    CK(hipMalloc(&d_x, (size_t)256 * (8u << 20) + (64 << 10)));
    for (size_t off = 0; off < (size_t)256 * (8u << 20); off += n_rnd * 2) CK(hipMemcpy((char*)d_x + off, d_rnd, n_rnd * 2, hipMemcpyDeviceToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    const Variant vs[] = {
        {"reg2   (registers, 2 waves/SIMD, 128x64 wave tile)", 8, 0, launch<0, 8, 4, 8>, 0},
        {"reg1   (registers, 1 wave/SIMD, 128x128 wave tile)", 4, 0, launch<0, 8, 8, 4>, 0},
        {"lds384 (LDS re-read 384 B/MFMA, 2 waves/SIMD)     ", 8, 384, launch<1, 8, 4, 8>, 0},
        {"lds256 (LDS re-read 256 B/MFMA, 1 wave/SIMD)      ", 4, 256, launch<1, 8, 8, 4>, 0},
        {"lds384b(as lds384 + the conv kernels' barriers)    ", 8, 384, launch<2, 8, 4, 8>, 0},
        {"dma_l2 (lds384b + 64 KB LDS-DMA per K-tile, L2 hits)", 8, 384, launch<3, 8, 4, 8>, 128u << 10},
        {"dma_mall(same, pixels: 1 MB window per workgroup)   ", 8, 384, launch<3, 8, 4, 8>, 1u << 20},
        {"dma_hbm (same, pixels: 8 MB window per workgroup)   ", 8, 384, launch<3, 8, 4, 8>, 8u << 20},
        {"1w_lds (1 wave/SIMD 128x128, hand-ordered, 256 B)    ", 4, 256, launch_1w<0>, 0},
        {"1w_dma (same + 64 KB LDS-DMA per K-tile, L2 hits)    ", 4, 256, launch_1w<1>, 128u << 10},
        {"1w_mall(same, pixels: 1 MB window per workgroup)     ", 4, 256, launch_1w<1>, 1u << 20},
    };
    printf("%-54s %9s %9s %9s %9s\n", "variant", "ms/launch", "TFLOP/s", "clock GHz", "pipe busy");
    for (const Variant& v : vs) {
        // 8 waves x 64 MFMAs per step == 4 waves x 128 MFMAs per step: the same MFMAs per launch and per SIMD in every variant
        const int steps = base_steps;
        const int mfma_per_step = v.waves == 8 ? 64 : 128;
        const double flop = 256.0 * v.waves * steps * (double)mfma_per_step * 16384.0;
        // back-to-back launches until warm_s has passed
        const auto t0 = std::chrono::steady_clock::now();
        int warm = 0;
        for (;;) {
            for (int i = 0; i < 20; ++i) v.launch(d_rnd, d_out, d_st, steps, s, d_rnd, d_x, v.x_window);
            warm += 20;
            CK(hipStreamSynchronize(s));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= warm_s) break;
        }
        const int timed = 60;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < timed; ++i) v.launch(d_rnd, d_out, d_st, steps, s, d_rnd, d_x, v.x_window);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= timed;
        std::vector<Stamp> st((size_t)256 * v.waves);
        CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
        std::vector<double> ghz, cyc;
        for (auto& q : st) { ghz.push_back((double)(q.c1 - q.c0) / (double)(q.r1 - q.r0) * 0.1); cyc.push_back((double)(q.c1 - q.c0)); }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        const double g = ghz[ghz.size() / 2], cy = cyc[cyc.size() / 2];
        // matrix-pipe occupancy inside the loop: MFMAs of one SIMD x 16 cycles / loop cycles
        const double busy = (double)(v.waves / 4) * steps * (double)mfma_per_step * 16.0 / cy;
        printf("%-54s %9.3f %9.1f %9.3f %9.3f   (%d warm-up launches, clock min %.3f max %.3f)\n", v.name, ms, flop / (ms * 1e-3) / 1e12, g, busy, warm, ghz.front(), ghz.back());
        fflush(stdout);
    }
    return 0;
}
