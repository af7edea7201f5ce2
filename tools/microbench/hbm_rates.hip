// Calibration: what a streaming kernel gets from this chip's memory system, by direction.
//   read  : every lane 16-byte loads (plain / non-temporal), grid-stride, values folded into one word per workgroup
//   write : 16-byte stores
//   copy  : load + store;  r4w1: four read streams + one write stream (the softmax apply pass's shape)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/hbm_rates.out tools/microbench/hbm_rates.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ p, size_t n, unsigned* out) {
    u32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i + 256 * (UNROLL - 1) < n; i += stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}
__global__ __launch_bounds__(256) void k_write(u32x4* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    const u32x4 v = {1, 2, 3, (unsigned)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
}
template <int NR>
__global__ __launch_bounds__(256) void k_rw(const u32x4* __restrict__ a, const u32x4* __restrict__ b, const u32x4* __restrict__ c,
                                            const u32x4* __restrict__ d, u32x4* __restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        u32x4 v = __builtin_nontemporal_load(a + i);
        if (NR > 1) v ^= __builtin_nontemporal_load(b + i);
        if (NR > 2) v ^= __builtin_nontemporal_load(c + i);
        if (NR > 3) v ^= __builtin_nontemporal_load(d + i);
        o[i] = v;
    }
}

// every workgroup re-reads its own 16 KB x UNROLL.. region `reps` times in ONE launch: what a CU can take in from its XCD's L2
template <int NT>
__global__ __launch_bounds__(256) void k_reread(const u32x4* __restrict__ p, size_t per_wg, int reps, unsigned* out) {
    u32x4 acc = {0, 0, 0, 0};
    const u32x4* q = p + (size_t)blockIdx.x * per_wg;
    for (int r = 0; r < reps; ++r)
        for (size_t i = threadIdx.x; i + 768 < per_wg; i += 1024) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(q + i + u * 256) : q[i + u * 256];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc ^= v[u];
            asm volatile("" : "+v"(acc));
        }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

int main() {
    const size_t bytes = (size_t)512 << 20, n = bytes / 16;
    u32x4 *buf[5]; unsigned* out;
    for (auto& b : buf) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
    CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double gb, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
        }
        printf("%-44s %.3f ms  %.2f TB/s\n", name, best, gb / best);
    };
    const double G = bytes / 1e9;
    for (int grid : {2048, 8192, 32768}) {
        printf("grid %d x 256 threads, %zu MB per stream\n", grid, bytes >> 20);
        timeit("read  plain  1 x 16 B per lane and step", G, [&] { hipLaunchKernelGGL((k_read<0, 1>), dim3(grid), dim3(256), 0, 0, buf[0], n, out); });
        timeit("read  plain  4 x 16 B per lane and step", G, [&] { hipLaunchKernelGGL((k_read<0, 4>), dim3(grid), dim3(256), 0, 0, buf[0], n, out); });
        timeit("read  nt     4 x 16 B per lane and step", G, [&] { hipLaunchKernelGGL((k_read<1, 4>), dim3(grid), dim3(256), 0, 0, buf[0], n, out); });
        timeit("read  nt     8 x 16 B per lane and step", G, [&] { hipLaunchKernelGGL((k_read<1, 8>), dim3(grid), dim3(256), 0, 0, buf[0], n, out); });
        timeit("write", G, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, buf[4], n); });
        timeit("copy  (1 read + 1 write)", 2 * G, [&] { hipLaunchKernelGGL((k_rw<1>), dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], n); });
        timeit("2 reads + 1 write", 3 * G, [&] { hipLaunchKernelGGL((k_rw<2>), dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], n); });
        timeit("4 reads + 1 write", 5 * G, [&] { hipLaunchKernelGGL((k_rw<4>), dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], n); });
    }
    // where the bytes come from: the same sweep over 16 MB (each XCD re-reads its own 2 MB share: L2), 128 MB (Infinity Cache), 512 MB
    for (size_t mb : {16, 128, 512}) {
        const size_t nn = (mb << 20) / 16;
        char name[64];
        snprintf(name, sizeof name, "read  nt  4 x 16 B, %zu MB swept 8 times", mb);
        timeit(name, 8.0 * (mb << 20) / 1e9, [&] { for (int r = 0; r < 8; ++r) hipLaunchKernelGGL((k_read<1, 4>), dim3(2048), dim3(256), 0, 0, buf[0], nn, out); });
        snprintf(name, sizeof name, "read  plain 4 x 16 B, %zu MB swept 8 times", mb);
        timeit(name, 8.0 * (mb << 20) / 1e9, [&] { for (int r = 0; r < 8; ++r) hipLaunchKernelGGL((k_read<0, 4>), dim3(2048), dim3(256), 0, 0, buf[0], nn, out); });
    }
    for (int wgs : {256, 1024, 2048}) {
        // 16 MB in all (2 MB per XCD), 64 passes in one launch
        const size_t per_wg = ((size_t)16 << 20) / 16 / wgs;
        char name[96];
        snprintf(name, sizeof name, "L2 re-read, %d workgroups x %zu KB x 64 passes", wgs, per_wg * 16 >> 10);
        timeit(name, 64.0 * (16 << 20) / 1e9, [&] { hipLaunchKernelGGL((k_reread<0>), dim3(wgs), dim3(256), 0, 0, buf[0], per_wg, 64, out); });
    }
    return 0;
}
