// Peaks-only regression heads (the detect3d call surface, detect.py:56-74).
//
// The reference's detector consumes the heat map densely (NMS + top-k) but reads the regression maps at the <= topk
// detected peaks only: `offset_fr_main` and `main_offset` through two gathers (models/model.py:47-50,124-128), and
// `vertex_offset` not at all.  Three of the four head branches (3 x 73 of the network's 436 GFLOP per image) are
// therefore evaluated here on the receptive field of the peaks instead of on the whole 96 x 320 map:
//
//   logits(peak)      <- 3x3 conv of h2 on the 3 x 3 window around the peak            (header.py:32-37)
//   h2 on that window <- 3x3 conv of h1 on the 5 x 5 window                            (header.py:24-31)
//   h1 on that window <- 3x3 dilation-6 conv of z at (5 x 5 positions) + 6 * {-1,0,1}  (header.py:15-23)
//
// The z samples of one peak are gathered into a 15 x 15 x 256 patch laid out so that patch pixel (5 a + i, 5 b + j) holds
// z(py + i - 2 + 6 (a - 1), px + j - 2 + 6 (b - 1)): on that patch the dilation-6 conv is an ordinary conv with taps at
// {0, 5, 10}^2 and NO padding, producing the 5 x 5 window; the next two convs are plain valid 3x3 convs (5x5 -> 3x3 -> 1x1).
// One "image" of the patch plan = one detection slot, so the existing MFMA conv kernels run it unchanged
// (rtm3d_amd/plan.py:build_peak_plan).  What the dense maps get from their zero borders has to be made explicit:
//   * z outside its padded tensor reads as zero (gather, below);
//   * h1 / h2 window positions that lie outside the image are the NEXT conv's zero padding, not conv results: they are
//     zeroed after each conv (patch_mask_kernel).
// This file: the gather, the mask op, and the launchers.  The final sub-pixel / vertex arithmetic is decode2d.hip's
// (rtm3d_decode2d_finish: the same fp32 operation order as the dense kernel).
#include "common.h"
#include "../../include/rtm3d_hip.h"

extern void rt_set_error(const char* fmt, ...);

struct PatchGatherArgs {
    const f16* z; int z_Hp, z_Wp, z_C, z_P, H, W;
    f16* patch; int32_t* yx;
    const int32_t* n; const float* peak_xy;
    int B, topk;
};

#define SP_PATCH 15
#define SP_PIECES (SP_PATCH * SP_PATCH * 32)        // 16-byte pieces of one 15 x 15 x 256 patch

__global__ __launch_bounds__(256) void patch_gather_kernel(const PatchGatherArgs a) {
    const int slot = blockIdx.x;
    const int b = slot / a.topk, rank = slot - b * a.topk;
    if (rank >= a.n[b]) {                               // empty slot: its patch keeps whatever it held; nobody reads its results
        if (threadIdx.x == 0) { a.yx[2 * slot] = -1; a.yx[2 * slot + 1] = -1; }
        return;
    }
    const int px = (int)a.peak_xy[2 * slot], py = (int)a.peak_xy[2 * slot + 1];
    if (threadIdx.x == 0) { a.yx[2 * slot] = py; a.yx[2 * slot + 1] = px; }
    const f16* zb = a.z + (size_t)b * a.z_Hp * a.z_Wp * a.z_C;
    f16* pb = a.patch + (size_t)slot * (SP_PATCH * SP_PATCH * 256);
    for (int p = threadIdx.x; p < SP_PIECES; p += 256) {
        const int pix = p >> 5, cs = p & 31;
        const int r = pix / SP_PATCH, c = pix - r * SP_PATCH;
        const int ra = r / 5, ri = r - ra * 5, ca = c / 5, ci = c - ca * 5;
        const int zy = py + ri - 2 + 6 * (ra - 1), zx = px + ci - 2 + 6 * (ca - 1);
        u32x4 v = {0u, 0u, 0u, 0u};
        // inside the padded tensor: its zero border IS the dilated conv's padding; beyond it (rows / columns only the masked
        // window positions would use): zero, and no access
        if (zy >= -a.z_P && zy < a.H + a.z_P && zx >= -a.z_P && zx < a.W + a.z_P)
            v = *(const u32x4*)(zb + ((size_t)(zy + a.z_P) * a.z_Wp + zx + a.z_P) * a.z_C + cs * 8);
        *(u32x4*)(pb + (size_t)pix * 256 + cs * 8) = v;
    }
}

extern "C" int rtm3d_gather_peak_patches(void* stream, const void* d_z, int z_H, int z_W, int z_C, int z_pad, int B, int topk,
                                         const int32_t* d_n, const float* d_peak_xy, void* d_patch, int32_t* d_yx,
                                         int patch_slots, int patch_S, size_t yx_bytes) {
    if (!d_z || !d_n || !d_peak_xy || !d_patch || !d_yx) { rt_set_error("gather_peak_patches: null pointer"); return 1; }
    // capacity of what the kernel writes: B * topk patches of SP_PATCH x SP_PATCH x 256 halves and 2 int32 per slot
    if (patch_S != SP_PATCH || B <= 0 || topk <= 0 || (long long)patch_slots < (long long)B * topk || yx_bytes < (size_t)8 * B * topk) {
        rt_set_error("gather_peak_patches: the patch tensor (%d slots of %d x %d) / (y, x) blob (%zu bytes) cannot hold %d x %d slots of %d x %d",
                     patch_slots, patch_S, patch_S, yx_bytes, B, topk, SP_PATCH, SP_PATCH);
        return 1;
    }
    if (B <= 0 || topk <= 0 || z_H <= 0 || z_W <= 0 || z_C != 256 || z_pad < 0) { rt_set_error("gather_peak_patches: bad shape (the fused map has 256 channels)"); return 1; }
    PatchGatherArgs a;
    a.z = (const f16*)d_z; a.z_Hp = z_H + 2 * z_pad; a.z_Wp = z_W + 2 * z_pad; a.z_C = z_C; a.z_P = z_pad; a.H = z_H; a.W = z_W;
    a.patch = (f16*)d_patch; a.yx = d_yx; a.n = d_n; a.peak_xy = d_peak_xy; a.B = B; a.topk = topk;
    hipLaunchKernelGGL(patch_gather_kernel, dim3(B * topk), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("gather_peak_patches launch: %s", hipGetErrorString(e)); return 1; }
    return 0;
}

// Window position (i, j) of slot s lies at image pixel (py + i - origin, px + j - origin): outside the image it is the next
// conv's zero padding.  One thread per window pixel; only border peaks write anything.
__global__ __launch_bounds__(256) void patch_mask_kernel(const PatchMaskArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int SS = a.S * a.S;
    if (idx >= a.n_slots * SS) return;
    const int slot = idx / SS, r = idx - slot * SS;
    const int py = a.yx[2 * slot], px = a.yx[2 * slot + 1];
    if (py < 0) return;
    const int i = r / a.S, j = r - i * a.S;
    const int y = py + i - a.origin, x = px + j - a.origin;
    if (y >= 0 && y < a.img_H && x >= 0 && x < a.img_W) return;
    u32x4* p = (u32x4*)(a.base + (size_t)idx * a.C);
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int c = 0; c < a.C / 8; ++c) p[c] = z;
}

hipError_t launch_patch_mask(const PatchMaskArgs& a, hipStream_t s) {
    const int total = a.n_slots * a.S * a.S;
    hipLaunchKernelGGL(patch_mask_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}
