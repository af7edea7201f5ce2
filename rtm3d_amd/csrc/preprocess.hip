// Input pipeline step in front of the hot path:
//   letterbox-pad an already resized uint8 HWC image into the network canvas with the image's own mean
//   colour (datasets/dataset_reader.py:175-195), then Normalize -> ToTensor -> ToNCHW
//   (preprocess/transforms.py:110-120, 312-322; datasets/dataset_reader.py:63-69).
// The reference normalises in float64 and rounds once to float32; there are only 3 x 256 distinct
// results, so the host computes them exactly the same way into a look-up table and the kernel is a pure
// HBM-bound gather: 3 B read + 12 B written per pixel.
#include "common.h"
#include "../../include/rtm3d_hip.h"

__global__ __launch_bounds__(256) void channel_sum_kernel(const uint8_t* __restrict__ img, int npix, unsigned long long* __restrict__ sums) {
    unsigned int s0 = 0, s1 = 0, s2 = 0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        s0 += img[3 * p]; s1 += img[3 * p + 1]; s2 += img[3 * p + 2];
    }
    __shared__ unsigned int sh[3][256];
    sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) for (int c = 0; c < 3; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 3) atomicAdd(&sums[threadIdx.x], (unsigned long long)sh[threadIdx.x][0]);
}

__global__ __launch_bounds__(256) void letterbox_normalize_kernel(const uint8_t* __restrict__ img, int h, int w, float* __restrict__ out,
                                                                 int H, int W, int pad_h, int pad_w, const float* __restrict__ lut,
                                                                 const unsigned long long* __restrict__ sums) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    const int sy = y - pad_h, sx = x - pad_w;
    const bool inside = sy >= 0 && sy < h && sx >= 0 && sx < w;
    const unsigned long long npix = (unsigned long long)h * w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // np.full(..., cv2.mean(img)[:3], dtype=np.uint8): arithmetic mean, truncated towards zero
        const unsigned int v = inside ? img[((size_t)sy * w + sx) * 3 + c] : (unsigned int)(sums[c] / npix);
        out[(size_t)c * H * W + idx] = lut[c * 256 + v];
    }
}

extern void rt_set_error(const char* fmt, ...);

extern "C" int rtm3d_preprocess(void* stream, const uint8_t* d_img_hwc, int h, int w, float* d_out_chw, int H, int W,
                                const float* d_lut, unsigned long long* d_sums3) {
    if (!d_img_hwc || !d_out_chw || !d_lut || !d_sums3) { rt_set_error("preprocess: null pointer"); return 1; }
    if (h < 1 || w < 1 || h > H || w > W) { rt_set_error("preprocess: image %dx%d does not fit the %dx%d canvas", h, w, H, W); return 1; }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_sums3, 0, 3 * sizeof(unsigned long long), s) != hipSuccess) { rt_set_error("preprocess: memset failed"); return 1; }
    const int npix = h * w;
    int blocks = (npix + 255) / 256;
    blocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(channel_sum_kernel, dim3(blocks), dim3(256), 0, s, d_img_hwc, npix, d_sums3);
    const int pad_h = (H - h) / 2, pad_w = (W - w) / 2;
    hipLaunchKernelGGL(letterbox_normalize_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, d_img_hwc, h, w, d_out_chw, H, W, pad_h,
                       pad_w, d_lut, d_sums3);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("preprocess launch: %s", hipGetErrorString(e)); return 1; }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Batched form: B ragged uint8 HWC images -> Resize (preprocess/transforms.py:480-495) -> mean-colour letterbox
// (datasets/dataset_reader.py:175-195) -> Normalize/ToTensor (transforms.py:110-120, 312-317) in TWO launches for the
// whole batch (interior + channel sums, then borders), writing either the reference's fp32 NCHW batch or directly the
// network's own operand: the 4-channel padded NHWC fp16 tensor the stem reads (no fp32 round trip through HBM).
//
// Resize = cv2.resize(..., INTER_LINEAR) on 8-bit data, restated from OpenCV's published algorithm (imgproc
// resize.cpp, 8UC3 path; OpenCV is a third-party dependency absent from /root/reference AND from this image, the
// reference pins no version - PARITY UNPINNED for this step):
//   fx = (float)((dx + 0.5) * (double)(src_w / dst_w) - 0.5); sx = floor(fx); fx -= sx; clamped at both ends (fx = 0);
//   coefficients a0 = round_half_even((1 - fx) * 2048), a1 = round_half_even(fx * 2048) as int16 (same for rows);
//   horizontal pass in int32: r = S[sx] * a0 + S[sx + 1] * a1; vertical pass + rounding:
//   dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
// With equal source and destination sizes this is the identity (OpenCV returns a copy), so the no-resize case is
// bit-identical to rtm3d_preprocess above.
#define PRE_MAX_BATCH 64
struct PreBatch {
    const uint8_t* img[PRE_MAX_BATCH];
    int h[PRE_MAX_BATCH], w[PRE_MAX_BATCH];       // source size
    int rh[PRE_MAX_BATCH], rw[PRE_MAX_BATCH];     // size after Resize (== source size: no resize)
};

__device__ __forceinline__ void resize_coef(int d, double scale, int ssize, int& s0, int& s1, int& c0, int& c1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (si < 0) { f = 0.f; si = 0; }
    if (si >= ssize - 1) { f = 0.f; si = ssize - 1; }
    s0 = si;
    s1 = si + 1 < ssize ? si + 1 : ssize - 1;
    c0 = (int)rintf((1.f - f) * 2048.f);
    c1 = (int)rintf(f * 2048.f);
}

__device__ __forceinline__ unsigned int resized_pixel(const uint8_t* __restrict__ img, int w, int c, int y0, int y1, int b0, int b1,
                                                      int x0, int x1, int a0, int a1) {
    const int r0 = (int)img[((size_t)y0 * w + x0) * 3 + c] * a0 + (int)img[((size_t)y0 * w + x1) * 3 + c] * a1;
    const int r1 = (int)img[((size_t)y1 * w + x0) * 3 + c] * a0 + (int)img[((size_t)y1 * w + x1) * 3 + c] * a1;
    return (unsigned int)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

// MODE 0: fp32 NCHW (B,3,H,W); MODE 1: fp16 NHWC4 with border P: [B][H+2P][W+2P][4], 4th channel 0
template <int MODE>
__device__ __forceinline__ void pre_store(void* out, int b, int y, int x, int H, int W, int P, const float* __restrict__ lut,
                                          const f16* __restrict__ lut16, unsigned int v0, unsigned int v1, unsigned int v2) {
    if (MODE == 0) {
        float* o = (float*)out + (size_t)b * 3 * H * W + (size_t)y * W + x;
        o[0] = lut[v0]; o[(size_t)H * W] = lut[256 + v1]; o[(size_t)2 * H * W] = lut[512 + v2];
    } else {
        const f16x4 px = {lut16[v0], lut16[256 + v1], lut16[512 + v2], (f16)0.f};
        *(f16x4*)((f16*)out + (((size_t)b * (H + 2 * P) + y + P) * (W + 2 * P) + x + P) * 4) = px;
    }
}

// interior: a workgroup takes bands of `band_rows` consecutive rows of the resized image (grid.x strides the bands, grid.y =
// image).  Per workgroup, once: the column coefficients (source column, a0, a1 - they depend on x only) and the fp16 table
// go to LDS.  Per band: the source rows it needs are ONE contiguous byte span of the HWC image; it is copied to LDS with
// 16-byte loads (head / tail bytes singly, nothing outside the image is touched) and the 12 byte-gathers per output pixel
// read LDS instead of global memory; the row coefficients are uniform per row.  Per-image channel sums via LDS + 3
// atomics per block.  History (32 images 360 x 1240 -> 371 x 1280, fp16 NHWC4 out): one thread per pixel redoing the
// fp64 coordinate arithmetic and an integer division 0.146 ms; column table 0.102 ms; + staged source rows: see
// profiles/r02_n1_input_path.txt.  A band whose source span does not fit the stage (very wide images) gathers from
// global memory as before.
template <int MODE>
__global__ __launch_bounds__(256) void pre_interior_kernel(const PreBatch pb, void* __restrict__ out, int H, int W, int P,
                                                          const float* __restrict__ lut, const f16* __restrict__ lut16,
                                                          unsigned long long* __restrict__ sums, int band_rows, int col_bytes,
                                                          int stage_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pre_dyn[];     // [rw] int2 (x0, a0 | a1 << 16), then the stage
    __shared__ f16 lut_s[768];
    __shared__ unsigned int sh[3][256];
    int2* const col = (int2*)pre_dyn;
    unsigned char* const stage = pre_dyn + col_bytes;
    const int b = blockIdx.y;
    const int rh = pb.rh[b], rw = pb.rw[b], h = pb.h[b], w = pb.w[b];
    const uint8_t* __restrict__ img = pb.img[b];
    const double sx_scale = (double)w / (double)rw, sy_scale = (double)h / (double)rh;
    const int pad_h = (H - rh) / 2, pad_w = (W - rw) / 2;
    const bool same = rh == h && rw == w;
    if (MODE == 1) for (int i = threadIdx.x; i < 768; i += 256) lut_s[i] = lut16[i];
    for (int x = threadIdx.x; x < rw; x += 256) {
        int x0 = x, x1 = x, a0 = 2048, a1 = 0;
        if (!same) resize_coef(x, sx_scale, w, x0, x1, a0, a1);
        col[x] = make_int2(x0 | ((x1 - x0) << 30), a0 | (a1 << 16));               // x1 - x0 is 0 or 1; a0, a1 <= 2048
    }
    unsigned int s0 = 0, s1 = 0, s2 = 0;
    const size_t pitch = (size_t)w * 3;
    for (int r0 = blockIdx.x * band_rows; r0 < rh; r0 += gridDim.x * band_rows) {
        const int r1 = min(r0 + band_rows, rh);
        // source rows of the band: y0 of its first row .. y1 of its last
        int ylo = r0, yhi = r1 - 1, t0, t1, t2;
        if (!same) { resize_coef(r0, sy_scale, h, ylo, t0, t1, t2); resize_coef(r1 - 1, sy_scale, h, t0, yhi, t1, t2); }
        const uint8_t* gs = img + (size_t)ylo * pitch;
        const size_t len = (size_t)(yhi - ylo + 1) * pitch;
        const int a0g = (int)((uintptr_t)gs & 15);
        const bool staged = len + 15 <= (size_t)stage_bytes;
        __syncthreads();                                   // column table written / previous band's reads done
        if (staged) {
            const uint8_t* ga = gs - a0g;                  // 16-byte aligned; chunk k covers ga + 16k .. + 15
            const int nchunks = (int)((a0g + len + 15) >> 4);
            for (int k = threadIdx.x; k < nchunks; k += 256) {
                const uint8_t* c = ga + (size_t)k * 16;
                if (c >= gs && c + 16 <= gs + len) {
                    *(u32x4*)(stage + (size_t)k * 16) = *(const u32x4*)c;
                } else {
                    for (int j = 0; j < 16; ++j) if (c + j >= gs && c + j < gs + len) stage[(size_t)k * 16 + j] = c[j];
                }
            }
            __syncthreads();
        }
        for (int y = r0; y < r1; ++y) {
            int y0 = y, y1 = y, b0 = 2048, b1 = 0;
            if (!same) resize_coef(y, sy_scale, h, y0, y1, b0, b1);
            for (int x = threadIdx.x; x < rw; x += 256) {
                const int2 c = col[x];
                const int x0 = c.x & 0x3fffffff, dx3 = ((unsigned int)c.x >> 30) * 3;
                const int a0 = c.y & 0xffff, a1 = (unsigned int)c.y >> 16;
                unsigned int v[3];
                if (staged) {
                    const unsigned char* p0 = stage + a0g + (size_t)(y0 - ylo) * pitch + (size_t)x0 * 3;
                    const unsigned char* p1 = stage + a0g + (size_t)(y1 - ylo) * pitch + (size_t)x0 * 3;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const int q0 = (int)p0[ch] * a0 + (int)p0[dx3 + ch] * a1;
                        const int q1 = (int)p1[ch] * a0 + (int)p1[dx3 + ch] * a1;
                        v[ch] = same ? (unsigned int)p0[ch] : (unsigned int)((((b0 * (q0 >> 4)) >> 16) + ((b1 * (q1 >> 4)) >> 16) + 2) >> 2);
                    }
                } else {
                    const uint8_t* p0 = img + (size_t)y0 * pitch + (size_t)x0 * 3;
                    const uint8_t* p1 = img + (size_t)y1 * pitch + (size_t)x0 * 3;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const int q0 = (int)p0[ch] * a0 + (int)p0[dx3 + ch] * a1;
                        const int q1 = (int)p1[ch] * a0 + (int)p1[dx3 + ch] * a1;
                        v[ch] = same ? (unsigned int)p0[ch] : (unsigned int)((((b0 * (q0 >> 4)) >> 16) + ((b1 * (q1 >> 4)) >> 16) + 2) >> 2);
                    }
                }
                s0 += v[0]; s1 += v[1]; s2 += v[2];
                if (MODE == 0) {
                    pre_store<0>(out, b, y + pad_h, x + pad_w, H, W, P, lut, lut16, v[0], v[1], v[2]);
                } else {
                    const f16x4 px = {lut_s[v[0]], lut_s[256 + v[1]], lut_s[512 + v[2]], (f16)0.f};
                    *(f16x4*)((f16*)out + (((size_t)b * (H + 2 * P) + y + pad_h + P) * (W + 2 * P) + x + pad_w + P) * 4) = px;
                }
            }
        }
    }
    sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) for (int c = 0; c < 3; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 3) atomicAdd(&sums[b * 3 + threadIdx.x], (unsigned long long)sh[threadIdx.x][0]);
}

// border: the canvas pixels outside the centred image get np.full(..., cv2.mean(img)[:3], dtype=np.uint8) of the RESIZED image
template <int MODE>
__global__ __launch_bounds__(256) void pre_border_kernel(const PreBatch pb, void* __restrict__ out, int H, int W, int P,
                                                        const float* __restrict__ lut, const f16* __restrict__ lut16,
                                                        const unsigned long long* __restrict__ sums) {
    const int b = blockIdx.y;
    const int rh = pb.rh[b], rw = pb.rw[b];
    const int pad_h = (H - rh) / 2, pad_w = (W - rw) / 2;
    const int nborder = H * W - rh * rw;
    const unsigned long long npix = (unsigned long long)rh * rw;
    const unsigned int m0 = (unsigned int)(sums[b * 3] / npix), m1 = (unsigned int)(sums[b * 3 + 1] / npix), m2 = (unsigned int)(sums[b * 3 + 2] / npix);
    // enumerate border pixels: rows above, rows below, then the left/right strips of the image rows
    const int top = pad_h * W, bot = (H - pad_h - rh) * W, side = W - rw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nborder; i += gridDim.x * 256) {
        int y, x;
        if (i < top) { y = i / W; x = i - y * W; }
        else if (i < top + bot) { const int j = i - top; y = pad_h + rh + j / W; x = j - (j / W) * W; }
        else { const int j = i - top - bot; const int r = j / side, q = j - r * side; y = pad_h + r; x = q < pad_w ? q : q + rw; }
        pre_store<MODE>(out, b, y, x, H, W, P, lut, lut16, m0, m1, m2);
    }
}

extern "C" int rtm3d_preprocess_batch(void* stream, int B, const uint8_t* const* h_imgs, const int* h_hw, const int* h_resized_hw,
                                      void* d_out, int out_mode, int H, int W, int out_border, const float* d_lut,
                                      const void* d_lut16, unsigned long long* d_sums) {
    if (B < 1 || !h_imgs || !h_hw || !d_out || !d_lut || !d_sums) { rt_set_error("preprocess_batch: bad arguments"); return 1; }
    if (out_mode != 0 && out_mode != 1) { rt_set_error("preprocess_batch: out_mode must be 0 (fp32 NCHW) or 1 (fp16 NHWC4)"); return 1; }
    if (out_mode == 1 && (!d_lut16 || out_border < 0)) { rt_set_error("preprocess_batch: the NHWC4 output needs the fp16 table and a border >= 0"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_sums, 0, (size_t)B * 3 * sizeof(unsigned long long), s) != hipSuccess) { rt_set_error("preprocess_batch: memset failed"); return 1; }
    for (int b0 = 0; b0 < B; b0 += PRE_MAX_BATCH) {
        const int nb = B - b0 < PRE_MAX_BATCH ? B - b0 : PRE_MAX_BATCH;
        PreBatch pb;
        int max_border = 0, max_rh = 0, max_rw = 0, max_w = 0;
        double max_scale = 0.0;
        for (int i = 0; i < nb; ++i) {
            const int h = h_hw[2 * (b0 + i)], w = h_hw[2 * (b0 + i) + 1];
            const int rh = h_resized_hw ? h_resized_hw[2 * (b0 + i)] : h, rw = h_resized_hw ? h_resized_hw[2 * (b0 + i) + 1] : w;
            if (!h_imgs[b0 + i] || h < 1 || w < 1 || rh < 1 || rw < 1 || rh > H || rw > W) {
                rt_set_error("preprocess_batch: image %d (%dx%d -> %dx%d) does not fit the %dx%d canvas", b0 + i, h, w, rh, rw, H, W);
                return 1;
            }
            pb.img[i] = h_imgs[b0 + i]; pb.h[i] = h; pb.w[i] = w; pb.rh[i] = rh; pb.rw[i] = rw;
            max_rh = rh > max_rh ? rh : max_rh; max_rw = rw > max_rw ? rw : max_rw; max_w = w > max_w ? w : max_w;
            max_scale = (double)h / rh > max_scale ? (double)h / rh : max_scale;
            max_border = H * W - rh * rw > max_border ? H * W - rh * rw : max_border;
        }
        // offsets of this sub-batch in the outputs
        void* o = out_mode == 0 ? (void*)((float*)d_out + (size_t)b0 * 3 * H * W)
                                : (void*)((f16*)d_out + (size_t)b0 * (H + 2 * out_border) * (W + 2 * out_border) * 4);
        unsigned long long* sm = d_sums + (size_t)b0 * 3;
        // bands of rows per workgroup: the stage holds (band_rows * scale + 2) source rows of the widest image; 48 KB of stage
        // leave room for two workgroups per CU.  About 8 workgroups per CU over the sub-batch.
        const int col_bytes = (int)(((size_t)max_rw * sizeof(int2) + 15) & ~(size_t)15);
        if (col_bytes > 60000) { rt_set_error("preprocess_batch: resized width %d exceeds the kernel's column table", max_rw); return 1; }
        int stage_bytes = 64 * 1024 - 8 * 1024 - col_bytes;            // total dynamic + static LDS stays under 64 KB
        stage_bytes = stage_bytes < 0 ? 0 : (stage_bytes > 48 * 1024 ? 48 * 1024 : stage_bytes);
        const size_t row_bytes = (size_t)max_w * 3;
        const int cap_rows = (int)((stage_bytes > 15 ? stage_bytes - 15 : 0) / (row_bytes ? row_bytes : 1));
        int band_rows = (int)((cap_rows - 2) / (max_scale > 1.0 ? max_scale : 1.0));
        band_rows = band_rows < 1 ? 1 : (band_rows > 16 ? 16 : band_rows);
        int bx = (2048 + nb - 1) / nb;
        const int bands = (max_rh + band_rows - 1) / band_rows;
        bx = bx > bands ? bands : bx;
        bx = bx < 1 ? 1 : bx;
        const size_t dyn = (size_t)col_bytes + stage_bytes;
        if (out_mode == 0) hipLaunchKernelGGL(pre_interior_kernel<0>, dim3(bx, nb), dim3(256), dyn, s, pb, o, H, W, out_border, d_lut, (const f16*)d_lut16, sm, band_rows, col_bytes, stage_bytes);
        else hipLaunchKernelGGL(pre_interior_kernel<1>, dim3(bx, nb), dim3(256), dyn, s, pb, o, H, W, out_border, d_lut, (const f16*)d_lut16, sm, band_rows, col_bytes, stage_bytes);
        if (max_border > 0) {
            int gx = (max_border + 255) / 256;
            gx = gx > 256 ? 256 : gx;
            if (out_mode == 0) hipLaunchKernelGGL(pre_border_kernel<0>, dim3(gx, nb), dim3(256), 0, s, pb, o, H, W, out_border, d_lut, (const f16*)d_lut16, sm);
            else hipLaunchKernelGGL(pre_border_kernel<1>, dim3(gx, nb), dim3(256), 0, s, pb, o, H, W, out_border, d_lut, (const f16*)d_lut16, sm);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt_set_error("preprocess_batch launch: %s", hipGetErrorString(e)); return 1; }
    return 0;
}
