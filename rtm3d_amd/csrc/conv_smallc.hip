// Register-direct MFMA convolution for the few-channel, high-resolution layers
// (DLA base_layer 3->16 7x7, level0 16->16, level1 16->32 s2, level2 entry 32->64; ResNet conv1 3->64).
//
// These layers are HBM-bound (72-124 FLOP/B): the goal is one coalesced pass over the input and the
// output, with the arithmetic off the critical path.  With <= 32 input channels a 16x16x32 MFMA
// K-step is "a few taps x all channels", and the 8 halves a lane feeds to the matrix core are 16
// contiguous bytes of one (shifted) input pixel - so the pixel operand is loaded straight from global
// memory into the MFMA operand registers (no LDS, no im2col), 16 consecutive pixels per instruction,
// and the whole filter bank lives in registers (20-72 VGPRs) for the lifetime of the wave.
//   CIN=16: K-step = taps (2s, 2s+1) x 16 ch      CIN=32: K-step = tap s x 32 ch
//   CIN=4 : K-step = filter row ky: 4 lane groups x (2 adjacent pixels x 4 ch)   [stem, image stored
//           as NHWC4 fp16 by nchw_to_nhwc4_kernel; the 4th channel and the 8th tap are zero-weighted]
// Weights are the MFMA A operand (rows = output channels), pixels the B operand: each lane ends up with
// 4 consecutive output channels of one pixel -> 8-byte NHWC stores, bias/ReLU fused.
#include "common.h"

struct __attribute__((aligned(8))) f16x8_a8 { f16 v[8]; };

template <int CIN, int NCT, int S, int TPW>
__global__ __launch_bounds__(256) void conv_smallc_kernel(const ConvKArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int frow = lane & 15, fk = lane >> 4;
    const ConvGroupArgs& g = a.g[0];
    const int ct0 = blockIdx.y * NCT;                       // first 16-channel tile of this block

    // the filter bank: [cout tile][k-step][lane][8 halves]
    f16x8 wf[NCT][S];
    {
        const f16* wp = a.wgt + g.w_off;
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int s = 0; s < S; ++s) wf[c][s] = *(const f16x8*)(wp + ((size_t)((ct0 + c) * S + s) * 64 + lane) * 8);
    }
    // per-lane element offset of its 16-byte piece in every k-step
    int koff[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (CIN == 16) { int t = 2 * s + (fk >> 1); t = t < a.ntaps ? t : a.ntaps - 1; koff[s] = g.tap_off[t] + (fk & 1) * 8; }
        else if (CIN == 32) koff[s] = g.tap_off[s] + fk * 8;
        else { const int kx = 2 * fk < 6 ? 2 * fk : 6; koff[s] = g.tap_off[s * 7 + kx]; }
    }
    const int ntiles = (a.M + 15) / 16;
    const float rcp_hw = 1.0f / (float)a.HmWm, rcp_w = 1.0f / (float)a.Wm;
    f32x4 bv[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) bv[c] = *(const f32x4*)(a.bias + g.bias_off + (ct0 + c) * 16 + fk * 4);
    const f16 lo = a.relu ? (f16)0.f : (f16)(-__builtin_inff());
    const f16x4 lo4 = {lo, lo, lo, lo};

    // pixel tile -> operand loads (rows past M re-read pixel M-1 and are not stored)
    auto locate = [&](int tile, int& n, int& y, int& x) {
        int m = tile * 16 + frow;
        m = m < a.M ? m : a.M - 1;
        n = div_small_q(m, a.HmWm, rcp_hw);
        const int rem = m - n * a.HmWm;
        y = div_small_q(rem, a.Wm, rcp_w);
        x = rem - y * a.Wm;
    };
    auto fetch = [&](int n, int y, int x, f16x8 (&xf)[S]) {
        const f16* ip = a.in + ((size_t)(n * a.in_Hp + y * a.in_stride + a.in_P) * a.in_Wp + x * a.in_stride + a.in_P) * a.in_C + g.in_coff;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (CIN == 4) {
                const f16x8_a8 t = *(const f16x8_a8*)(ip + koff[s]);
                __builtin_memcpy(&xf[s], &t, 16);
            } else {
                xf[s] = *(const f16x8*)(ip + koff[s]);
            }
        }
    };

    // software pipeline: the operands of tile it+1 are in flight while tile it is multiplied and stored
    const int tile0 = wave * TPW;
    if (tile0 >= ntiles) return;
    int n, y, x;
    f16x8 xf[S];
    locate(tile0, n, y, x);
    fetch(n, y, x, xf);
#pragma unroll 1
    for (int it = 0; it < TPW; ++it) {
        const int tile = tile0 + it;
        const bool more = it + 1 < TPW && tile + 1 < ntiles;
        int nn = n, ny = y, nx = x;
        f16x8 xn[S];
        if (more) {
            locate(tile + 1, nn, ny, nx);
            fetch(nn, ny, nx, xn);
        }
        f32x4 acc[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c][s], xf[s], acc[c], 0, 0, 0);
        f16x4 h[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const f32x4 v = acc[c] + bv[c];
            const f16x4 t = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            h[c] = __builtin_elementwise_max(t, lo4);
        }
        const bool valid = tile * 16 + frow < a.M;
        const int oy = y * a.out_scale + g.out_oy, ox = x * a.out_scale + g.out_ox;
        f16* op = (f16*)a.out + ((size_t)(n * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + g.out_coff + ct0 * 16;
        if (NCT % 2 == 0) {
            // v_permlane16_swap pairs two 16-channel tiles: a lane then owns 8 consecutive channels (16-byte store)
            const int so = (fk & 1) * 16 + (fk >> 1) * 8;
#pragma unroll
            for (int c = 0; c < NCT; c += 2) {
                uint32_t u0[2], u1[2];
                __builtin_memcpy(u0, &h[c], 8);
                __builtin_memcpy(u1, &h[c + 1 < NCT ? c + 1 : c], 8);
                const auto s0 = __builtin_amdgcn_permlane16_swap(u0[0], u1[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(u0[1], u1[1], false, false);
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                if (valid) *(u32x4*)(op + c * 16 + so) = o;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
                if (valid) *(f16x4*)(op + c * 16 + fk * 4) = h[c];
        }
        if (!more) break;
        n = nn; y = ny; x = nx;
#pragma unroll
        for (int s = 0; s < S; ++s) xf[s] = xn[s];
    }
}

// Vertical walk (stride-1 layers): a wave owns a 16-pixel-wide column strip and walks down R output rows.
// The K-steps are whole filter rows (CIN=4: one step per row; CIN=16: two steps per row, taps (ky,0),(ky,1)
// and (ky,2),zero), so moving down one output row keeps KR-1 of the KR operand rows in registers: one
// (CIN=16: two) 16-byte loads per lane and tile instead of 7 (5).  The loads of the next row are in flight
// while the current tile is multiplied and stored.
template <int CIN, int NCT, int KR, int SPR, int R>
__global__ __launch_bounds__(256) void conv_smallc_rows_kernel(const ConvKArgs a, const int tiles_x, const int strips_y, const int nwaves) {
    constexpr int S = KR * SPR;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const int frow = lane & 15, fk = lane >> 4;
    const ConvGroupArgs& g = a.g[0];
    const int ct0 = blockIdx.y * NCT;
    const int xb = wave % tiles_x, wq = wave / tiles_x;
    const int sy = wq % strips_y, n = wq / strips_y;
    const int H = a.HmWm / a.Wm, W = a.Wm;

    f16x8 wf[NCT][S];
    {
        const f16* wp = a.wgt + g.w_off;
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int s = 0; s < S; ++s) wf[c][s] = *(const f16x8*)(wp + ((size_t)((ct0 + c) * S + s) * 64 + lane) * 8);
    }
    // per-lane element offset of its 16-byte piece, per filter row and k-step of the row
    int koff[KR][SPR];
#pragma unroll
    for (int ky = 0; ky < KR; ++ky) {
        if (CIN == 4) {
            const int kx = 2 * fk < 6 ? 2 * fk : 6;
            koff[ky][0] = g.tap_off[ky * 7 + kx];
        } else {
            koff[ky][0] = g.tap_off[ky * 3 + (fk >> 1)] + (fk & 1) * 8;
            koff[ky][SPR - 1] = g.tap_off[ky * 3 + 2] + (fk & 1) * 8;        // lane groups 2,3 carry zero weights there
        }
    }
    f32x4 bv[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) bv[c] = *(const f32x4*)(a.bias + g.bias_off + (ct0 + c) * 16 + fk * 4);
    const f16 lo = a.relu ? (f16)0.f : (f16)(-__builtin_inff());
    const f16x4 lo4 = {lo, lo, lo, lo};

    const int y0 = sy * R, y1 = y0 + R < H ? y0 + R : H;
    int x = xb * 16 + frow;
    const bool xvalid = x < W;
    x = xvalid ? x : W - 1;
    const f16* ip = a.in + ((size_t)(n * a.in_Hp + y0 + a.in_P) * a.in_Wp + x + a.in_P) * a.in_C + g.in_coff;
    const size_t in_pitch = (size_t)a.in_Wp * a.in_C;
    f16* op = (f16*)a.out + ((size_t)(n * a.out_Hp + y0 * a.out_scale + g.out_oy + a.out_P) * a.out_Wp + x * a.out_scale + g.out_ox + a.out_P) * a.out_C
              + g.out_coff + ct0 * 16;
    const size_t out_pitch = (size_t)a.out_Wp * a.out_C * a.out_scale;

    auto fetch = [&](const f16* p, f16x8& dst) {
        if (CIN == 4) {
            const f16x8_a8 t = *(const f16x8_a8*)p;
            __builtin_memcpy(&dst, &t, 16);
        } else {
            dst = *(const f16x8*)p;
        }
    };
    f16x8 xr[KR][SPR];
#pragma unroll
    for (int ky = 0; ky < KR; ++ky)
#pragma unroll
        for (int sp = 0; sp < SPR; ++sp) fetch(ip + koff[ky][sp], xr[ky][sp]);

#pragma unroll 1
    for (int y = y0; y < y1; ++y) {
        f16x8 xn[SPR];
        const bool more = y + 1 < y1;
        if (more) {
#pragma unroll
            for (int sp = 0; sp < SPR; ++sp) fetch(ip + in_pitch + koff[KR - 1][sp], xn[sp]);
        }
        f32x4 acc[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < KR; ++ky)
#pragma unroll
            for (int sp = 0; sp < SPR; ++sp)
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c][ky * SPR + sp], xr[ky][sp], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const f32x4 v = acc[c] + bv[c];
            const f16x4 t = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            const f16x4 h = __builtin_elementwise_max(t, lo4);
            if (xvalid) *(f16x4*)(op + c * 16 + fk * 4) = h;
        }
        if (!more) break;
        ip += in_pitch;
        op += out_pitch;
#pragma unroll
        for (int ky = 0; ky + 1 < KR; ++ky)
#pragma unroll
            for (int sp = 0; sp < SPR; ++sp) xr[ky][sp] = xr[ky + 1][sp];
#pragma unroll
        for (int sp = 0; sp < SPR; ++sp) xr[KR - 1][sp] = xn[sp];
    }
}

// fp32 NCHW (B,3,H,W) image -> padded NHWC4 fp16 (4th channel = 0): the stem's operand layout.
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ in, f16* __restrict__ out, int B, int H, int W,
                                                            int Hp, int Wp, int P) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * H * W) return;
    const int n = div_small_q(idx, H * W, 1.0f / (float)(H * W)), rem = idx - n * (H * W);
    const int y = div_small_q(rem, W, 1.0f / (float)W), x = rem - y * W;
    const size_t plane = (size_t)H * W;
    const float* p = in + (size_t)n * 3 * plane + (size_t)y * W + x;
    f16x4 v = {(f16)p[0], (f16)p[plane], (f16)p[2 * plane], (f16)0.f};
    *(f16x4*)(out + ((size_t)(n * Hp + y + P) * Wp + x + P) * 4) = v;
}

bool conv_smallc_supported(int cin, int cout, int ntaps) {
    if (cin == 16 && ntaps == 9) return cout == 16 || cout == 32;
    if (cin == 32 && (ntaps == 9 || ntaps == 1)) return cout == 64;
    if (cin == 4 && ntaps == 49) return cout == 16 || cout == 64;
    return false;
}

#define LAUNCH(CIN, NCT, S, TPW, GY)                                                                         \
    do {                                                                                                     \
        const int ntiles = (a.M + 15) / 16;                                                                  \
        dim3 grid((ntiles + 4 * (TPW) - 1) / (4 * (TPW)), (GY), 1), block(256);                              \
        hipLaunchKernelGGL((conv_smallc_kernel<CIN, NCT, S, TPW>), grid, block, 0, s, a);                    \
    } while (0)

#define LAUNCH_ROWS(CIN, NCT, KR, SPR, R, GY)                                                                 \
    do {                                                                                                     \
        const int H = a.HmWm / a.Wm, B = a.M / a.HmWm;                                                       \
        const int tiles_x = (a.Wm + 15) / 16, strips_y = (H + (R) - 1) / (R);                                \
        const int nwaves = B * strips_y * tiles_x;                                                           \
        dim3 grid((nwaves + 3) / 4, (GY), 1), block(256);                                                    \
        hipLaunchKernelGGL((conv_smallc_rows_kernel<CIN, NCT, KR, SPR, R>), grid, block, 0, s, a, tiles_x, strips_y, nwaves); \
    } while (0)

hipError_t launch_conv_smallc(const ConvKArgs& a, hipStream_t s) {
    if (a.in_stride == 1 && a.out_scale == 1 && a.cin == 4 && a.ntaps == 49 && a.cout == 16) { LAUNCH_ROWS(4, 1, 7, 1, 32, 1); return hipGetLastError(); }
    if (a.in_stride == 1 && a.out_scale == 1 && a.cin == 16 && a.ntaps == 9 && a.cout == 16 && a.ksteps == 6) { LAUNCH_ROWS(16, 1, 3, 2, 32, 1); return hipGetLastError(); }
    if (a.cin == 16 && a.ntaps == 9 && a.cout == 16) LAUNCH(16, 1, 5, 8, 1);
    else if (a.cin == 16 && a.ntaps == 9 && a.cout == 32) LAUNCH(16, 2, 5, 8, 1);
    else if (a.cin == 32 && a.ntaps == 9 && a.cout == 64) LAUNCH(32, 2, 9, 8, 2);
    else if (a.cin == 32 && a.ntaps == 1 && a.cout == 64) LAUNCH(32, 4, 1, 8, 1);
    else if (a.cin == 4 && a.ntaps == 49 && a.cout == 16) LAUNCH(4, 1, 7, 8, 1);
    else if (a.cin == 4 && a.ntaps == 49 && a.cout == 64) LAUNCH(4, 4, 7, 8, 1);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_nchw_to_nhwc4(const float* in, f16* out, int B, int H, int W, int Hp, int Wp, int P, hipStream_t s) {
    const int total = B * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3((total + 255) / 256), dim3(256), 0, s, in, out, B, H, W, Hp, Wp, P);
    return hipGetLastError();
}
