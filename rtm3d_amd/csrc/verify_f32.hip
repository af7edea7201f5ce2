// fp32 VERIFICATION executor of the plan (SURVEY.md H2 regime ii): the same recorded op list - tap tables, channel
// slices, sub-pixel phases of the transposed convolutions, folded BatchNorm, composed 1x1 pairs - run on padded NHWC
// *fp32* tensors with fp32 weights and fp64 accumulation, one simple kernel per op kind.  It exists to separate "what the
// fp16 storage of the MFMA kernels costs" from "is the graph right": the device pipeline with these logits must land on
// the CPU reference's boxes (models/model.py:20-27 -> :29-75 -> utils/model_utils.py:264-312) to fp32 round-off.
// Not a fast path and never selected by Model.forward: ~2-4 TFLOP/s (uniform-address operand loads, no LDS, no MFMA).
#include "common.h"
#include "../../include/rtm3d_hip.h"

extern void rt_set_error(const char* fmt, ...);
#define VF_FAIL(...) do { rt_set_error(__VA_ARGS__); return 1; } while (0)

#define VC_PIX 4      // output pixels per thread (consecutive in the iteration domain)

struct VConvArgs {
    const float* in; const float* w; const float* bias; const float* res; float* out;
    int M, HmWm, Wm;
    int in_Hp, in_Wp, in_C, in_P, in_coff, in_stride;
    int out_Hp, out_Wp, out_C, out_P, out_coff, out_scale, out_oy, out_ox;
    int res_Hp, res_Wp, res_C, res_P, res_coff;
    int cin, cout, ntaps, relu, out_nchw, out_H, out_W;
    int tap_off[RT_MAX_TAPS];      // (dy * in_Wp + dx) * in_C, elements
};

// block = (64 output channels, 4 pixel groups); thread = one output channel x VC_PIX pixels; K loop over (tap, cin):
// the weight load is coalesced over the channel lanes, the pixel operand is one 16-byte load at a lane-uniform address
__global__ __launch_bounds__(256) void vconv_f32_kernel(VConvArgs a) {
    const int c = blockIdx.y * 64 + threadIdx.x;
    const int m0 = (blockIdx.x * 4 + threadIdx.y) * VC_PIX;
    if (m0 >= a.M) return;
    const bool cok = c < a.cout;
    const int cw = cok ? c : a.cout - 1;
    size_t ibase[VC_PIX];
    int n_[VC_PIX], y_[VC_PIX], x_[VC_PIX];
#pragma unroll
    for (int j = 0; j < VC_PIX; ++j) {
        const int m = m0 + j < a.M ? m0 + j : a.M - 1;
        const int n = m / a.HmWm, r = m - n * a.HmWm;
        const int y = r / a.Wm, x = r - y * a.Wm;
        n_[j] = n; y_[j] = y; x_[j] = x;
        ibase[j] = ((size_t)(n * a.in_Hp + y * a.in_stride + a.in_P) * a.in_Wp + x * a.in_stride + a.in_P) * a.in_C + a.in_coff;
    }
    double acc[VC_PIX] = {0.0, 0.0, 0.0, 0.0};
    for (int t = 0; t < a.ntaps; ++t) {
        const float* wt = a.w + (size_t)t * a.cin * a.cout + cw;
        const int toff = a.tap_off[t];
        for (int k = 0; k < a.cin; k += 4) {
            const float w0 = wt[(size_t)(k + 0) * a.cout], w1 = wt[(size_t)(k + 1) * a.cout];
            const float w2 = wt[(size_t)(k + 2) * a.cout], w3 = wt[(size_t)(k + 3) * a.cout];
#pragma unroll
            for (int j = 0; j < VC_PIX; ++j) {
                const f32x4 xv = *(const f32x4*)(a.in + (ptrdiff_t)ibase[j] + toff + k);
                acc[j] += (double)xv[0] * (double)w0;
                acc[j] += (double)xv[1] * (double)w1;
                acc[j] += (double)xv[2] * (double)w2;
                acc[j] += (double)xv[3] * (double)w3;
            }
        }
    }
    if (!cok) return;
    const float b = a.bias[c];
#pragma unroll
    for (int j = 0; j < VC_PIX; ++j) {
        if (m0 + j >= a.M) break;
        const int oy = y_[j] * a.out_scale + a.out_oy, ox = x_[j] * a.out_scale + a.out_ox;
        float v = (float)acc[j] + b;                    // fp32 from here on, as the reference's conv + BN epilogue
        if (a.res)
            v += a.res[((size_t)(n_[j] * a.res_Hp + oy + a.res_P) * a.res_Wp + ox + a.res_P) * a.res_C + a.res_coff + c];
        if (a.relu) v = v > 0.f ? v : 0.f;
        if (a.out_nchw)
            a.out[((size_t)(n_[j] * a.cout + c) * a.out_H + oy) * a.out_W + ox] = v;
        else
            a.out[((size_t)(n_[j] * a.out_Hp + oy + a.out_P) * a.out_Wp + ox + a.out_P) * a.out_C + a.out_coff + c] = v;
    }
}

static int vt_ok(const rtm3d_vtensor* t) { return t && t->d && t->Hp > 0 && t->Wp > 0 && t->C > 0 && t->P >= 0 && t->coff >= 0; }

extern "C" int rtm3d_verify_conv_f32(void* stream, const rtm3d_vconv_desc* d) {
    if (!d || !vt_ok(&d->in) || !d->d_w || !d->d_bias) VF_FAIL("verify_conv_f32: null argument");
    if (!d->out_nchw_f32 && !vt_ok(&d->out)) VF_FAIL("verify_conv_f32: bad output tensor");
    if (d->out_nchw_f32 && !d->out.d) VF_FAIL("verify_conv_f32: null NCHW output");
    if (d->B <= 0 || d->Hm <= 0 || d->Wm <= 0 || d->cin <= 0 || d->cout <= 0 || (d->cin & 3) || (d->in.C & 3) || (d->in.coff & 3))
        VF_FAIL("verify_conv_f32: bad shape (cin, channel pitch and channel offset of the input must be multiples of 4)");
    if (d->ntaps < 1 || d->ntaps > RTM3D_MAX_TAPS || d->in_stride < 1 || d->out_scale < 1) VF_FAIL("verify_conv_f32: bad taps / stride");
    if (d->in.coff + d->cin > d->in.C) VF_FAIL("verify_conv_f32: input channel slice out of range");
    if (!d->out_nchw_f32 && d->out.coff + d->cout > d->out.C) VF_FAIL("verify_conv_f32: output channel slice out of range");
    VConvArgs a;
    a.in = d->in.d; a.w = d->d_w; a.bias = d->d_bias; a.res = d->res.d; a.out = d->out.d;
    a.M = d->B * d->Hm * d->Wm; a.HmWm = d->Hm * d->Wm; a.Wm = d->Wm;
    a.in_Hp = d->in.Hp; a.in_Wp = d->in.Wp; a.in_C = d->in.C; a.in_P = d->in.P; a.in_coff = d->in.coff; a.in_stride = d->in_stride;
    a.out_Hp = d->out.Hp; a.out_Wp = d->out.Wp; a.out_C = d->out.C; a.out_P = d->out.P; a.out_coff = d->out.coff;
    a.out_scale = d->out_scale; a.out_oy = d->out_oy; a.out_ox = d->out_ox;
    a.res_Hp = d->res.Hp; a.res_Wp = d->res.Wp; a.res_C = d->res.C; a.res_P = d->res.P; a.res_coff = d->res.coff;
    a.cin = d->cin; a.cout = d->cout; a.ntaps = d->ntaps; a.relu = d->relu;
    a.out_nchw = d->out_nchw_f32; a.out_H = d->out_H; a.out_W = d->out_W;
    // every tap of every iteration-domain pixel must stay inside the padded input (the zero border is the padding)
    const int Hi = d->in.Hp - 2 * d->in.P, Wi = d->in.Wp - 2 * d->in.P;
    for (int t = 0; t < d->ntaps; ++t) {
        const int dy = d->tap_dy[t], dx = d->tap_dx[t];
        if (dy < -d->in.P || (d->Hm - 1) * d->in_stride + dy >= Hi + d->in.P || dx < -d->in.P || (d->Wm - 1) * d->in_stride + dx >= Wi + d->in.P)
            VF_FAIL("verify_conv_f32: tap (%d,%d) leaves the padded input", dy, dx);
        a.tap_off[t] = (dy * d->in.Wp + dx) * d->in.C;
    }
    if (d->out_nchw_f32) {
        if ((d->Hm - 1) * d->out_scale + d->out_oy >= d->out_H || (d->Wm - 1) * d->out_scale + d->out_ox >= d->out_W) VF_FAIL("verify_conv_f32: NCHW output too small");
    } else {
        const int Ho = d->out.Hp - 2 * d->out.P, Wo = d->out.Wp - 2 * d->out.P;
        if ((d->Hm - 1) * d->out_scale + d->out_oy >= Ho || (d->Wm - 1) * d->out_scale + d->out_ox >= Wo) VF_FAIL("verify_conv_f32: output tensor too small");
    }
    if (a.res) {
        if (!vt_ok(&d->res) || d->res.coff + d->cout > d->res.C) VF_FAIL("verify_conv_f32: bad residual tensor");
        const int Hr = d->res.Hp - 2 * d->res.P, Wr = d->res.Wp - 2 * d->res.P;
        if ((d->Hm - 1) * d->out_scale + d->out_oy >= Hr || (d->Wm - 1) * d->out_scale + d->out_ox >= Wr) VF_FAIL("verify_conv_f32: residual tensor too small");
    }
    dim3 block(64, 4, 1), grid((a.M + 4 * VC_PIX - 1) / (4 * VC_PIX), (d->cout + 63) / 64, 1);
    if (grid.y > 65535u) VF_FAIL("verify_conv_f32: too many output channels");
    hipLaunchKernelGGL(vconv_f32_kernel, grid, block, 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) VF_FAIL("verify_conv_f32 launch: %s", hipGetErrorString(e));
    return 0;
}

// ---------------------------------------------------------------------------------------------------- max pooling
struct VPoolArgs {
    const float* in; float* out;
    int in_Hp, in_Wp, in_C, in_P, in_coff, out_Hp, out_Wp, out_C, out_P, out_coff;
    int total, Ho, Wo, C, k, stride, pad;
};

// Window positions in the border read its zeros: as in the fp16 path (and include/rtm3d_hip.h, rtm3d_op_maxpool) the pooled
// maps are post-ReLU, so the zero border is the reference's -inf padding.
__global__ __launch_bounds__(256) void vpool_f32_kernel(VPoolArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const int c = i % a.C;
    int r = i / a.C;
    const int x = r % a.Wo; r /= a.Wo;
    const int y = r % a.Ho;
    const int n = r / a.Ho;
    float v = -INFINITY;
    for (int ky = 0; ky < a.k; ++ky)
        for (int kx = 0; kx < a.k; ++kx) {
            const int iy = y * a.stride - a.pad + ky + a.in_P, ix = x * a.stride - a.pad + kx + a.in_P;
            v = fmaxf(v, a.in[((size_t)(n * a.in_Hp + iy) * a.in_Wp + ix) * a.in_C + a.in_coff + c]);
        }
    a.out[((size_t)(n * a.out_Hp + y + a.out_P) * a.out_Wp + x + a.out_P) * a.out_C + a.out_coff + c] = v;
}

extern "C" int rtm3d_verify_maxpool_f32(void* stream, const rtm3d_vtensor* in, const rtm3d_vtensor* out, int B, int Ho, int Wo,
                                        int channels, int ksize, int stride, int pad) {
    if (!vt_ok(in) || !vt_ok(out) || B <= 0 || Ho <= 0 || Wo <= 0 || channels <= 0 || ksize < 1 || stride < 1 || pad < 0) VF_FAIL("verify_maxpool_f32: bad arguments");
    if (pad > in->P || (Ho - 1) * stride - pad + ksize - 1 + in->P >= in->Hp || (Wo - 1) * stride - pad + ksize - 1 + in->P >= in->Wp)
        VF_FAIL("verify_maxpool_f32: window leaves the padded input");
    if (Ho + 2 * out->P > out->Hp || Wo + 2 * out->P > out->Wp || in->coff + channels > in->C || out->coff + channels > out->C) VF_FAIL("verify_maxpool_f32: slice out of range");
    VPoolArgs a = {in->d, out->d, in->Hp, in->Wp, in->C, in->P, in->coff, out->Hp, out->Wp, out->C, out->P, out->coff,
                   B * Ho * Wo * channels, Ho, Wo, channels, ksize, stride, pad};
    hipLaunchKernelGGL(vpool_f32_kernel, dim3((a.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) VF_FAIL("verify_maxpool_f32 launch: %s", hipGetErrorString(e));
    return 0;
}

// ---------------------------------------------------------------------------------------------------- softmax fusion
struct VSoftArgs {
    const float* u[3]; int u_Hp[3], u_Wp[3], u_C[3], u_P[3], u_coff[3];
    const float* zin; float* zout;
    int zi_Hp, zi_Wp, zi_C, zi_P, zi_coff, zo_Hp, zo_Wp, zo_C, zo_P, zo_coff;
    int n_u, B, H, W, C;
    double* stats;          // [n_u][B][C][2] = (max, sum exp(u - max))
};

// one workgroup per (operand, image, 64-channel run): lanes = channels (coalesced), 4 row groups; max then sum of exp
__global__ __launch_bounds__(256) void vsoft_stats_kernel(VSoftArgs a) {
    __shared__ double red[4][64];
    const int cblocks = (a.C + 63) / 64;
    int bid = blockIdx.x;
    const int cb = bid % cblocks; bid /= cblocks;
    const int n = bid % a.B;
    const int i = bid / a.B;
    const int c = cb * 64 + threadIdx.x;
    const bool ok = c < a.C;
    const float* u = a.u[i];
    const int HW = a.H * a.W;
    float mx = -INFINITY;
    if (ok)
        for (int p = threadIdx.y; p < HW; p += 4) {
            const int y = p / a.W, x = p - y * a.W;
            mx = fmaxf(mx, u[((size_t)(n * a.u_Hp[i] + y + a.u_P[i]) * a.u_Wp[i] + x + a.u_P[i]) * a.u_C[i] + a.u_coff[i] + c]);
        }
    red[threadIdx.y][threadIdx.x] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf((float)red[0][threadIdx.x], (float)red[1][threadIdx.x]), fmaxf((float)red[2][threadIdx.x], (float)red[3][threadIdx.x]));
    __syncthreads();
    double s = 0.0;
    if (ok)
        for (int p = threadIdx.y; p < HW; p += 4) {
            const int y = p / a.W, x = p - y * a.W;
            s += (double)expf(u[((size_t)(n * a.u_Hp[i] + y + a.u_P[i]) * a.u_Wp[i] + x + a.u_P[i]) * a.u_C[i] + a.u_coff[i] + c] - mx);
        }
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && ok) {
        double* st = a.stats + ((size_t)(i * a.B + n) * a.C + c) * 2;
        st[0] = (double)mx;
        st[1] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// z_out = z_in + u_0 * softmax(u_0) + u_1 * softmax(u_1) + ...   in the plan's operand order, fp32 adds (keypoint_fpn_fusion.py:60-69)
__global__ __launch_bounds__(256) void vsoft_apply_kernel(VSoftArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)a.B * a.H * a.W * a.C;
    if (i >= total) return;
    const int c = (int)(i % a.C);
    size_t r = i / a.C;
    const int x = (int)(r % a.W); r /= a.W;
    const int y = (int)(r % a.H);
    const int n = (int)(r / a.H);
    float z = a.zin[((size_t)(n * a.zi_Hp + y + a.zi_P) * a.zi_Wp + x + a.zi_P) * a.zi_C + a.zi_coff + c];
    for (int k = 0; k < a.n_u; ++k) {
        const float u = a.u[k][((size_t)(n * a.u_Hp[k] + y + a.u_P[k]) * a.u_Wp[k] + x + a.u_P[k]) * a.u_C[k] + a.u_coff[k] + c];
        const double* st = a.stats + ((size_t)(k * a.B + n) * a.C + c) * 2;
        const float sm = (float)((double)expf(u - (float)st[0]) / st[1]);
        z += u * sm;
    }
    a.zout[((size_t)(n * a.zo_Hp + y + a.zo_P) * a.zo_Wp + x + a.zo_P) * a.zo_C + a.zo_coff + c] = z;
}

extern "C" size_t rtm3d_verify_softmax_workspace_bytes(int B, int C, int n_u) {
    return B > 0 && C > 0 && n_u > 0 ? (size_t)n_u * B * C * 2 * sizeof(double) : 0;
}

extern "C" int rtm3d_verify_softmax_fuse_f32(void* stream, const rtm3d_vtensor* z_in, const rtm3d_vtensor* z_out, int n_u,
                                             const rtm3d_vtensor* u, int B, int H, int W, int C, void* d_workspace) {
    if (!vt_ok(z_in) || !vt_ok(z_out) || !u || n_u < 1 || n_u > 3 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || !d_workspace) VF_FAIL("verify_softmax_fuse_f32: bad arguments");
    VSoftArgs a;
    for (int i = 0; i < 3; ++i) {
        const rtm3d_vtensor* t = &u[i < n_u ? i : 0];
        if (!vt_ok(t) || H + 2 * t->P > t->Hp || W + 2 * t->P > t->Wp || t->coff + C > t->C) VF_FAIL("verify_softmax_fuse_f32: bad operand %d", i);
        a.u[i] = t->d; a.u_Hp[i] = t->Hp; a.u_Wp[i] = t->Wp; a.u_C[i] = t->C; a.u_P[i] = t->P; a.u_coff[i] = t->coff;
    }
    if (H + 2 * z_in->P > z_in->Hp || W + 2 * z_in->P > z_in->Wp || z_in->coff + C > z_in->C ||
        H + 2 * z_out->P > z_out->Hp || W + 2 * z_out->P > z_out->Wp || z_out->coff + C > z_out->C) VF_FAIL("verify_softmax_fuse_f32: z slice out of range");
    a.zin = z_in->d; a.zout = z_out->d;
    a.zi_Hp = z_in->Hp; a.zi_Wp = z_in->Wp; a.zi_C = z_in->C; a.zi_P = z_in->P; a.zi_coff = z_in->coff;
    a.zo_Hp = z_out->Hp; a.zo_Wp = z_out->Wp; a.zo_C = z_out->C; a.zo_P = z_out->P; a.zo_coff = z_out->coff;
    a.n_u = n_u; a.B = B; a.H = H; a.W = W; a.C = C; a.stats = (double*)d_workspace;
    const int cblocks = (C + 63) / 64;
    hipLaunchKernelGGL(vsoft_stats_kernel, dim3(n_u * B * cblocks), dim3(64, 4, 1), 0, (hipStream_t)stream, a);
    const size_t total = (size_t)B * H * W * C;
    hipLaunchKernelGGL(vsoft_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) VF_FAIL("verify_softmax_fuse_f32 launch: %s", hipGetErrorString(e));
    return 0;
}
