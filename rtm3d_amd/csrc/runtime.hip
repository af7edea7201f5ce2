// Runtime of librtm3d_hip.so: context (activation workspace + packed weights + launch plan) and
// the C ABI declared in include/rtm3d_hip.h.  The plan is recorded once by the host
// (rtm3d_amd/plan.py) and replayed by rtm3d_forward on the caller's HIP stream.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "common.h"
#include "../../include/rtm3d_hip.h"

static thread_local char g_err[512] = "";
void rt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define RT_FAIL(...) do { rt_set_error(__VA_ARGS__); return 1; } while (0)
#define RT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rt_set_error("%s: %s", #expr, hipGetErrorString(e_)); return 1; } } while (0)

struct Tensor {
    int B, H, W, C, P, Hp, Wp;
    f16* alloc;     // allocation start (guard band in front)
    f16* base;      // padded element [0][0][0][0]
    size_t elems;   // padded elements (without guards)
};

enum OpKind { OP_CONV_MFMA, OP_CONV_MFMA256, OP_CONV64_HALO, OP_CONV64_ROOT, OP_CONV64S2_HALO, OP_CONV128_HALO, OP_STEM_FUSED, OP_CONV32S2_FUSED, OP_CONV_SMALLC, OP_INPUT4, OP_HEADOUT, OP_MAXPOOL, OP_SOFTMAX, OP_PATCH_MASK };

struct Op {
    OpKind kind;
    std::string name;
    double flops, bytes;
    ConvKArgs conv; int groups, bn_tile, epi_nchw, out_slot, ksize;
    int ticket_slot = -1;           // conv64_halo: index of this op's ticket counter (ctx->tile_ctr + 8 + slot)
    float* stat_out = nullptr;      // softmax partials written by this conv's epilogue (halo kernel), or null
    StemKArgs stem; int stem_cout;
    PoolKArgs pool;
    SoftmaxKArgs sm;
    HeadOutArgs ho;
    StemFusedArgs sf;
    Conv32S2Args c32;
    PatchMaskArgs pm;
    RootKArgs root;
};

struct rtm3d_ctx {
    int device;
    int n_cus = 256;
    std::vector<Tensor> tensors;
    std::vector<void*> blobs;
    std::vector<size_t> blob_bytes;
    std::vector<Op> ops;
    std::vector<void*> extra;   // other device allocations (softmax partials)
    // spatial-softmax partials emitted by conv epilogues for the next softmax op (softmax_stat_slot)
    float* stat_buf = nullptr;
    int stat_chunks = 0, stat_B = 0;
    int stat_tensor[3] = {-1, -1, -1};
    unsigned int* tile_ctr = nullptr;   // [0,8): per-XCD ticket counters of the persistent conv256 kernels (self-resetting);
                                        // [8, 8+TICKET_SLOTS): one counter per conv64_halo op; all zeroed at the head of every forward
    int ticket_slots_used = 0;
    float* slab = nullptr;             // split-K partial sums (shared by all split ops of the plan: they run one after the other)
    size_t slab_floats = 0;
    // live probe: hipEvent pairs around one op of every replay (bench.py roofline)
    int probe_op = -1;
    std::vector<hipEvent_t> probe_ev;   // 2 * PROBE_RING events
    int probe_count = 0;
    // hipGraph replay (rtm3d_ctx_set_graph): one instantiated graph per distinct (input, 4 x output) pointer tuple
    int graph_mode = 0;
    hipStream_t capture_stream = nullptr;
    // a context owns ONE activation workspace and one set of ticket counters: replays must not overlap.  Calls on the same
    // stream are ordered by the stream; a call on a different stream first waits for the previous replay (done_ev)
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    hipEvent_t done_ev = nullptr;
    struct GraphEntry { const void* key[5]; hipGraph_t graph; hipGraphExec_t exec; unsigned long long last_use; };
    std::vector<GraphEntry> graphs;
    unsigned long long graph_clock = 0, graph_hits = 0, graph_captures = 0, graph_refused = 0;
    int test_memset_in_replay = 0;     // rtm3d_ctx_debug_memset_in_replay: a hipMemsetAsync in front of every replay (tests the guard above)
    // stage marks (rtm3d_forward_marks): events on the caller's stream in front of given ops of ONE replay, and one behind the last op
    int n_marks = 0;
    int mark_op[8];
    hipEvent_t mark_ev[9];
};

static const int PROBE_RING = 64;
static const int TICKET_SLOTS = 56;
static const int SPLIT_CTRS = 256;       // per-tile arrival counters of the split-K convolutions (self-resetting, shared by all ops)
static const size_t TILE_CTR_WORDS = 8 + TICKET_SLOTS + SPLIT_CTRS;
static const size_t DEBUG_WORD0 = 1024;           // diagnostic builds (-DC256_STAMPS) dump in-kernel stamps behind the counters
static const size_t DEBUG_WORDS = 16384;

static int ensure_tile_ctr(rtm3d_ctx* ctx) {
    if (ctx->tile_ctr) return 0;
    RT_HIP(hipMalloc((void**)&ctx->tile_ctr, (DEBUG_WORD0 + DEBUG_WORDS) * sizeof(unsigned int)));
    RT_HIP(hipMemset(ctx->tile_ctr, 0, (DEBUG_WORD0 + DEBUG_WORDS) * sizeof(unsigned int)));
    ctx->extra.push_back(ctx->tile_ctr);
    return 0;
}

extern "C" const char* rtm3d_last_error(void) { return g_err; }
extern "C" int rtm3d_abi_version(void) { return RTM3D_ABI_VERSION; }

extern "C" int rtm3d_ctx_create(int device, rtm3d_ctx** out) {
    if (!out) RT_FAIL("ctx_create: null out");
    RT_HIP(hipSetDevice(device));
    // The persistent convolution kernels deal workgroup b to XCD b & 7 and size their grids as CUs / 8 per XCD
    // (conv_mfma256.hip): they are written for the 8-XCD, 256-CU SPX partition of an MI355X.  Another partition
    // mode would still compute correctly (tickets are per counter, not per physical XCD) but the L2 locality the
    // tile order is built for would be gone, so it is refused instead of silently running slow.
    int xccs = 0, cus = 0;
    RT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, device) != hipSuccess) xccs = 0;
    if ((xccs != 0 && xccs != 8) || cus < 8 || (cus % 8) != 0)
        RT_FAIL("ctx_create: device %d reports %d XCCs / %d CUs; librtm3d_hip is built for the 8-XCD (SPX) MI355X", device, xccs, cus);
    rtm3d_ctx* c = new rtm3d_ctx();
    c->device = device;
    c->n_cus = cus;
    *out = c;
    return 0;
}

extern "C" void rtm3d_ctx_destroy(rtm3d_ctx* ctx) {
    if (!ctx) return;
    for (auto& t : ctx->tensors) (void)hipFree(t.alloc);
    for (auto p : ctx->blobs) (void)hipFree(p);
    for (auto p : ctx->extra) (void)hipFree(p);
    for (auto e : ctx->probe_ev) (void)hipEventDestroy(e);
    for (auto& ge : ctx->graphs) { (void)hipGraphExecDestroy(ge.exec); (void)hipGraphDestroy(ge.graph); }
    if (ctx->capture_stream) (void)hipStreamDestroy(ctx->capture_stream);
    if (ctx->done_ev) (void)hipEventDestroy(ctx->done_ev);
    delete ctx;
}

static const size_t GUARD = 4096;   // halves of slack on both sides of every activation buffer

extern "C" int rtm3d_tensor_create(rtm3d_ctx* ctx, int B, int H, int W, int C, int pad, int* id) {
    if (!ctx || !id) RT_FAIL("tensor_create: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || pad < 0 || (C % 4) != 0) RT_FAIL("tensor_create: bad shape B=%d H=%d W=%d C=%d pad=%d (C must be a multiple of 4)", B, H, W, C, pad);
    Tensor t;
    t.B = B; t.H = H; t.W = W; t.C = C; t.P = pad; t.Hp = H + 2 * pad; t.Wp = W + 2 * pad;
    t.elems = (size_t)B * t.Hp * t.Wp * C;
    if (t.elems >= ((size_t)1 << 32) - 2 * GUARD) RT_FAIL("tensor_create: %zu elements exceed the 32-bit offset range of the kernels", t.elems);
    const size_t bytes = (t.elems + 2 * GUARD) * sizeof(f16);
    RT_HIP(hipMalloc((void**)&t.alloc, bytes));
    RT_HIP(hipMemset(t.alloc, 0, bytes));
    t.base = t.alloc + GUARD;
    ctx->tensors.push_back(t);
    *id = (int)ctx->tensors.size() - 1;
    return 0;
}

static Tensor* get_tensor(rtm3d_ctx* ctx, int id) {
    if (id < 0 || id >= (int)ctx->tensors.size()) return nullptr;
    return &ctx->tensors[id];
}

extern "C" int rtm3d_tensor_download(rtm3d_ctx* ctx, int id, int c0, int C, float* h_nchw) {
    Tensor* t = ctx ? get_tensor(ctx, id) : nullptr;
    if (!t || !h_nchw || c0 < 0 || C <= 0 || c0 + C > t->C) RT_FAIL("tensor_download: bad arguments");
    std::vector<f16> host(t->elems);
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(host.data(), t->base, t->elems * sizeof(f16), hipMemcpyDeviceToHost));
    for (int n = 0; n < t->B; ++n)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < t->H; ++y)
                for (int x = 0; x < t->W; ++x)
                    h_nchw[(((size_t)n * C + c) * t->H + y) * t->W + x] =
                        (float)host[(((size_t)n * t->Hp + y + t->P) * t->Wp + x + t->P) * t->C + c0 + c];
    return 0;
}

extern "C" int rtm3d_tensor_upload(rtm3d_ctx* ctx, int id, int c0, int C, const float* h_nchw) {
    Tensor* t = ctx ? get_tensor(ctx, id) : nullptr;
    if (!t || !h_nchw || c0 < 0 || C <= 0 || c0 + C > t->C) RT_FAIL("tensor_upload: bad arguments");
    std::vector<f16> host(t->elems);
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(host.data(), t->base, t->elems * sizeof(f16), hipMemcpyDeviceToHost));
    for (int n = 0; n < t->B; ++n)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < t->H; ++y)
                for (int x = 0; x < t->W; ++x)
                    host[(((size_t)n * t->Hp + y + t->P) * t->Wp + x + t->P) * t->C + c0 + c] =
                        (f16)h_nchw[(((size_t)n * C + c) * t->H + y) * t->W + x];
    RT_HIP(hipMemcpy(t->base, host.data(), t->elems * sizeof(f16), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int rtm3d_blob_create(rtm3d_ctx* ctx, const void* h_data, size_t bytes, int* id) {
    if (!ctx || !h_data || !id || bytes == 0) RT_FAIL("blob_create: bad arguments");
    void* p = nullptr;
    RT_HIP(hipMalloc(&p, bytes + 256));
    RT_HIP(hipMemset(p, 0, bytes + 256));
    RT_HIP(hipMemcpy(p, h_data, bytes, hipMemcpyHostToDevice));
    ctx->blobs.push_back(p);
    ctx->blob_bytes.push_back(bytes);
    *id = (int)ctx->blobs.size() - 1;
    return 0;
}

extern "C" int rtm3d_tensor_info(rtm3d_ctx* ctx, int id, void** d_base, int* B, int* H, int* W, int* C, int* border) {
    Tensor* t = ctx ? get_tensor(ctx, id) : nullptr;
    if (!t) RT_FAIL("tensor_info: bad tensor %d", id);
    if (d_base) *d_base = t->base;
    if (B) *B = t->B;
    if (H) *H = t->H;
    if (W) *W = t->W;
    if (C) *C = t->C;
    if (border) *border = t->P;
    return 0;
}

static void* get_blob(rtm3d_ctx* ctx, int id, size_t* bytes);
extern "C" int rtm3d_blob_address(rtm3d_ctx* ctx, int id, void** d_ptr, size_t* bytes) {
    void* p = ctx ? get_blob(ctx, id, bytes) : nullptr;
    if (!p || !d_ptr) RT_FAIL("blob_address: bad blob %d", id);
    *d_ptr = p;
    return 0;
}

static void* get_blob(rtm3d_ctx* ctx, int id, size_t* bytes) {
    if (id < 0 || id >= (int)ctx->blobs.size()) return nullptr;
    if (bytes) *bytes = ctx->blob_bytes[id];
    return ctx->blobs[id];
}

extern "C" int rtm3d_op_conv(rtm3d_ctx* ctx, const rtm3d_conv_desc* d) {
    if (!ctx || !d) RT_FAIL("op_conv: null argument");
    Tensor* in = get_tensor(ctx, d->in_tensor);
    // rtm3d_conv_desc.s2d_tensor is tensor id + 1 (0 = none: a zero-initialised descriptor asks for no copy)
    if (d->s2d_tensor < 0) RT_FAIL("op_conv: s2d_tensor = %d (tensor id + 1, 0 = none)", d->s2d_tensor);
    const int s2d_id = d->s2d_tensor - 1;
    // (out_tensor < 0 together with a space-to-depth tensor: only the space-to-depth copy of the output is written)
    const bool s2d_only = !d->out_nchw_f32 && d->out_tensor < 0 && s2d_id >= 0;
    Tensor* out = (d->out_nchw_f32 || s2d_only) ? nullptr : get_tensor(ctx, d->out_tensor);
    Tensor* res = d->res_tensor >= 0 ? get_tensor(ctx, d->res_tensor) : nullptr;
    if (!in) RT_FAIL("op_conv: bad input tensor %d", d->in_tensor);
    if (!d->out_nchw_f32 && !s2d_only && !out) RT_FAIL("op_conv: bad output tensor %d", d->out_tensor);
    if (d->res_tensor >= 0 && !res) RT_FAIL("op_conv: bad residual tensor %d", d->res_tensor);
    if (d->groups < 1 || d->groups > RT_MAX_GROUPS || d->ntaps < 1 || d->ntaps > RT_MAX_TAPS) RT_FAIL("op_conv: groups/ntaps out of range");
    if (d->out_nchw_f32 < 0 || d->out_nchw_f32 > 4) RT_FAIL("op_conv: out_nchw_f32 out of range");
    size_t wbytes = 0, bbytes = 0;
    const f16* w = (const f16*)get_blob(ctx, d->w_blob, &wbytes);
    const float* bias = (const float*)get_blob(ctx, d->bias_blob, &bbytes);
    if (!w || !bias) RT_FAIL("op_conv: bad weight/bias blob");

    Op op;
    ConvKArgs& a = op.conv;
    memset(&a, 0, sizeof(a));
    a.in = in->base; a.wgt = w; a.bias = bias; a.res = res ? res->base : nullptr; a.out = out ? (void*)out->base : nullptr;
    a.HmWm = d->Hm * d->Wm; a.Wm = d->Wm; a.M = in->B * a.HmWm;
    a.in_Hp = in->Hp; a.in_Wp = in->Wp; a.in_C = in->C; a.in_stride = d->in_stride; a.in_P = in->P;
    a.out_scale = d->out_scale;
    if (out) { a.out_Hp = out->Hp; a.out_Wp = out->Wp; a.out_C = out->C; a.out_P = out->P; }
    else { a.out_C = d->cout * d->groups; a.out_H = d->out_H; a.out_W = d->out_W; }
    if (res) { a.res_Hp = res->Hp; a.res_Wp = res->Wp; a.res_C = res->C; a.res_P = res->P; }
    a.cin = d->cin; a.cout = d->cout; a.ntaps = d->ntaps; a.relu = d->relu;
    // bounds: every tap of every iteration pixel must stay inside the padded input
    for (int g = 0; g < d->groups; ++g) {
        for (int t = 0; t < d->ntaps; ++t) {
            const int dy = d->tap_dy[g][t], dx = d->tap_dx[g][t];
            const int ylo = dy, yhi = (d->Hm - 1) * d->in_stride + dy, xlo = dx, xhi = (d->Wm - 1) * d->in_stride + dx;
            // (in_s2d: in_tensor is the half-resolution space-to-depth copy of the map the taps are stated on; its one-pixel border
            // stands for a two-pixel border of that map)
            const int Hin = d->in_s2d ? 2 * in->H : in->H, Win = d->in_s2d ? 2 * in->W : in->W, Pin = d->in_s2d ? (in->P >= 1 ? 2 : 0) : in->P;
            if (ylo < -Pin || xlo < -Pin || yhi >= Hin + Pin || xhi >= Win + Pin)
                RT_FAIL("op_conv: tap (%d,%d) leaves the padded input (H=%d W=%d pad=%d, Hm=%d Wm=%d stride=%d)", dy, dx, in->H, in->W, in->P, d->Hm, d->Wm, d->in_stride);
            const int dc = d->tap_dc[g][t];
            if (dc && d->kernel != 0 && d->kernel != 2) RT_FAIL("op_conv: per-tap channel offsets need kernel 0 or 2");
            if ((dc % 8) || d->in_coff[g] + dc < 0 || d->in_coff[g] + dc + d->cin > in->C) RT_FAIL("op_conv: tap %d names channels [%d, %d) outside the %d-channel input tensor", t, d->in_coff[g] + dc, d->in_coff[g] + dc + d->cin, in->C);
            a.g[g].tap_off[t] = (dy * in->Wp + dx) * in->C + dc;
        }
        if (d->in_coff[g] < 0 || d->in_coff[g] + d->cin > in->C || (d->in_coff[g] % 8)) RT_FAIL("op_conv: input channel slice out of range");
        const int oyhi = (d->Hm - 1) * d->out_scale + d->out_oy[g], oxhi = (d->Wm - 1) * d->out_scale + d->out_ox[g];
        if (out) {
            if (d->out_coff[g] < 0 || d->out_coff[g] + d->cout > out->C || (d->out_coff[g] % 8)) RT_FAIL("op_conv: output channel slice out of range");
            if (oyhi >= out->H || oxhi >= out->W || d->out_oy[g] < 0 || d->out_ox[g] < 0) RT_FAIL("op_conv: output pixel out of range");
            if (out->B != in->B) RT_FAIL("op_conv: batch mismatch");
        } else if (!s2d_only) {
            if (oyhi >= d->out_H || oxhi >= d->out_W) RT_FAIL("op_conv: NCHW output pixel out of range");
        }
        if (res) {
            if (d->res_coff[g] < 0 || d->res_coff[g] + d->cout > res->C || (d->res_coff[g] % 8)) RT_FAIL("op_conv: residual channel slice out of range");
            if (oyhi >= res->H || oxhi >= res->W) RT_FAIL("op_conv: residual pixel out of range");
            // a residual must not alias the output ("in place"): partial pixel tiles are padded with copies of their last pixel,
            // and a copy that loads the residual after another copy has stored would add it twice
            if (out && res == out && d->res_coff[g] < d->out_coff[g] + d->cout && d->out_coff[g] < d->res_coff[g] + d->cout)
                RT_FAIL("op_conv: the residual channel slice [%d,%d) overlaps the output slice [%d,%d) of the same tensor", d->res_coff[g], d->res_coff[g] + d->cout, d->out_coff[g], d->out_coff[g] + d->cout);
        }
        a.g[g].in_coff = d->in_coff[g]; a.g[g].out_coff = d->out_coff[g]; a.g[g].res_coff = d->res_coff[g];
        a.g[g].out_oy = d->out_oy[g]; a.g[g].out_ox = d->out_ox[g];
    }
    if (s2d_id >= 0) {
        Tensor* s2 = get_tensor(ctx, s2d_id);
        if (!s2 || (!out && !s2d_only)) RT_FAIL("op_conv: bad space-to-depth tensor %d", s2d_id);
        if (s2d_only && d->kernel != 0) RT_FAIL("op_conv: writing ONLY the space-to-depth copy needs kernel 0");
        if ((d->kernel != 0 && d->kernel != 5) || d->groups != 1 || d->out_scale != 1 || d->out_oy[0] || d->out_ox[0] || (d->Hm & 1) || (d->Wm & 1) || (d->cout % 8))
            RT_FAIL("op_conv: the space-to-depth copy needs kernel 0 or 5, one group, out_scale 1 and an even output height / width");
        if (s2->B != in->B || s2->H * 2 != d->Hm || s2->W * 2 != d->Wm || d->s2d_coff < 0 || d->s2d_coff + 4 * d->cout > s2->C || (d->s2d_coff % 8))
            RT_FAIL("op_conv: space-to-depth slice mismatch (half resolution, 4 x cout channels)");
        if (s2 == out || s2 == in || s2 == res) RT_FAIL("op_conv: the space-to-depth copy aliases an operand");
        a.s2d = s2->base; a.s_Hp = s2->Hp; a.s_Wp = s2->Wp; a.s_C = s2->C; a.s_P = s2->P; a.s_coff = d->s2d_coff;
    }
    if (d->in_s2d) {
        if (d->kernel != 7 || d->groups != 1) RT_FAIL("op_conv: a space-to-depth INPUT is read by kernel 7 only");
        if (d->in_coff[0] + 4 * d->cin > in->C || in->P < 1) RT_FAIL("op_conv: the space-to-depth input needs 4 x cin channels behind in_coff and a border >= 1");
    }
    op.groups = d->groups; op.epi_nchw = d->out_nchw_f32 ? 1 : 0; op.out_slot = d->out_nchw_f32 - 1;
    int stat_slot = -1;
    const double M = (double)a.M;
    op.flops = 2.0 * M * d->groups * (double)d->cin * d->ntaps * d->cout;
    op.bytes = 2.0 * M * d->groups * (d->cin + d->cout * (d->out_nchw_f32 ? 2 : 1)) + (res ? 2.0 * M * d->groups * d->cout : 0.0)
               + (s2d_id >= 0 && !s2d_only ? 2.0 * M * d->cout : 0.0);
    if (d->kernel == 2) {
        if (d->cin % 64 || d->cout % 256 || d->out_nchw_f32) RT_FAIL("op_conv(mfma256): needs cin %% 64 == 0, cout %% 256 == 0, NHWC output (cin=%d cout=%d)", d->cin, d->cout);
        a.cpt = d->cin / 64; a.ksteps = d->ntaps * a.cpt;
        a.MT = (a.M + 255) / 256; a.NT = d->cout / 256;
        const size_t per_group = (size_t)d->cout * a.ksteps * 64;
        if (wbytes != per_group * d->groups * sizeof(f16)) RT_FAIL("op_conv(mfma256): packed weight blob has %zu bytes, expected %zu", wbytes, per_group * d->groups * sizeof(f16));
        if (bbytes != (size_t)d->cout * d->groups * sizeof(float)) RT_FAIL("op_conv(mfma256): bias blob size mismatch");
        for (int g = 0; g < d->groups; ++g) { a.g[g].w_off = (uint32_t)(per_group * g); a.g[g].bias_off = d->cout * g; }
        op.kind = OP_CONV_MFMA256; op.bn_tile = 256;
        stat_slot = d->softmax_stat_slot;
        if (ensure_tile_ctr(ctx)) return 1;
        op.name = d->ntaps == 1 ? "conv1x1_mfma256" : (d->ntaps == 4 ? "deconv4x4_phase_mfma256" : (conv_mfma256_uses_lattice(a, d->groups) ? "conv3x3_mfma256_lattice" : "conv3x3_mfma256"));
    } else if (d->kernel == 5) {
        // 64 -> 64 channel 3x3 halo kernel with the filter bank in registers (conv64_halo.hip)
        if (d->out_nchw_f32) RT_FAIL("op_conv(conv64): NCHW output unsupported");
        a.cpt = 1; a.ksteps = 9; a.MT = 0; a.NT = 1;
        if (!conv64_halo_supported(a, d->groups)) RT_FAIL("op_conv(conv64): needs one 64->64 3x3 stride-1 conv on a map with W %% 32 == 0, H %% 8 == 0");
        if (wbytes != (size_t)9 * 64 * 64 * sizeof(f16) || bbytes != 64 * sizeof(float)) RT_FAIL("op_conv(conv64): weight/bias blob size mismatch");
        if (ctx->ticket_slots_used >= TICKET_SLOTS) RT_FAIL("op_conv(conv64): out of ticket counters");
        if (ensure_tile_ctr(ctx)) return 1;
        a.g[0].w_off = 0; a.g[0].bias_off = 0;
        op.kind = OP_CONV64_HALO; op.bn_tile = 64; op.ticket_slot = ctx->ticket_slots_used++;
        op.name = "conv3x3_c64_halo";
    } else if (d->kernel == 7) {
        // 64 -> 128 channel 3x3 STRIDE-2 halo kernel with the filter bank in registers (conv64s2_halo.hip)
        if (d->out_nchw_f32 || res) RT_FAIL("op_conv(conv64s2): NCHW output / residual unsupported");
        a.cpt = 1; a.ksteps = 9; a.MT = 0; a.NT = 1;
        a.in_s2d = d->in_s2d ? 1 : 0;
        if (!conv64s2_halo_supported(a, d->groups)) RT_FAIL("op_conv(conv64s2): needs one 64->128 3x3 stride-2 conv onto a map with W %% 32 == 0, H %% 4 == 0 (input border >= 1)");
        if (wbytes != (size_t)9 * 64 * 128 * sizeof(f16) || bbytes != 128 * sizeof(float)) RT_FAIL("op_conv(conv64s2): weight/bias blob size mismatch");
        if (ctx->ticket_slots_used >= TICKET_SLOTS) RT_FAIL("op_conv(conv64s2): out of ticket counters");
        if (ensure_tile_ctr(ctx)) return 1;
        a.g[0].w_off = 0; a.g[0].bias_off = 0;
        op.kind = OP_CONV64S2_HALO; op.bn_tile = 128; op.ticket_slot = ctx->ticket_slots_used++;
        op.name = "conv3x3s2_c64_halo";
    } else if (d->kernel == 6) {
        // 3x3 halo kernel for multiples of 128 channels, weights streamed through an LDS ring (conv128_halo.hip); the weight
        // blob is the generic kernel's packing for 128-channel tiles
        if (d->out_nchw_f32) RT_FAIL("op_conv(conv128): NCHW output unsupported");
        a.cpt = d->cin / 64; a.ksteps = 9 * a.cpt; a.MT = 0; a.NT = 1;
        if (!conv128_halo_supported(a, d->groups)) RT_FAIL("op_conv(conv128): needs one 3x3 stride-1 conv with cin, cout %% 128 == 0 on a map with W %% 32 == 0, H %% 8 == 0");
        if (d->bn_tile != 128 || wbytes != (size_t)9 * d->cin * d->cout * sizeof(f16) || bbytes != (size_t)d->cout * sizeof(float))
            RT_FAIL("op_conv(conv128): weight/bias blob size mismatch (expects the bn_tile = 128 packing)");
        if (ctx->ticket_slots_used >= TICKET_SLOTS) RT_FAIL("op_conv(conv128): out of ticket counters");
        if (ensure_tile_ctr(ctx)) return 1;
        a.g[0].w_off = 0; a.g[0].bias_off = 0;
        op.kind = OP_CONV128_HALO; op.bn_tile = 128; op.ticket_slot = ctx->ticket_slots_used++;
        op.name = "conv3x3_c128_halo";
    } else if (d->kernel == 0) {
        const int BN = d->bn_tile;
        if (BN != 16 && BN != 32 && BN != 64 && BN != 128) RT_FAIL("op_conv: bn_tile must be 16/32/64/128");
        if (d->cin % 64) RT_FAIL("op_conv(mfma): cin=%d is not a multiple of 64", d->cin);
        if (!d->out_nchw_f32 && (d->cout % BN)) RT_FAIL("op_conv(mfma): cout=%d is not a multiple of the tile %d", d->cout, BN);
        const int cout_pad = (d->cout + BN - 1) / BN * BN;
        a.cpt = d->cin / 64; a.ksteps = d->ntaps * a.cpt;
        a.MT = (a.M + 127) / 128; a.NT = cout_pad / BN;
        const size_t per_group = (size_t)cout_pad * a.ksteps * 64;
        if (wbytes != per_group * d->groups * sizeof(f16)) RT_FAIL("op_conv(mfma): packed weight blob has %zu bytes, expected %zu", wbytes, per_group * d->groups * sizeof(f16));
        if (bbytes != (size_t)cout_pad * d->groups * sizeof(float)) RT_FAIL("op_conv(mfma): bias blob has %zu bytes, expected %zu", bbytes, (size_t)cout_pad * d->groups * sizeof(float));
        for (int g = 0; g < d->groups; ++g) { a.g[g].w_off = (uint32_t)(per_group * g); a.g[g].bias_off = cout_pad * g; }
        op.kind = OP_CONV_MFMA; op.bn_tile = BN;
        op.name = d->ntaps == 1 ? "conv1x1_mfma" : (d->ntaps == 4 ? "deconv4x4_phase_mfma" : "conv3x3_mfma");
        // Small launches (no more workgroups than CUs: small batches; bs=1 gives DLA level5 four pixel tiles) run on
        // conv_mfma_deep_kernel: a 4-slot LDS ring instead of one stage in flight.  When the launch would fill fewer than
        // half the CUs and has a K loop of >= 8 steps, K is also cut into ranges of >= 4 steps so that the grid about fills
        // the chip (filling it twice measured slower); the partial tiles go through an fp32 slab and the last workgroup of
        // a tile to arrive sums them in split order (deterministic).
        if (!d->out_nchw_f32 && (BN == 64 || BN == 128)) {
            const long long wgs = (long long)a.MT * a.NT * d->groups;
            if (wgs <= ctx->n_cus) {
                a.deep = 1;
                op.name += "_deep";
                if (ensure_tile_ctr(ctx)) return 1;
            }
            if (a.deep && wgs * 2 <= ctx->n_cus && wgs <= SPLIT_CTRS && a.ksteps >= 8) {
                int ks = (int)(ctx->n_cus / wgs);
                if (ks > a.ksteps / 4) ks = a.ksteps / 4;
                if (ks > 16) ks = 16;
                if (ks >= 2) {
                    const size_t need = (size_t)ks * wgs * 128 * BN;
                    if (need > ctx->slab_floats) {
                        // (re)allocate; earlier ops keep a stale pointer, so patch every split op recorded so far
                        float* nslab = nullptr;
                        RT_HIP(hipMalloc((void**)&nslab, need * sizeof(float)));
                        ctx->extra.push_back(nslab);
                        for (auto& prev : ctx->ops) if (prev.kind == OP_CONV_MFMA && prev.conv.ksplit > 1) prev.conv.slab = nslab;
                        ctx->slab = nslab; ctx->slab_floats = need;
                    }
                    a.ksplit = ks; a.slab = ctx->slab;
                    op.name += "_splitk";
                }
            }
        }
    } else if (d->kernel == 3) {
        if (d->groups != 1 || d->out_nchw_f32 || res) RT_FAIL("op_conv(smallc): groups/NCHW output/residual unsupported");
        if (!conv_smallc_supported(d->cin, d->cout, d->ntaps)) RT_FAIL("op_conv(smallc): no kernel for cin=%d cout=%d ntaps=%d", d->cin, d->cout, d->ntaps);
        int S = d->cin == 16 ? 5 : (d->cin == 32 ? d->ntaps : 7);
        // cin=16 also comes packed by filter rows (6 k-steps: taps (ky,0),(ky,1) | (ky,2),zero) for the
        // vertical-walk kernel of stride-1 layers; the blob size tells the two layouts apart
        if (d->cin == 16 && d->cout == 16 && d->in_stride == 1 && d->out_scale == 1 && wbytes == (size_t)(d->cout / 16) * 6 * 64 * 8 * sizeof(f16)) S = 6;
        if (wbytes != (size_t)(d->cout / 16) * S * 64 * 8 * sizeof(f16)) RT_FAIL("op_conv(smallc): weight blob size mismatch");
        a.ksteps = S;
        if (bbytes != (size_t)d->cout * sizeof(float)) RT_FAIL("op_conv(smallc): bias blob size mismatch");
        if (d->cin == 4 && in->P < 4) RT_FAIL("op_conv(smallc): the NHWC4 stem input needs a border of 4");
        a.g[0].w_off = 0; a.g[0].bias_off = 0;
        op.kind = OP_CONV_SMALLC;
        op.name = d->cin == 4 ? "stem7x7_regmfma" : "conv_smallc_regmfma";
    } else {
        RT_FAIL("op_conv: unknown kernel %d (0 = MFMA 128-px tile, 2 = MFMA 256x256 tile, 3 = register-direct MFMA, 5 / 6 / 7 = halo kernels)", d->kernel);
    }
    if (d->softmax_stat_slot >= 0 && stat_slot < 0) RT_FAIL("op_conv: softmax_stat_slot needs kernel = 2");
    if (stat_slot >= 0) {
        // epilogue-emitted spatial-softmax partials: [slot][image][chunk][256][2] floats, chunk = 128-pixel run
        if (stat_slot > 2) RT_FAIL("op_conv: softmax_stat_slot out of range");
        if (!conv_mfma256_uses_halo(a, d->groups)) RT_FAIL("op_conv: softmax_stat_slot set on a conv that does not take the halo-tile kernel");
        if (d->cout != 256 || out->C != 256) RT_FAIL("op_conv: softmax partials need a 256-channel output tensor");
        for (int g = 0; g < d->groups; ++g) if (d->out_coff[g] != 0) RT_FAIL("op_conv: softmax partials need output channel offset 0");
        const int chunks = d->groups * (d->Hm / 8) * (d->Wm / 32) * 2;
        if (!ctx->stat_buf) {
            RT_HIP(hipMalloc((void**)&ctx->stat_buf, (size_t)3 * in->B * chunks * 256 * 2 * sizeof(float)));
            ctx->extra.push_back(ctx->stat_buf);
            ctx->stat_chunks = chunks; ctx->stat_B = in->B;
        } else if (ctx->stat_chunks != chunks || ctx->stat_B != in->B) {
            RT_FAIL("op_conv: softmax partial producers of one fusion must have equal shapes");
        }
        ctx->stat_tensor[stat_slot] = d->out_tensor;
        op.stat_out = ctx->stat_buf + (size_t)stat_slot * in->B * chunks * 256 * 2;
    }
    ctx->ops.push_back(op);
    return 0;
}

extern "C" int rtm3d_op_input_nhwc4(rtm3d_ctx* ctx, int out_tensor) {
    Tensor* o = ctx ? get_tensor(ctx, out_tensor) : nullptr;
    if (!o || o->C != 4) RT_FAIL("op_input_nhwc4: output tensor must have 4 channels");
    Op op;
    op.kind = OP_INPUT4; op.name = "nchw_f32_to_nhwc4_f16"; op.stem_cout = 0;
    memset(&op.stem, 0, sizeof(op.stem));
    op.stem.out = o->base; op.stem.B = o->B; op.stem.H = o->H; op.stem.W = o->W;
    op.stem.out_Hp = o->Hp; op.stem.out_Wp = o->Wp; op.stem.out_P = o->P;
    op.flops = 0; op.bytes = (double)o->B * o->H * o->W * (12.0 + 8.0);
    ctx->ops.push_back(op);
    return 0;
}

extern "C" int rtm3d_op_stem_fused(rtm3d_ctx* ctx, int x4_tensor, int out_tensor, int out_coff, int w_base_blob, int b_base_blob,
                                   int w_l0_blob, int b_l0_blob, int w_l1_blob, int b_l1_blob) {
    Tensor* x = ctx ? get_tensor(ctx, x4_tensor) : nullptr;
    Tensor* o = ctx ? get_tensor(ctx, out_tensor) : nullptr;
    if (!x || !o) RT_FAIL("op_stem_fused: bad tensors");
    const bool three = w_l1_blob >= 0;
    const int oc = three ? 32 : 16, sc = three ? 2 : 1;
    if (x->C != 4 || x->P < 4) RT_FAIL("op_stem_fused: the input must be the NHWC4 image tensor with a border >= 4");
    if (o->H * sc != x->H || o->W * sc != x->W || o->B != x->B || out_coff < 0 || out_coff + oc > o->C || (out_coff % 8)) RT_FAIL("op_stem_fused: output slice mismatch");
    if (x->H % 16 || x->W % 32) RT_FAIL("op_stem_fused: needs H %% 16 == 0 and W %% 32 == 0 (got %dx%d)", x->H, x->W);
    // the kernel addresses one image (3 fp32 planes) through a buffer descriptor with 32-bit byte offsets
    if ((long long)x->H * x->W * 3 * 4 >= (1LL << 31)) RT_FAIL("op_stem_fused: image of %dx%d exceeds the 2 GiB the stem addresses per image", x->H, x->W);
    size_t wb = 0, bb = 0, wl = 0, bl = 0, w1b = 0, b1b = 0;
    const f16* w0 = (const f16*)get_blob(ctx, w_base_blob, &wb);
    const float* b0 = (const float*)get_blob(ctx, b_base_blob, &bb);
    const f16* w1 = (const f16*)get_blob(ctx, w_l0_blob, &wl);
    const float* b1 = (const float*)get_blob(ctx, b_l0_blob, &bl);
    const f16* w2 = three ? (const f16*)get_blob(ctx, w_l1_blob, &w1b) : nullptr;
    const float* b2 = three ? (const float*)get_blob(ctx, b_l1_blob, &b1b) : nullptr;
    if (!w0 || !b0 || !w1 || !b1 || wb != 7 * 64 * 8 * sizeof(f16) || wl != 5 * 64 * 8 * sizeof(f16) || bb != 16 * sizeof(float) || bl != 16 * sizeof(float))
        RT_FAIL("op_stem_fused: weight/bias blob size mismatch");
    if (three && (!w2 || !b2 || w1b != 2 * 5 * 64 * 8 * sizeof(f16) || b1b != 32 * sizeof(float))) RT_FAIL("op_stem_fused: level1 weight/bias blob size mismatch");
    // the conversion pass in front of this op becomes unnecessary: the kernel reads the caller's fp32 batch directly
    for (auto& prev : ctx->ops)
        if (prev.kind == OP_INPUT4 && prev.stem.out == x->base) prev.stem_cout = -1;
    Op op;
    op.kind = OP_STEM_FUSED; op.name = three ? "stem7x7+3x3+3x3s2_fused" : "stem7x7+conv3x3_fused";
    StemFusedArgs& a = op.sf;
    memset(&a, 0, sizeof(a));
    a.x4 = x->base; a.out = o->base; a.w_base = w0; a.w_l0 = w1; a.w_l1 = w2; a.b_base = b0; a.b_l0 = b1; a.b_l1 = b2;
    a.B = x->B; a.H = x->H; a.W = x->W;
    a.x_Hp = x->Hp; a.x_Wp = x->Wp; a.x_P = x->P;
    a.o_Hp = o->Hp; a.o_Wp = o->Wp; a.o_C = o->C; a.o_P = o->P; a.o_coff = out_coff;
    a.tiles_x = x->W / 32; a.tiles_y = x->H / 16;
    const double px = (double)x->B * x->H * x->W;
    // the reference layers (3 real input channels): 7x7 3->16, 3x3 16->16 at full resolution, 3x3 16->32 at half
    op.flops = 2.0 * px * (49.0 * 3 * 16 + 9.0 * 16 * 16 + (three ? 9.0 * 16 * 32 / 4 : 0.0));
    op.bytes = px * (8.0 + (three ? 64.0 / 4 : 32.0));           // NHWC4 image in, 16-channel map (or the 32-channel half-res map) out
    ctx->ops.push_back(op);
    return 0;
}

extern "C" int rtm3d_op_conv32s2_fused(rtm3d_ctx* ctx, int in_tensor, int in_coff, int conv_tensor, int conv_coff, int proj_tensor, int proj_coff,
                                       int w_conv_blob, int b_conv_blob, int w_proj_blob, int b_proj_blob) {
    Tensor* x = ctx ? get_tensor(ctx, in_tensor) : nullptr;
    Tensor* oc = ctx ? get_tensor(ctx, conv_tensor) : nullptr;
    Tensor* op_ = ctx ? get_tensor(ctx, proj_tensor) : nullptr;
    if (!x || !oc || !op_) RT_FAIL("op_conv32s2_fused: bad tensors");
    if (x->P < 1 || in_coff < 0 || in_coff + 32 > x->C || (in_coff % 8)) RT_FAIL("op_conv32s2_fused: input slice mismatch (32 channels, border >= 1)");
    if (x->H % 16 || x->W % 64) RT_FAIL("op_conv32s2_fused: needs H %% 16 == 0 and W %% 64 == 0 (got %dx%d)", x->H, x->W);
    for (Tensor* o : {oc, op_})
        if (o->H * 2 != x->H || o->W * 2 != x->W || o->B != x->B) RT_FAIL("op_conv32s2_fused: outputs must have half the input resolution");
    if (conv_coff < 0 || conv_coff + 64 > oc->C || (conv_coff % 8) || proj_coff < 0 || proj_coff + 64 > op_->C || (proj_coff % 8)) RT_FAIL("op_conv32s2_fused: output slice mismatch");
    size_t wc = 0, bc = 0, wp = 0, bp = 0;
    const f16* w0 = (const f16*)get_blob(ctx, w_conv_blob, &wc);
    const float* b0 = (const float*)get_blob(ctx, b_conv_blob, &bc);
    const f16* w1 = (const f16*)get_blob(ctx, w_proj_blob, &wp);
    const float* b1 = (const float*)get_blob(ctx, b_proj_blob, &bp);
    if (!w0 || !b0 || !w1 || !b1 || wc != 9 * 4 * 64 * 8 * sizeof(f16) || wp != 4 * 64 * 8 * sizeof(f16) || bc != 64 * sizeof(float) || bp != 64 * sizeof(float))
        RT_FAIL("op_conv32s2_fused: weight/bias blob size mismatch");
    if (ctx->ticket_slots_used >= TICKET_SLOTS) RT_FAIL("op_conv32s2_fused: out of ticket counters");
    if (ensure_tile_ctr(ctx)) return 1;
    Op op;
    op.kind = OP_CONV32S2_FUSED; op.name = "pool+proj1x1+conv3x3s2_fused"; op.ticket_slot = ctx->ticket_slots_used++;
    Conv32S2Args& a = op.c32;
    memset(&a, 0, sizeof(a));
    a.in = x->base; a.out_conv = oc->base; a.out_proj = op_->base; a.w_conv = w0; a.w_proj = w1; a.b_conv = b0; a.b_proj = b1;
    a.B = x->B; a.Ho = oc->H; a.Wo = oc->W;
    a.in_Hp = x->Hp; a.in_Wp = x->Wp; a.in_C = x->C; a.in_P = x->P; a.in_coff = in_coff;
    a.oc_Hp = oc->Hp; a.oc_Wp = oc->Wp; a.oc_C = oc->C; a.oc_P = oc->P; a.oc_coff = conv_coff;
    a.op_Hp = op_->Hp; a.op_Wp = op_->Wp; a.op_C = op_->C; a.op_P = op_->P; a.op_coff = proj_coff;
    const double opx = (double)x->B * oc->H * oc->W;
    op.flops = 2.0 * opx * (9.0 * 32 * 64 + 32.0 * 64);
    op.bytes = opx * (4 * 64.0 + 2 * 128.0);                   // 32-channel input once, two 64-channel outputs
    ctx->ops.push_back(op);
    return 0;
}

extern "C" int rtm3d_op_conv64_root(rtm3d_ctx* ctx, int in_tensor, int in_coff, int res_tensor, int res_coff, int conv_relu,
                                    int w_conv_blob, int b_conv_blob, int w_root_blob, int b_root_blob,
                                    int out_tensor, int out_coff, int root_relu, int pool_tensor, int pool_coff,
                                    int s2d_tensor, int s2d_coff) {
    Tensor* in = ctx ? get_tensor(ctx, in_tensor) : nullptr;
    Tensor* res = ctx ? get_tensor(ctx, res_tensor) : nullptr;
    Tensor* out = ctx && out_tensor >= 0 ? get_tensor(ctx, out_tensor) : nullptr;
    Tensor* pool = ctx && pool_tensor >= 0 ? get_tensor(ctx, pool_tensor) : nullptr;
    Tensor* s2d = ctx && s2d_tensor >= 0 ? get_tensor(ctx, s2d_tensor) : nullptr;
    if (!in || !res || (out_tensor >= 0 && !out) || (pool_tensor >= 0 && !pool) || (s2d_tensor >= 0 && !s2d)) RT_FAIL("op_conv64_root: bad tensors");
    if (!out && !s2d) RT_FAIL("op_conv64_root: out_tensor < 0 (no ordinary copy of the root output) needs a space-to-depth copy");
    if (s2d) {
        if (s2d->H * 2 != in->H || s2d->W * 2 != in->W || s2d->B != in->B || s2d_coff < 0 || s2d_coff + 256 > s2d->C || (s2d_coff % 8))
            RT_FAIL("op_conv64_root: space-to-depth output slice mismatch (half resolution, 4 x 64 channels)");
        if (s2d == out || s2d == res || s2d == in) RT_FAIL("op_conv64_root: the space-to-depth copy aliases an operand");
    }
    if (in->P < 1 || in_coff < 0 || in_coff + 64 > in->C || (in_coff % 8)) RT_FAIL("op_conv64_root: input slice mismatch (64 channels, border >= 1)");
    if (in->H % 8 || in->W % 32) RT_FAIL("op_conv64_root: needs H %% 8 == 0 and W %% 32 == 0 (got %dx%d)", in->H, in->W);
    for (Tensor* t : {res, out})
        if (t && (t->H != in->H || t->W != in->W || t->B != in->B)) RT_FAIL("op_conv64_root: residual / output shape mismatch");
    if (res_coff < 0 || res_coff + 64 > res->C || (res_coff % 8) || (out && (out_coff < 0 || out_coff + 64 > out->C || (out_coff % 8)))) RT_FAIL("op_conv64_root: residual / output slice mismatch");
    if (out && res == out && res_coff < out_coff + 64 && out_coff < res_coff + 64) RT_FAIL("op_conv64_root: the root output overlaps x1 (another workgroup may still read it)");
    if (out && in == out && in_coff < out_coff + 64 && out_coff < in_coff + 64) RT_FAIL("op_conv64_root: the root output overlaps the conv input");
    if (pool) {
        if (pool->H * 2 != in->H || pool->W * 2 != in->W || pool->B != in->B || pool_coff < 0 || pool_coff + 64 > pool->C || (pool_coff % 8))
            RT_FAIL("op_conv64_root: pooled output slice mismatch (half resolution, 64 channels)");
        // the pooled map and the space-to-depth copy are both half resolution and may be slices of ONE tensor: different lanes of
        // one launch write them, so their channel ranges must not meet
        if (s2d && pool == s2d && pool_coff < s2d_coff + 256 && s2d_coff < pool_coff + 64)
            RT_FAIL("op_conv64_root: the pooled slice [%d,%d) overlaps the space-to-depth slice [%d,%d) of the same tensor", pool_coff, pool_coff + 64, s2d_coff, s2d_coff + 256);
        if (pool == in || pool == res || (out && pool == out)) RT_FAIL("op_conv64_root: the pooled output aliases an operand");
    }
    size_t wc = 0, bc = 0, wr = 0, br = 0;
    const f16* w0 = (const f16*)get_blob(ctx, w_conv_blob, &wc);
    const float* b0 = (const float*)get_blob(ctx, b_conv_blob, &bc);
    const f16* w1 = (const f16*)get_blob(ctx, w_root_blob, &wr);
    const float* b1 = (const float*)get_blob(ctx, b_root_blob, &br);
    if (!w0 || !b0 || !w1 || !b1 || wc != (size_t)9 * 64 * 64 * sizeof(f16) || bc != 64 * sizeof(float) || wr != (size_t)64 * 128 * sizeof(f16) || br != 64 * sizeof(float))
        RT_FAIL("op_conv64_root: weight/bias blob size mismatch");
    if (ctx->ticket_slots_used >= TICKET_SLOTS) RT_FAIL("op_conv64_root: out of ticket counters");
    if (ensure_tile_ctr(ctx)) return 1;
    Op op;
    ConvKArgs& a = op.conv;
    memset(&a, 0, sizeof(a));
    a.in = in->base; a.wgt = w0; a.bias = b0; a.res = res->base; a.out = nullptr;
    a.HmWm = in->H * in->W; a.Wm = in->W; a.M = in->B * a.HmWm;
    a.in_Hp = in->Hp; a.in_Wp = in->Wp; a.in_C = in->C; a.in_stride = 1; a.in_P = in->P; a.out_scale = 1;
    a.res_Hp = res->Hp; a.res_Wp = res->Wp; a.res_C = res->C; a.res_P = res->P;
    a.cin = 64; a.cout = 64; a.ntaps = 9; a.relu = conv_relu ? 1 : 0; a.cpt = 1; a.ksteps = 9; a.NT = 1;
    for (int t = 0; t < 9; ++t) a.g[0].tap_off[t] = ((t / 3 - 1) * in->Wp + (t % 3 - 1)) * in->C;
    a.g[0].in_coff = in_coff; a.g[0].res_coff = res_coff;
    RootKArgs& r = op.root;
    memset(&r, 0, sizeof(r));
    r.w = w1; r.bias = b1;
    if (out) { r.out = out->base; r.o_Hp = out->Hp; r.o_Wp = out->Wp; r.o_C = out->C; r.o_P = out->P; r.o_coff = out_coff; }
    r.relu = root_relu ? 1 : 0;
    if (pool) { r.pool = pool->base; r.p_Hp = pool->Hp; r.p_Wp = pool->Wp; r.p_C = pool->C; r.p_P = pool->P; r.p_coff = pool_coff; }
    if (s2d) { r.s2d = s2d->base; r.s_Hp = s2d->Hp; r.s_Wp = s2d->Wp; r.s_C = s2d->C; r.s_P = s2d->P; r.s_coff = s2d_coff; }
    op.kind = OP_CONV64_ROOT; op.groups = 1; op.bn_tile = 64; op.epi_nchw = 0; op.out_slot = -1; op.ticket_slot = ctx->ticket_slots_used++;
    op.name = pool ? "conv3x3_c64+root1x1+pool_fused" : "conv3x3_c64+root1x1_fused";
    if (s2d) op.name += "+s2d";
    const double M = (double)a.M;
    op.flops = 2.0 * M * (9.0 * 64 * 64 + 128.0 * 64);
    op.bytes = 2.0 * M * (64.0 * 2 + (out ? 64.0 : 0.0) + (pool ? 16.0 : 0.0) + (s2d ? 64.0 : 0.0));      // conv input, x1, root output (+ the pooled map, + the second copy)
    ctx->ops.push_back(op);
    return 0;
}

extern "C" int rtm3d_op_patch_mask(rtm3d_ctx* ctx, int tensor, int yx_blob, int img_H, int img_W, int origin) {
    Tensor* t = ctx ? get_tensor(ctx, tensor) : nullptr;
    if (!t) RT_FAIL("op_patch_mask: bad tensor");
    if (t->P != 0 || t->H != t->W || (t->C % 8) || img_H < 1 || img_W < 1 || origin < 0 || origin >= t->H) RT_FAIL("op_patch_mask: the patch tensor must be square, borderless, with a multiple of 8 channels");
    size_t bytes = 0;
    const int32_t* yx = (const int32_t*)get_blob(ctx, yx_blob, &bytes);
    if (!yx || bytes < (size_t)t->B * 2 * sizeof(int32_t)) RT_FAIL("op_patch_mask: the (y, x) blob needs 2 int32 per slot");
    Op op;
    op.kind = OP_PATCH_MASK; op.name = "patch_mask";
    op.pm.base = t->base; op.pm.yx = yx; op.pm.n_slots = t->B; op.pm.S = t->H; op.pm.C = t->C;
    op.pm.img_H = img_H; op.pm.img_W = img_W; op.pm.origin = origin;
    op.flops = 0; op.bytes = 0;
    ctx->ops.push_back(op);
    return 0;
}

extern "C" int rtm3d_op_headout(rtm3d_ctx* ctx, int in_tensor, int w_blob, int bias_blob, int nheads, const int* cout4) {
    Tensor* in = ctx ? get_tensor(ctx, in_tensor) : nullptr;
    if (!in || !cout4) RT_FAIL("op_headout: bad arguments");
    if (nheads < 1 || nheads > 4) RT_FAIL("op_headout: nheads must be in [1,4]");
    if (in->C != 256 * nheads || in->P < 1) RT_FAIL("op_headout: input must be the nheads x 256 channel head tensor with a border >= 1");
    size_t wb = 0, bb = 0;
    const f16* w = (const f16*)get_blob(ctx, w_blob, &wb);
    const float* b = (const float*)get_blob(ctx, bias_blob, &bb);
    if (!w || !b || wb != (size_t)nheads * 9 * 8 * 64 * 8 * sizeof(f16) || bb != (size_t)nheads * 16 * sizeof(float)) RT_FAIL("op_headout: weight/bias blob size mismatch");
    Op op;
    op.kind = OP_HEADOUT; op.name = "conv3x3_headout_halo";
    HeadOutArgs& a = op.ho;
    memset(&a, 0, sizeof(a));
    a.in = in->base; a.wgt = w; a.bias = b;
    int csum = 0;
    for (int i = 0; i < nheads; ++i) { if (cout4[i] < 1 || cout4[i] > 16) RT_FAIL("op_headout: cout must be in [1,16]"); a.cout[i] = cout4[i]; csum += cout4[i]; }
    a.nheads = nheads;
    a.B = in->B; a.H = in->H; a.W = in->W;
    a.in_Hp = in->Hp; a.in_Wp = in->Wp; a.in_C = in->C; a.in_P = in->P;
    // 16-row tiles (less halo) once they still give every CU its two workgroups twice over, 8-row tiles for small batches
    a.tile_rows = (long long)in->B * ((in->H + 15) / 16) * ((in->W + 31) / 32) * nheads >= 4LL * ctx->n_cus ? 16 : 8;
    a.tiles_x = (in->W + 31) / 32; a.tiles_y = (in->H + a.tile_rows - 1) / a.tile_rows;
    op.flops = 2.0 * in->B * in->H * in->W * 9.0 * 256.0 * csum;
    op.bytes = (double)in->B * in->H * in->W * (256.0 * nheads * 2 + csum * 4.0);
    ctx->ops.push_back(op);
    return 0;
}

extern "C" int rtm3d_op_maxpool(rtm3d_ctx* ctx, int in_tensor, int in_coff, int out_tensor, int out_coff,
                                int channels, int ksize, int stride, int pad) {
    Tensor* in = ctx ? get_tensor(ctx, in_tensor) : nullptr;
    Tensor* out = ctx ? get_tensor(ctx, out_tensor) : nullptr;
    if (!in || !out) RT_FAIL("op_maxpool: bad tensors");
    if (channels % 8 || in_coff % 8 || out_coff % 8 || in_coff + channels > in->C || out_coff + channels > out->C) RT_FAIL("op_maxpool: bad channel slice");
    if (pad > in->P) RT_FAIL("op_maxpool: input border %d narrower than pool padding %d", in->P, pad);
    if ((out->H - 1) * stride - pad + ksize - 1 >= in->H + in->P || (out->W - 1) * stride - pad + ksize - 1 >= in->W + in->P) RT_FAIL("op_maxpool: window leaves the padded input");
    Op op;
    op.kind = OP_MAXPOOL; op.name = "maxpool";
    PoolKArgs& a = op.pool;
    a.in = in->base; a.out = out->base; a.B = in->B; a.Ho = out->H; a.Wo = out->W; a.C8 = channels / 8;
    a.in_Hp = in->Hp; a.in_Wp = in->Wp; a.in_C = in->C; a.in_P = in->P; a.in_coff = in_coff;
    a.out_Hp = out->Hp; a.out_Wp = out->Wp; a.out_C = out->C; a.out_P = out->P; a.out_coff = out_coff;
    a.ksize = ksize; a.stride = stride; a.pad = pad;
    op.flops = 0;
    op.bytes = 2.0 * in->B * channels * ((double)in->H * in->W + (double)out->H * out->W);
    ctx->ops.push_back(op);
    return 0;
}

extern "C" int rtm3d_op_maxpool_s2d(rtm3d_ctx* ctx, int in_tensor, int in_coff, int out_tensor, int out_coff, int channels) {
    Tensor* in = ctx ? get_tensor(ctx, in_tensor) : nullptr;
    Tensor* out = ctx ? get_tensor(ctx, out_tensor) : nullptr;
    if (!in || !out) RT_FAIL("op_maxpool_s2d: bad tensors");
    if (channels % 8 || in_coff % 8 || out_coff % 8 || in_coff < 0 || out_coff < 0 || in_coff + 4 * channels > in->C || out_coff + channels > out->C) RT_FAIL("op_maxpool_s2d: bad channel slices (input: 4 x channels)");
    if (in->H != out->H || in->W != out->W || in->B != out->B) RT_FAIL("op_maxpool_s2d: the space-to-depth copy and the pooled map have one resolution");
    if (in == out && in_coff < out_coff + channels && out_coff < in_coff + 4 * channels) RT_FAIL("op_maxpool_s2d: output overlaps the input slices");
    Op op;
    op.kind = OP_MAXPOOL; op.name = "maxpool_s2d";
    PoolKArgs& a = op.pool;
    a.in = in->base; a.out = out->base; a.B = in->B; a.Ho = out->H; a.Wo = out->W; a.C8 = channels / 8;
    a.in_Hp = in->Hp; a.in_Wp = in->Wp; a.in_C = in->C; a.in_P = in->P; a.in_coff = in_coff;
    a.out_Hp = out->Hp; a.out_Wp = out->Wp; a.out_C = out->C; a.out_P = out->P; a.out_coff = out_coff;
    a.ksize = 0; a.stride = 0; a.pad = 0;                       // ksize 0 = the space-to-depth form: max over the four phase slices of one pixel
    op.flops = 0;
    op.bytes = 2.0 * in->B * channels * 5.0 * (double)out->H * out->W;
    ctx->ops.push_back(op);
    return 0;
}

extern "C" int rtm3d_op_softmax_fuse(rtm3d_ctx* ctx, int z_in, int z_out, int n_u, const int* u_tensors) {
    Tensor* zi = ctx ? get_tensor(ctx, z_in) : nullptr;
    Tensor* zo = ctx ? get_tensor(ctx, z_out) : nullptr;
    if (!zi || !zo || n_u < 1 || n_u > 3 || !u_tensors) RT_FAIL("op_softmax_fuse: bad arguments");
    if (zi->C != 256 || zo->C != 256 || zi->H != zo->H || zi->W != zo->W || zi->B != zo->B) RT_FAIL("op_softmax_fuse: z tensors must be 256-channel with equal shape");
    Op op;
    op.kind = OP_SOFTMAX; op.name = "softmax_fuse";
    SoftmaxKArgs& a = op.sm;
    memset(&a, 0, sizeof(a));
    a.z_in = zi->base; a.z_out = zo->base; a.n_u = n_u;
    a.B = zi->B; a.H = zi->H; a.W = zi->W; a.C = 256;
    a.z_Hp = zo->Hp; a.z_Wp = zo->Wp; a.z_C = zo->C; a.z_P = zo->P;
    a.zi_Hp = zi->Hp; a.zi_Wp = zi->Wp; a.zi_C = zi->C; a.zi_P = zi->P;
    for (int i = 0; i < n_u; ++i) {
        Tensor* u = get_tensor(ctx, u_tensors[i]);
        if (!u || u->C != 256 || u->H != zi->H || u->W != zi->W || u->B != zi->B) RT_FAIL("op_softmax_fuse: u tensor %d mismatch", i);
        a.u[i] = u->base; a.u_Hp[i] = u->Hp; a.u_Wp[i] = u->Wp; a.u_C[i] = u->C; a.u_P[i] = u->P;
    }
    // statistics pass: one workgroup per (image, chunk): two rows at the batch sizes the path is quoted on, single rows for small batches
    const bool small = (long long)a.B * ((a.H + 1) / 2) < 2LL * ctx->n_cus;
    a.rows_per_chunk = small ? 1 : 2;
    a.chunks = (a.H + a.rows_per_chunk - 1) / a.rows_per_chunk;
    // the apply pass (which does not touch the partials) runs one row per workgroup, rows cut into up to four column segments of
    // >= 64 pixels (same box, bs=32: 0.503 ms with two full rows per workgroup, 0.495 with one, 0.483 with quarter rows), and
    // into more (up to eight, >= 32 pixels) until a small batch has about two workgroups per CU (bs=1: 48 workgroups -> 384)
    a.apply_rows = 1; a.xsplit = 1;
    while (a.xsplit < 4 && a.W / (a.xsplit * 2) >= 64) a.xsplit *= 2;
    while ((long long)a.B * a.H * a.xsplit < 2LL * ctx->n_cus && a.xsplit < 8 && a.W / (a.xsplit * 2) >= 32) a.xsplit *= 2;
    a.seg_w = ((a.W + a.xsplit - 1) / a.xsplit + 7) / 8 * 8;
    a.apply_chunks = a.H * a.xsplit;
    // partials already emitted by the producers' epilogues (rtm3d_conv_desc.softmax_stat_slot)?
    bool emitted = ctx->stat_buf != nullptr && ctx->stat_B == a.B;
    for (int i = 0; i < n_u && emitted; ++i) emitted = ctx->stat_tensor[i] == u_tensors[i];
    for (int i = 0; i < 3; ++i) if ((i < n_u) != (ctx->stat_tensor[i] >= 0) && ctx->stat_buf) emitted = false;
    if (ctx->stat_buf && !emitted) RT_FAIL("op_softmax_fuse: the recorded softmax_stat_slot producers do not match this fusion's operands");
    if (emitted) {
        a.partial = ctx->stat_buf;
        a.partial_chunks = ctx->stat_chunks;
        ctx->stat_buf = nullptr;                       // consumed (the allocation stays owned by ctx->extra)
        ctx->stat_tensor[0] = ctx->stat_tensor[1] = ctx->stat_tensor[2] = -1;
    } else {
        void* part = nullptr;
        RT_HIP(hipMalloc(&part, (size_t)n_u * a.B * a.chunks * a.C * 2 * sizeof(float)));
        ctx->extra.push_back(part);
        a.partial = (float*)part;
        a.partial_chunks = 0;
    }
    void* st = nullptr;
    RT_HIP(hipMalloc(&st, (size_t)n_u * a.B * a.C * 2 * sizeof(float)));
    ctx->extra.push_back(st);
    a.stats = (float*)st;
    op.flops = 0;
    // apply pass: read z_in and every u once, write z_out; the stand-alone reduction pass (when the producers did not
    // emit the partials from their epilogues) reads every u once more
    op.bytes = 2.0 * a.B * a.H * a.W * 256.0 * ((emitted ? 1.0 : 2.0) * n_u + 2.0);
    ctx->ops.push_back(op);
    return 0;
}

static int launch_op(rtm3d_ctx* ctx, Op& op, hipStream_t s, const float* d_in, float* const d_out[4]) {
    hipError_t e = hipSuccess;
    unsigned int* const ctr256 = ctx->tile_ctr;       // the conv256 kernels' shared per-XCD ticket counters
    switch (op.kind) {
        case OP_CONV_MFMA: {
            ConvKArgs a = op.conv;
            if (op.epi_nchw) a.out = d_out[op.out_slot];
            e = a.deep ? launch_conv_mfma_deep(a, op.bn_tile, op.groups, ctx->tile_ctr + 8 + TICKET_SLOTS, s)
                       : launch_conv_mfma(a, op.bn_tile, op.groups, op.epi_nchw, s);
            break;
        }
        case OP_CONV_MFMA256: e = launch_conv_mfma256(op.conv, op.groups, ctr256, op.stat_out, s); break;
        case OP_CONV64_HALO: e = launch_conv64_halo(op.conv, ctx->n_cus, ctx->tile_ctr + 8 + op.ticket_slot, s); break;
        case OP_CONV64S2_HALO: e = launch_conv64s2_halo(op.conv, ctx->n_cus, ctx->tile_ctr + 8 + op.ticket_slot, s); break;
        case OP_CONV64_ROOT: e = launch_conv64_root(op.conv, op.root, ctx->n_cus, ctx->tile_ctr + 8 + op.ticket_slot, s); break;
        case OP_CONV128_HALO: e = launch_conv128_halo(op.conv, ctx->n_cus, ctx->tile_ctr + 8 + op.ticket_slot, s); break;
        case OP_STEM_FUSED: {
            StemFusedArgs a = op.sf;
            a.x_nchw = d_in;                     // null: the NHWC4 tensor was filled by rtm3d_preprocess_batch
            e = launch_stem_fused(a, s);
            break;
        }
        case OP_CONV32S2_FUSED: e = launch_conv32s2_fused(op.c32, ctx->n_cus, ctx->tile_ctr + 8 + op.ticket_slot, s); break;
        case OP_CONV_SMALLC: e = launch_conv_smallc(op.conv, s); break;
        case OP_INPUT4: if (!d_in || op.stem_cout == -1) break;   // filled by rtm3d_preprocess_batch (out_mode 1), or its only
                                                                  // consumer is the fused stem, which reads the fp32 batch itself
            e = launch_nchw_to_nhwc4(d_in, op.stem.out, op.stem.B, op.stem.H, op.stem.W, op.stem.out_Hp, op.stem.out_Wp, op.stem.out_P, s); break;
        case OP_HEADOUT: {
            HeadOutArgs a = op.ho;
            for (int i = 0; i < 4; ++i) a.out[i] = d_out[i];
            e = launch_conv_headout(a, s);
            break;
        }
        case OP_MAXPOOL: e = launch_maxpool(op.pool, s); break;
        case OP_SOFTMAX: e = launch_softmax_fuse(op.sm, s); break;
        case OP_PATCH_MASK: e = launch_patch_mask(op.pm, s); break;
    }
    if (e != hipSuccess) { rt_set_error("launch of op '%s' failed: %s", op.name.c_str(), hipGetErrorString(e)); return 1; }
    return 0;
}

// Zeroes the ticket / arrival counters.  A kernel of our own instead of hipMemsetAsync: inside a captured graph the memset
// becomes a memset NODE, and with three or more graph execs of different node counts alive in one context, re-launching an
// older exec ran the kernel node BEHIND that memset node with stale arguments (round 3: the fp32 -> NHWC4 conversion of one
// graph had no effect, the first convolution of another faulted on an unmapped address; all-kernel graphs alternate fine).
__global__ void zero_words_kernel(unsigned int* __restrict__ p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}
static hipError_t zero_counters(rtm3d_ctx* ctx, hipStream_t s) {
    if (!ctx->tile_ctr) return hipSuccess;
    hipLaunchKernelGGL(zero_words_kernel, dim3(((int)TILE_CTR_WORDS + 255) / 256), dim3(256), 0, s, ctx->tile_ctr, (int)TILE_CTR_WORDS);
    return hipGetLastError();
}

static int replay_eager(rtm3d_ctx* ctx, hipStream_t s, const float* d_in, float* const d_out_logits[4], bool probes) {
    // The persistent convs share 8 self-resetting ticket counters; after an aborted launch (or a replay torn down half
    // way) they would be left non-zero and later launches would silently skip tiles.  Zeroing them in stream order at
    // the head of every replay costs one tiny launch (zero_counters).
    RT_HIP(zero_counters(ctx, s));
    if (ctx->test_memset_in_replay && ctx->tile_ctr) RT_HIP(hipMemsetAsync(ctx->tile_ctr, 0, sizeof(unsigned int), s));   // (test hook only)
    const int n = (int)ctx->ops.size();
    for (int i = 0; i < n; ++i) {
        Op& op = ctx->ops[i];
        for (int m = 0; m < ctx->n_marks; ++m)
            if (ctx->mark_op[m] == i) RT_HIP(hipEventRecord(ctx->mark_ev[m], s));
        const bool probe = probes && (i == ctx->probe_op);
        const int slot = ctx->probe_count % PROBE_RING;
        if (probe) RT_HIP(hipEventRecord(ctx->probe_ev[2 * slot], s));
        if (launch_op(ctx, op, s, d_in, d_out_logits)) return 1;
        if (probe) { RT_HIP(hipEventRecord(ctx->probe_ev[2 * slot + 1], s)); ctx->probe_count++; }
    }
    if (ctx->n_marks) RT_HIP(hipEventRecord(ctx->mark_ev[ctx->n_marks], s));
    return 0;
}

static const size_t GRAPH_CACHE = 8;

// number of nodes of `graph` that are not kernel nodes (-1: the runtime could not enumerate them, treated as a refusal)
static int graph_non_kernel_nodes(hipGraph_t graph) {
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) return -1;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) return -1;
    int bad = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nodes[i], &ty) != hipSuccess) return -1;
        if (ty != hipGraphNodeTypeKernel) ++bad;
    }
    return bad;
}

static int forward_on_stream(rtm3d_ctx* ctx, hipStream_t s, const float* d_in, float* const d_out_logits[4]);

// Replays of one context never overlap: the workspace and the ticket counters are shared.  Every replay (rtm3d_forward and
// rtm3d_forward_timed alike) is bracketed by these two: a replay on another stream than the previous one waits for it.
static int replay_begin(rtm3d_ctx* ctx, hipStream_t s) {
    if (ctx->has_last && ctx->last_stream != s) RT_HIP(hipStreamWaitEvent(s, ctx->done_ev, 0));
    return 0;
}
static int replay_end(rtm3d_ctx* ctx, hipStream_t s) {
    if (!ctx->done_ev) RT_HIP(hipEventCreateWithFlags(&ctx->done_ev, hipEventDisableTiming));
    RT_HIP(hipEventRecord(ctx->done_ev, s));
    ctx->last_stream = s; ctx->has_last = true;
    return 0;
}

extern "C" int rtm3d_forward(rtm3d_ctx* ctx, void* stream, const float* d_in, float* const d_out_logits[4]) {
    if (!ctx || !d_out_logits) RT_FAIL("forward: null argument");
    if (!d_out_logits[0]) RT_FAIL("forward: null logits buffer 0");
    if (ctx->ops.empty()) RT_FAIL("forward: empty plan");
    hipStream_t s = (hipStream_t)stream;
    if (replay_begin(ctx, s)) return 1;
    const int rc = forward_on_stream(ctx, s, d_in, d_out_logits);
    // recorded even after a failed replay: whatever part of it was enqueued still owns the workspace
    if (replay_end(ctx, s)) return 1;
    return rc;
}

static int forward_on_stream(rtm3d_ctx* ctx, hipStream_t s, const float* d_in, float* const d_out_logits[4]) {
    if (!ctx->graph_mode || ctx->probe_op >= 0) return replay_eager(ctx, s, d_in, d_out_logits, true);
    // graph replay: the kernel arguments are baked into the graph, so it is keyed by the caller's buffers (a serving
    // loop re-uses the same few buffers: torch's caching allocator hands back the same blocks)
    const void* key[5] = {d_in, d_out_logits[0], d_out_logits[1], d_out_logits[2], d_out_logits[3]};
    ctx->graph_clock++;
    for (auto& ge : ctx->graphs)
        if (!memcmp(ge.key, key, sizeof(key))) { ge.last_use = ctx->graph_clock; ctx->graph_hits++; RT_HIP(hipGraphLaunch(ge.exec, s)); return 0; }
    // a caller that hands over fresh buffers every time would pay a capture per call: give up on graphs for this context
    if (++ctx->graph_captures > 32 && ctx->graph_hits < ctx->graph_captures) { ctx->graph_mode = 0; return replay_eager(ctx, s, d_in, d_out_logits, true); }
    // capture on a private stream (the caller's may be the legacy default stream, which cannot be captured)
    if (!ctx->capture_stream) RT_HIP(hipStreamCreateWithFlags(&ctx->capture_stream, hipStreamNonBlocking));
    RT_HIP(hipStreamBeginCapture(ctx->capture_stream, hipStreamCaptureModeThreadLocal));
    const int rc = replay_eager(ctx, ctx->capture_stream, d_in, d_out_logits, false);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(ctx->capture_stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return 1; }
    if (ec != hipSuccess) { rt_set_error("forward: graph capture failed: %s", hipGetErrorString(ec)); return 1; }
    // ALL-KERNEL GRAPHS ONLY (include/rtm3d_hip.h, rtm3d_ctx_set_graph): with several graph execs of different node counts alive,
    // re-launching an older exec ran the kernel node behind a memset node with stale arguments (round 3).  Nothing in this file
    // captures a hipMemset*Async / hipMemcpy*Async today; if a future op launcher does, the capture is refused here instead of
    // bringing the fault back: the graph is dropped, the context leaves graph mode and this replay runs eagerly.
    {
        int bad = graph_non_kernel_nodes(graph);
        if (bad != 0) {
            (void)hipGraphDestroy(graph);
            ctx->graph_mode = 0;
            ctx->graph_refused++;
            rt_set_error("forward: the captured plan holds %d non-kernel graph node(s) (memset / memcpy nodes are not allowed in a replay graph); graph mode is off for this context", bad);
            return replay_eager(ctx, s, d_in, d_out_logits, true);
        }
    }
    rtm3d_ctx::GraphEntry ge;
    memcpy(ge.key, key, sizeof(key));
    ge.graph = graph;
    const hipError_t ei = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0);
    if (ei != hipSuccess) { (void)hipGraphDestroy(graph); rt_set_error("forward: hipGraphInstantiate failed: %s", hipGetErrorString(ei)); return 1; }
    ge.last_use = ctx->graph_clock;
    if (ctx->graphs.size() >= GRAPH_CACHE) {            // evict the least recently used entry
        size_t lru = 0;
        for (size_t i = 1; i < ctx->graphs.size(); ++i) if (ctx->graphs[i].last_use < ctx->graphs[lru].last_use) lru = i;
        (void)hipGraphExecDestroy(ctx->graphs[lru].exec); (void)hipGraphDestroy(ctx->graphs[lru].graph);
        ctx->graphs[lru] = ge;
    } else {
        ctx->graphs.push_back(ge);
    }
    RT_HIP(hipGraphLaunch(ge.exec, s));
    return 0;
}

extern "C" int rtm3d_input_tensor(rtm3d_ctx* ctx, void** d_base, int* B, int* H, int* W, int* border) {
    if (!ctx || !d_base || !B || !H || !W || !border) RT_FAIL("input_tensor: null argument");
    for (auto& op : ctx->ops)
        if (op.kind == OP_INPUT4) {
            *d_base = op.stem.out; *B = op.stem.B; *H = op.stem.H; *W = op.stem.W; *border = op.stem.out_P;
            return 0;
        }
    RT_FAIL("input_tensor: the plan has no NHWC4 input tensor");
}

extern "C" int rtm3d_ctx_set_graph(rtm3d_ctx* ctx, int enable) {
    if (!ctx) RT_FAIL("ctx_set_graph: null context");
    ctx->graph_mode = enable ? 1 : 0;
    return 0;
}

extern "C" int rtm3d_ctx_graph_stats(rtm3d_ctx* ctx, int* captures, int* hits, int* enabled) {
    if (!ctx) RT_FAIL("ctx_graph_stats: null context");
    if (captures) *captures = (int)ctx->graph_captures;
    if (hits) *hits = (int)ctx->graph_hits;
    if (enabled) *enabled = ctx->graph_mode;
    return 0;
}

extern "C" int rtm3d_ctx_debug_memset_in_replay(rtm3d_ctx* ctx, int enable) {
    if (!ctx) RT_FAIL("ctx_debug_memset_in_replay: null context");
    if (ensure_tile_ctr(ctx)) return 1;
    ctx->test_memset_in_replay = enable ? 1 : 0;
    return 0;
}

extern "C" int rtm3d_ctx_debug_read_words(rtm3d_ctx* ctx, int offset, int n, unsigned int* h_out) {
    if (!ctx || !h_out || offset < 0 || n < 1 || (size_t)offset + (size_t)n > DEBUG_WORDS) RT_FAIL("ctx_debug_read_words: bad arguments");
    if (ensure_tile_ctr(ctx)) return 1;
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(h_out, ctx->tile_ctr + DEBUG_WORD0 + offset, (size_t)n * sizeof(unsigned int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int rtm3d_probe_set(rtm3d_ctx* ctx, int op_index) {
    if (!ctx || op_index >= (int)ctx->ops.size()) RT_FAIL("probe_set: bad op index");
    if (ctx->probe_ev.empty() && op_index >= 0) {
        ctx->probe_ev.resize(2 * PROBE_RING);
        for (auto& e : ctx->probe_ev) RT_HIP(hipEventCreate(&e));
    }
    ctx->probe_op = op_index;
    ctx->probe_count = 0;
    return 0;
}

extern "C" int rtm3d_probe_read(rtm3d_ctx* ctx, double* avg_ms, int* count) {
    if (!ctx || !avg_ms || !count) RT_FAIL("probe_read: null argument");
    const int n = ctx->probe_count < PROBE_RING ? ctx->probe_count : PROBE_RING;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        RT_HIP(hipEventSynchronize(ctx->probe_ev[2 * i + 1]));
        RT_HIP(hipEventElapsedTime(&ms, ctx->probe_ev[2 * i], ctx->probe_ev[2 * i + 1]));
        sum += ms;
    }
    *avg_ms = n ? sum / n : 0.0;
    *count = n;
    return 0;
}

extern "C" int rtm3d_forward_timed(rtm3d_ctx* ctx, void* stream, const float* d_in, float* const d_out_logits[4],
                                   float* h_ms, int cap, int* n_ops) {
    if (!ctx || !d_out_logits) RT_FAIL("forward_timed: null argument");
    const int n = (int)ctx->ops.size();
    if (n_ops) *n_ops = n;
    if (!h_ms) return 0;
    if (cap < n) RT_FAIL("forward_timed: capacity %d < %d ops", cap, n);
    if (!d_out_logits[0]) RT_FAIL("forward_timed: null logits buffer 0");
    std::vector<hipEvent_t> ev(n + 1, nullptr);
    hipStream_t s = (hipStream_t)stream;
    // same ordering rule as rtm3d_forward (the timed pass shares the workspace and the ticket counters with it); the events
    // are destroyed on every path out
    int rc = 0;
    hipError_t e = hipSuccess;
    for (auto& v : ev) if ((e = hipEventCreate(&v)) != hipSuccess) { rc = 1; break; }
    if (!rc && replay_begin(ctx, s)) rc = 2;
    if (!rc) {
        if ((e = zero_counters(ctx, s)) != hipSuccess) rc = 1;
        if (!rc && (e = hipEventRecord(ev[0], s)) != hipSuccess) rc = 1;
        for (int i = 0; i < n && !rc; ++i) {
            if (launch_op(ctx, ctx->ops[i], s, d_in, d_out_logits)) { rc = 2; break; }
            if ((e = hipEventRecord(ev[i + 1], s)) != hipSuccess) rc = 1;
        }
        if (replay_end(ctx, s) && !rc) rc = 2;
        if (!rc && (e = hipEventSynchronize(ev[n])) != hipSuccess) rc = 1;
        for (int i = 0; i < n && !rc; ++i) if ((e = hipEventElapsedTime(&h_ms[i], ev[i], ev[i + 1])) != hipSuccess) rc = 1;
    }
    for (auto& v : ev) if (v) (void)hipEventDestroy(v);
    if (rc == 1) rt_set_error("forward_timed: %s", hipGetErrorString(e));
    return rc ? 1 : 0;
}

// Wall time of STAGES of one real eager replay: an event on the caller's stream in front of each op of mark_ops[] (ascending, e.g.
// the first op of the backbone, of the neck and of the heads) and one behind the last op; h_ms[i] = time from mark i to mark i + 1
// (the last: to the end).  Unlike the per-op pass (rtm3d_forward_timed) nothing is synchronised between the ops.
extern "C" int rtm3d_forward_marks(rtm3d_ctx* ctx, void* stream, const float* d_in, float* const d_out_logits[4],
                                   int n_marks, const int* mark_ops, float* h_ms) {
    if (!ctx || !d_out_logits || !d_out_logits[0] || !mark_ops || !h_ms || n_marks < 1 || n_marks > 8) RT_FAIL("forward_marks: bad arguments");
    for (int i = 0; i < n_marks; ++i)
        if (mark_ops[i] < 0 || mark_ops[i] >= (int)ctx->ops.size() || (i && mark_ops[i] <= mark_ops[i - 1]))
            RT_FAIL("forward_marks: mark %d (op %d) must be an ascending op index", i, mark_ops[i]);
    hipStream_t s = (hipStream_t)stream;
    int rc = 0;
    hipError_t e = hipSuccess;
    for (int i = 0; i <= n_marks && !rc; ++i) { ctx->mark_ev[i] = nullptr; if ((e = hipEventCreate(&ctx->mark_ev[i])) != hipSuccess) rc = 1; }
    if (!rc) {
        for (int i = 0; i < n_marks; ++i) ctx->mark_op[i] = mark_ops[i];
        ctx->n_marks = n_marks;
        if (replay_begin(ctx, s)) rc = 2;
        if (!rc && replay_eager(ctx, s, d_in, d_out_logits, true)) rc = 2;
        ctx->n_marks = 0;
        if (replay_end(ctx, s) && !rc) rc = 2;
        if (!rc && (e = hipEventSynchronize(ctx->mark_ev[n_marks])) != hipSuccess) rc = 1;
        for (int i = 0; i < n_marks && !rc; ++i) if ((e = hipEventElapsedTime(&h_ms[i], ctx->mark_ev[i], ctx->mark_ev[i + 1])) != hipSuccess) rc = 1;
    }
    ctx->n_marks = 0;
    for (int i = 0; i <= n_marks; ++i) if (ctx->mark_ev[i]) { (void)hipEventDestroy(ctx->mark_ev[i]); ctx->mark_ev[i] = nullptr; }
    if (rc == 1) rt_set_error("forward_marks: %s", hipGetErrorString(e));
    return rc ? 1 : 0;
}

extern "C" int rtm3d_op_info(rtm3d_ctx* ctx, int i, double* flops, double* bytes, const char** name) {
    if (!ctx || i < 0 || i >= (int)ctx->ops.size()) RT_FAIL("op_info: index out of range");
    if (flops) *flops = ctx->ops[i].flops;
    if (bytes) *bytes = ctx->ops[i].bytes;
    if (name) *name = ctx->ops[i].name.c_str();
    return 0;
}

// A HIP stream whose kernels may only run on the first `n_cus` compute units (hipExtStreamCreateWithCUMask).
// Used for the latency-bound 3D decode: its few long-lived wavefronts then never sit on the CUs the
// MFMA convolutions of the next batch need whole (LDS/VGPR co-residency), see rtm3d_amd/pipeline.py.
extern "C" int rtm3d_stream_create_cumask(int device, int n_cus, void** stream) {
    if (!stream || n_cus < 1) RT_FAIL("stream_create_cumask: bad arguments");
    RT_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, device));
    const int total = prop.multiProcessorCount;
    if (n_cus > total) n_cus = total;
    const int words = (total + 31) / 32;
    std::vector<uint32_t> mask(words, 0u);
    for (int i = 0; i < n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    RT_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask.data()));
    *stream = (void*)s;
    return 0;
}

extern "C" int rtm3d_stream_destroy(void* stream) {
    if (stream) RT_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
