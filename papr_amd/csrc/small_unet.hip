// K5c: the whole U-Net render head as ONE call each way -- the reference's SmallUNet.forward (models/unet.py:206-258) in its shipped variant
// (transposed-convolution upsampling, single convolution per stage, no normalisation, no affine modulation, render_scale 1):
//
//   x (c_in) -inc-> x1 (128) -pool-> -down1-> x2 (256) -pool-> -down2-> x3 (512)
//   x3 -up1.up-> (256) ++ x2 -up1.conv-> y1 (256) -up2.up-> (128) ++ x1 -up2.conv-> y2 (128) -outc (1x1)-> out (n_classes)
//
// The kernels are the ones of conv.hip / unet.hip; what this file adds is the arrangement between them, which a layer-by-layer caller cannot have
// (the single-layer entry points spend a launch per layer on the input's maximum and the weight split, and torch glue on every seam: 83 launches
// between the attention tail's forward and backward kernels, 1.47 ms of a 10.9 ms training step; 50 here):
//   * one launch in front splits ALL 3x3 weights into their f16 planes (the mirrored data-gradient forms too when a backward pass follows) and
//     takes the input map's maximum; every other tensor maximum a layer needs is left by the kernel that PRODUCES the tensor (one atomicMax per
//     workgroup into a 256-word slot, papr_common.h), the skip concatenation's as the larger of its two producers' -- a convolution scales by a power of two >= max |x|, so a bound
//     serves;
//   * the skip concatenations are never copied: inc / down1 write x1 / x2 straight into the left half of the concatenated map, the transposed
//     convolutions into the right half (row stride = the concatenation's width); backwards the two halves of its gradient are read in place;
//   * ReLU masks and the sum of a skip tensor's two gradients ride in the kernels at the seams: the 1x1 head's and the transposed convolutions'
//     data-gradient epilogues mask by the ReLU output they flow into, and one pass does pool-backward + skip gradient + mask for x1 and x2;
//   * the transposed convolution's weight-gradient launch sums its own bias gradient (no statistics launch).
// Layout of `state` (what the backward pass reads again) and of the two scratch regions: unet_layout() below.
#include "unet_parts.h"

namespace {

constexpr int C1 = 128, C2 = 256, C3 = 512;      // models/unet.py:194-198
constexpr int N_SLOTS = 16;
// maximum slots (papr_common.h: PAPR_SLOT_W words each)
enum { S_IN = 0, S_X1, S_X2, S_X3, S_UP1, S_Y1, S_UP2, S_Y2, S_DY2, S_DCAT2, S_DY1, S_DCAT1, S_DX3, S_DX2, S_DX1 };

struct Layer3 { int c_in, c_out; };
struct Dims {
    int B, H, W, c_in, ncls;
    long M1, M2, M3;
    Layer3 L[5];
};
static Dims dims_of(int B, int H, int W, int c_in, int ncls) {
    Dims d;
    d.B = B; d.H = H; d.W = W; d.c_in = c_in; d.ncls = ncls;
    d.M1 = (long)B * H * W; d.M2 = d.M1 / 4; d.M3 = d.M1 / 16;
    d.L[0] = {c_in, C1}; d.L[1] = {C1, C2}; d.L[2] = {C2, C3}; d.L[3] = {C3, C2}; d.L[4] = {C2, C1};
    return d;
}
static size_t al(size_t n) { return (n + 255) / 256 * 256; }

// `state`: [16 maximum slots x 256 u32][forward planes x5][mirrored planes x5 (keep)][cat2 M1 x 256][cat1 M2 x 512][pool1 M2 x 128]
//          [pool2 M3 x 256][x3 M3 x 512][y1 M2 x 256][y2 M1 x 128][which1 M2 x 32 u32][which2 M3 x 64 u32][forward scratch: the split launches' partial sums]
struct Layout {
    size_t slots, planes_f[5], planes_b[5], cat2, cat1, pool1, pool2, x3, y1, y2, which1, which2, scratch, total;
};
static size_t plane_halfs(const Layer3& l, bool mirrored) {           // both planes of one weight
    const long n = mirrored ? l.c_in : l.c_out, c = mirrored ? l.c_out : l.c_in;
    return (size_t)2 * ((n + 127) / 128 * 128) * 9 * c;
}
static size_t fwd_partial_bytes(const Dims& d) {
    const long M[5] = {d.M1, d.M2, d.M3, d.M2, d.M1};
    size_t mx = 0;
    for (int i = 0; i < 5; ++i) {
        const int sp = papr_i_conv_splits(M[i], d.L[i].c_in, d.L[i].c_out);
        const size_t b = sp > 1 ? (size_t)sp * M[i] * d.L[i].c_out * sizeof(float) : 0;
        mx = b > mx ? b : mx;
    }
    return mx;
}
static Layout unet_layout(const Dims& d, bool keep) {
    Layout y;
    size_t o = 0;
    y.slots = o; o += al((size_t)N_SLOTS * PAPR_SLOT_W * sizeof(unsigned));
    for (int i = 0; i < 5; ++i) { y.planes_f[i] = o; o += al(plane_halfs(d.L[i], false) * sizeof(_Float16)); }
    for (int i = 0; i < 5; ++i) { y.planes_b[i] = o; o += keep ? al(plane_halfs(d.L[i], true) * sizeof(_Float16)) : 0; }
    y.cat2 = o; o += al((size_t)d.M1 * 2 * C1 * 4);
    y.cat1 = o; o += al((size_t)d.M2 * 2 * C2 * 4);
    y.pool1 = o; o += al((size_t)d.M2 * C1 * 4);
    y.pool2 = o; o += al((size_t)d.M3 * C2 * 4);
    y.x3 = o; o += al((size_t)d.M3 * C3 * 4);
    y.y1 = o; o += al((size_t)d.M2 * C2 * 4);
    y.y2 = o; o += al((size_t)d.M1 * C1 * 4);
    y.which1 = o; o += keep ? al((size_t)d.M2 * (C1 / 4) * 4) : 0;
    y.which2 = o; o += keep ? al((size_t)d.M3 * (C2 / 4) * 4) : 0;
    y.scratch = o; o += al(fwd_partial_bytes(d));
    y.total = o;
    return y;
}

// backward workspace: the gradient maps and the reductions' partial sums
struct BwdLayout { size_t d_y2, d_cat2, d_y1, d_cat1, d_x3, d_pool2, d_x2, d_pool1, d_x1, partial, total; };
static BwdLayout bwd_layout(const Dims& d) {
    BwdLayout y;
    size_t o = 0;
    y.d_y2 = o; o += al((size_t)d.M1 * C1 * 4);
    y.d_cat2 = o; o += al((size_t)d.M1 * 2 * C1 * 4);
    y.d_y1 = o; o += al((size_t)d.M2 * C2 * 4);
    y.d_cat1 = o; o += al((size_t)d.M2 * 2 * C2 * 4);
    y.d_x3 = o; o += al((size_t)d.M3 * C3 * 4);
    y.d_pool2 = o; o += al((size_t)d.M3 * C2 * 4);
    y.d_x2 = o; o += al((size_t)d.M2 * C2 * 4);
    y.d_pool1 = o; o += al((size_t)d.M2 * C1 * 4);
    y.d_x1 = o; o += al((size_t)d.M1 * C1 * 4);
    // one region for whichever reduction runs (they are serial on the stream): data-gradient split launches, weight-gradient chunks
    const long M[5] = {d.M1, d.M2, d.M3, d.M2, d.M1};
    size_t mx = papr_conv1x1_bwd_workspace_bytes(d.M1, C1, d.ncls);
    for (int i = 0; i < 5; ++i) {
        const size_t w = papr_i_conv3x3_wgrad_partial_bytes(M[i], d.L[i].c_in, d.L[i].c_out);
        mx = w > mx ? w : mx;
        const int sp = papr_i_conv_splits(M[i], d.L[i].c_out, d.L[i].c_in);       // (the data-gradient: channels exchanged)
        const size_t b = sp > 1 ? (size_t)sp * M[i] * d.L[i].c_in * sizeof(float) : 0;
        mx = b > mx ? b : mx;
    }
    const size_t u1 = papr_i_upconv_wgrad_bytes(d.M3, C3, C2), u2 = papr_i_upconv_wgrad_bytes(d.M2, C2, C1);
    mx = u1 > mx ? u1 : mx;
    mx = u2 > mx ? u2 : mx;
    y.partial = o; o += al(mx);
    y.total = o;
    return y;
}

static int check_desc(const char* who, const papr_unet_desc* u) {
    PAPR_REQUIRE(u, "%s: null descriptor", who);
    PAPR_REQUIRE(u->B >= 1 && u->H >= 4 && u->W >= 4 && u->H % 4 == 0 && u->W % 4 == 0, "%s: B %d, H %d, W %d (H and W multiples of 4: two exact poolings)", who, u->B, u->H,
                 u->W);
    PAPR_REQUIRE(u->c_in >= 32 && u->c_in % 32 == 0 && u->n_classes >= 1 && u->n_classes <= 4, "%s: c_in %d (multiple of 32), n_classes %d (1 .. 4)", who, u->c_in,
                 u->n_classes);
    PAPR_REQUIRE((long)u->B * u->H * u->W * 4 * C3 < (1L << 31), "%s: map too large for 32-bit pixel arithmetic", who);
    for (int i = 0; i < 5; ++i) PAPR_REQUIRE(u->conv_w[i] && u->conv_b[i], "%s: 3x3 layer %d: null weight / bias", who, i);
    for (int i = 0; i < 2; ++i) PAPR_REQUIRE(u->up_w[i] && u->up_b[i], "%s: transposed convolution %d: null weight / bias", who, i);
    PAPR_REQUIRE(u->out_w && u->out_b, "%s: 1x1 head: null weight / bias", who);
    return 0;
}

}  // namespace

extern "C" size_t papr_small_unet_state_bytes(int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t keep) {
    return unet_layout(dims_of(B, H, W, c_in, 3), keep != 0).total;
}
extern "C" size_t papr_small_unet_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t n_classes) {
    return bwd_layout(dims_of(B, H, W, c_in, n_classes)).total;
}

extern "C" int papr_small_unet_fwd(const papr_unet_desc* u, const float* x, float* out, void* state, int32_t keep, papr_stream_t stream) {
    if (int rc = check_desc("papr_small_unet_fwd", u)) return rc;
    PAPR_REQUIRE(x && out && state, "papr_small_unet_fwd: null pointer");
    hipStream_t s = as_stream(stream);
    const Dims d = dims_of(u->B, u->H, u->W, u->c_in, u->n_classes);
    const Layout y = unet_layout(d, keep != 0);
    char* base = static_cast<char*>(state);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    unsigned* slots = reinterpret_cast<unsigned*>(base + y.slots);
    float *cat2 = F(y.cat2), *cat1 = F(y.cat1), *pool1 = F(y.pool1), *pool2 = F(y.pool2), *x3 = F(y.x3), *y1 = F(y.y1), *y2 = F(y.y2);
    unsigned* which1 = keep ? reinterpret_cast<unsigned*>(base + y.which1) : nullptr;
    unsigned* which2 = keep ? reinterpret_cast<unsigned*>(base + y.which2) : nullptr;
    float* partial = F(y.scratch);
    const bool one = u->one_product != 0;

    // ---- one launch: the input's maximum, the slots, every weight's planes
    PaprSplitJob jobs[PAPR_UNET_MAX_JOBS];
    _Float16 *ph[5], *pl[5];
    int nj = 0;
    for (int i = 0; i < 5; ++i) {
        const long n_pad = (d.L[i].c_out + 127) / 128 * 128;
        ph[i] = reinterpret_cast<_Float16*>(base + y.planes_f[i]);
        pl[i] = ph[i] + n_pad * 9 * d.L[i].c_in;
        const int64_t* st = u->conv_w_stride[i];
        jobs[nj++] = PaprSplitJob{u->conv_w[i], d.L[i].c_out, d.L[i].c_in, st[0], st[1], st[2], st[3], 0, ph[i], pl[i]};
    }
    if (keep)
        for (int i = 0; i < 5; ++i) {                 // the data-gradient's weight: channels exchanged, taps mirrored
            const long n_pad = (d.L[i].c_in + 127) / 128 * 128;
            _Float16* h = reinterpret_cast<_Float16*>(base + y.planes_b[i]);
            const int64_t* st = u->conv_w_stride[i];
            jobs[nj++] = PaprSplitJob{u->conv_w[i], d.L[i].c_in, d.L[i].c_out, st[1], st[0], st[2], st[3], 1, h, h + n_pad * 9 * d.L[i].c_out};
        }
    if (int rc = papr_i_unet_prep(x, d.M1 * d.c_in / 4, slots + S_IN * PAPR_SLOT_W, slots, N_SLOTS, jobs, nj, s)) return rc;

    auto S = [&](int slot) { return slots + slot * PAPR_SLOT_W; };
    auto conv = [&](int i, const float* in, int Hh, int Ww, int xslot, int xslot2, float* o, int ldo, int oslot) {
        PaprConvLaunch c{};
        c.x = in; c.B = d.B; c.H = Hh; c.W = Ww; c.c_in = d.L[i].c_in;
        c.w_hi = ph[i]; c.w_lo = pl[i];
        c.bias = u->conv_b[i]; c.c_out = d.L[i].c_out; c.relu = 1;
        c.out = o; c.ldo = ldo;
        c.xmax = S(xslot); c.n_xmax = PAPR_SLOT_W; c.xmax2 = xslot2 >= 0 ? S(xslot2) : nullptr; c.out_max = oslot >= 0 ? S(oslot) : nullptr;
        c.partial = partial; c.one_product = one;
        return papr_i_conv3x3(c, s);
    };
    const int H = d.H, W = d.W;
    // inc -> x1 = cat2[:, :128]
    if (int rc = conv(0, x, H, W, S_IN, -1, cat2, 2 * C1, S_X1)) return rc;
    // pool -> down1 -> x2 = cat1[:, :256]          (max |pool(x)| <= max |x|: the pooled maps reuse their sources' slots)
    if (int rc = papr_i_maxpool2_fwd(cat2, 2 * C1, d.B, H, W, C1, pool1, which1, s)) return rc;
    if (int rc = conv(1, pool1, H / 2, W / 2, S_X1, -1, cat1, 2 * C2, S_X2)) return rc;
    // pool -> down2 -> x3
    if (int rc = papr_i_maxpool2_fwd(cat1, 2 * C2, d.B, H / 2, W / 2, C2, pool2, which2, s)) return rc;
    if (int rc = conv(2, pool2, H / 4, W / 4, S_X2, -1, x3, C3, S_X3)) return rc;
    // up1: transposed convolution into cat1[:, 256:], 3x3 over the concatenation -> y1
    if (int rc = papr_i_upconv_fwd(x3, d.B, H / 4, W / 4, C3, u->up_w[0], u->up_b[0], C2, cat1 + C2, 2 * C2, S(S_UP1), one, s)) return rc;
    if (int rc = conv(3, cat1, H / 2, W / 2, S_X2, S_UP1, y1, C2, S_Y1)) return rc;
    // up2 -> y2
    if (int rc = papr_i_upconv_fwd(y1, d.B, H / 2, W / 2, C2, u->up_w[1], u->up_b[1], C1, cat2 + C1, 2 * C1, S(S_UP2), one, s)) return rc;
    if (int rc = conv(4, cat2, H, W, S_X1, S_UP2, y2, C1, -1)) return rc;       // (y2 only feeds the fp32 1x1 head: no scale needed)
    // the 1x1 head
    return papr_conv1x1_fwd(y2, d.M1, C1, u->out_w, u->out_b, d.ncls, out, stream);
}

extern "C" int papr_small_unet_bwd(const papr_unet_desc* u, const float* x, const float* d_out, void* state, float* d_x, const papr_unet_grads* g, void* workspace,
                                   papr_stream_t stream) {
    if (int rc = check_desc("papr_small_unet_bwd", u)) return rc;
    PAPR_REQUIRE(x && d_out && state && g && workspace, "papr_small_unet_bwd: null pointer");
    for (int i = 0; i < 5; ++i) PAPR_REQUIRE(g->conv_w[i] && g->conv_b[i], "papr_small_unet_bwd: 3x3 layer %d: null gradient buffer", i);
    for (int i = 0; i < 2; ++i) PAPR_REQUIRE(g->up_w[i] && g->up_b[i], "papr_small_unet_bwd: transposed convolution %d: null gradient buffer", i);
    PAPR_REQUIRE(g->out_w && g->out_b, "papr_small_unet_bwd: 1x1 head: null gradient buffer");
    hipStream_t s = as_stream(stream);
    const Dims d = dims_of(u->B, u->H, u->W, u->c_in, u->n_classes);
    const Layout y = unet_layout(d, true);
    const BwdLayout b = bwd_layout(d);
    char* base = static_cast<char*>(state);
    char* wb = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    auto G = [&](size_t off) { return reinterpret_cast<float*>(wb + off); };
    unsigned* slots = reinterpret_cast<unsigned*>(base + y.slots);
    float *cat2 = F(y.cat2), *cat1 = F(y.cat1), *pool1 = F(y.pool1), *pool2 = F(y.pool2), *x3 = F(y.x3), *y1 = F(y.y1), *y2 = F(y.y2);
    const unsigned* which1 = reinterpret_cast<const unsigned*>(base + y.which1);
    const unsigned* which2 = reinterpret_cast<const unsigned*>(base + y.which2);
    float *d_y2 = G(b.d_y2), *d_cat2 = G(b.d_cat2), *d_y1 = G(b.d_y1), *d_cat1 = G(b.d_cat1), *d_x3 = G(b.d_x3), *d_pool2 = G(b.d_pool2), *d_x2 = G(b.d_x2),
          *d_pool1 = G(b.d_pool1), *d_x1 = G(b.d_x1);
    float* partial = G(b.partial);
    const bool one = u->one_product != 0;
    const int H = d.H, W = d.W;

    // the data-gradient of 3x3 layer i: the same kernel over the mirrored planes, no bias, no activation
    auto S = [&](int slot) { return slots + slot * PAPR_SLOT_W; };
    auto dgrad = [&](int i, const float* dy, int Hh, int Ww, int dyslot, float* o, int oslot) {
        const long n_pad = (d.L[i].c_in + 127) / 128 * 128;
        const _Float16* h = reinterpret_cast<const _Float16*>(base + y.planes_b[i]);
        PaprConvLaunch c{};
        c.x = dy; c.B = d.B; c.H = Hh; c.W = Ww; c.c_in = d.L[i].c_out;
        c.w_hi = h; c.w_lo = h + n_pad * 9 * d.L[i].c_out;
        c.bias = nullptr; c.c_out = d.L[i].c_in; c.relu = 0;
        c.out = o; c.ldo = d.L[i].c_in;
        c.xmax = S(dyslot); c.n_xmax = PAPR_SLOT_W; c.out_max = oslot >= 0 ? S(oslot) : nullptr;
        c.partial = partial; c.one_product = one;
        return papr_i_conv3x3(c, s);
    };
    auto wgrad = [&](int i, const float* dy, const float* in, int Hh, int Ww, int dyslot, int xslot, int xslot2) {
        return papr_i_conv3x3_wgrad(dy, in, d.B, Hh, Ww, d.L[i].c_in, d.L[i].c_out, g->conv_w[i], g->conv_b[i], S(dyslot), S(xslot), xslot2 >= 0 ? S(xslot2) : nullptr,
                                    partial, one, s);
    };

    // 1x1 head: d_y2 = (d_out w) * (y2 > 0)
    if (int rc = papr_i_conv1x1_bwd(d_out, y2, d.M1, C1, u->out_w, d.ncls, y2, d_y2, S(S_DY2), g->out_w, g->out_b, partial, s)) return rc;
    // up2.conv (over cat2 = [x1 | up2.up(y1)])
    if (int rc = dgrad(4, d_y2, H, W, S_DY2, d_cat2, S_DCAT2)) return rc;
    if (int rc = wgrad(4, d_y2, cat2, H, W, S_DY2, S_X1, S_UP2)) return rc;
    // up2.up: its output is the right half of cat2; its input y1 is up1.conv's ReLU output
    if (int rc = papr_i_upconv_dgrad(d_cat2 + C1, 2 * C1, d.B, H / 2, W / 2, C2, u->up_w[1], C1, y1, d_y1, S(S_DY1), one, s)) return rc;
    if (int rc = papr_i_upconv_wgrad(d_cat2 + C1, 2 * C1, y1, d.B, H / 2, W / 2, C2, C1, S(S_Y1), S(S_DCAT2), g->up_w[1], g->up_b[1], partial, one, s)) return rc;
    // up1.conv (over cat1 = [x2 | up1.up(x3)])
    if (int rc = dgrad(3, d_y1, H / 2, W / 2, S_DY1, d_cat1, S_DCAT1)) return rc;
    if (int rc = wgrad(3, d_y1, cat1, H / 2, W / 2, S_DY1, S_X2, S_UP1)) return rc;
    // up1.up: input x3 = down2's ReLU output
    if (int rc = papr_i_upconv_dgrad(d_cat1 + C2, 2 * C2, d.B, H / 4, W / 4, C3, u->up_w[0], C2, x3, d_x3, S(S_DX3), one, s)) return rc;
    if (int rc = papr_i_upconv_wgrad(d_cat1 + C2, 2 * C2, x3, d.B, H / 4, W / 4, C3, C2, S(S_X3), S(S_DCAT1), g->up_w[0], g->up_b[0], partial, one, s)) return rc;
    // down2
    if (int rc = dgrad(2, d_x3, H / 4, W / 4, S_DX3, d_pool2, -1)) return rc;
    if (int rc = wgrad(2, d_x3, pool2, H / 4, W / 4, S_DX3, S_X2, -1)) return rc;
    // x2 feeds the pooling and the skip: d_x2 = (pool-backward(d_pool2) + d_cat1[:, :256]) * (x2 > 0)
    if (int rc = papr_i_maxpool2_bwd_fused(d_pool2, which2, d.B, H / 2, W / 2, C2, d_cat1, 2 * C2, cat1, 2 * C2, d_x2, S(S_DX2), s)) return rc;
    // down1
    if (int rc = dgrad(1, d_x2, H / 2, W / 2, S_DX2, d_pool1, -1)) return rc;
    if (int rc = wgrad(1, d_x2, pool1, H / 2, W / 2, S_DX2, S_X1, -1)) return rc;
    if (int rc = papr_i_maxpool2_bwd_fused(d_pool1, which1, d.B, H, W, C1, d_cat2, 2 * C1, cat2, 2 * C1, d_x1, S(S_DX1), s)) return rc;
    // inc
    if (d_x)
        if (int rc = dgrad(0, d_x1, H, W, S_DX1, d_x, -1)) return rc;
    return wgrad(0, d_x1, x, H, W, S_DX1, S_IN, -1);
}
