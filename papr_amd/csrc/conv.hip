// K5: 3x3 convolution (stride 1, zero padding 1) over an NHWC map as an implicit GEMM on the f16 matrix pipe with split
// fp32 operands -- the 3x3 layers of the U-Net render head (reference models/unet.py:16-33 DoubleConv with single=True,
// applied in SmallUNet, :182-258).
//
//   out[p][n] = act( bias[n] + sum_{tap, c} x[p + off(tap)][c] * w[n][tap][c] )        p = pixel, zero outside the image
//
// is a GEMM with M = B*H*W pixels, N = C_out, K = 9*C_in whose A rows are gathered: k-slab s (32 channels of one tap)
// of pixel row p is 128 contiguous bytes of the row of pixel p + off(tap) (NHWC keeps a pixel's channels together).
// Arithmetic as in gemm.hip / chain.hip: a * b ~ hi_a hi_b + hi_a lo_b + lo_a hi_b with f16 halves, fp32 accumulation.
// The nine taps of an output pixel read nine different input pixels, so the power-of-two scale that brings the
// activations into f16 range is one per TENSOR (max |x| from papr_tensor_absmax): an element 2^-11 below the maximum
// still keeps 22 bits, smaller ones lose absolute precision of 2^-39 of the maximum -- far below the fp32 rounding of a
// 288..4608-term sum (the reference's own convolutions run in TF32 on its hardware: 10-bit operands).
// The same kernel is the data-gradient: d_in = conv(d_out, w'), w'[c][tap][n] = w[n][flipped tap][c] (host: ops.py).
//
// Shape: 256 threads = 2 x 2 waves of 64 pixels x 64 channels over a 128 x 128 tile; both operands go through LDS (40 KB:
// [A hi | A lo | W hi | W lo] x 128 rows x 32 k, pitch 40 halfs), single-buffered with the next slab already in
// registers, so three workgroups share a CU and overlap each other's barriers.  Small maps (few tiles) deal the nine
// taps to 3 or 9 workgroups per tile, whose partial sums a second kernel adds in a fixed order with bias and ReLU.  W arrives pre-split as row-major
// f16 planes (papr_conv3x3_prepare_weight).  The MFMA takes the W fragment as its row operand: a lane owns one pixel and
// its registers are runs of four consecutive output channels (16-byte stores, bias / ReLU in registers).
#include "papr_common.h"
#include "h3_common.h"
#include "unet_parts.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float comp4c(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

constexpr int CV_BM = 128, CV_BN = 128, CV_BK = 32, CV_HP = CV_BK + 8;     // tile, k-slab, LDS row pitch in halfs
constexpr int CV_PLANE = 128 * CV_HP;                                     // halfs per plane
constexpr size_t CV_LDS_BYTES = (size_t)4 * CV_PLANE * sizeof(_Float16);

struct ConvArgs {
    const float* x; int B, H, W, C;
    const _Float16* w_hi; const _Float16* w_lo; long K;       // planes [N_pad][K], K = 9 C
    const float* bias; float* out; int N; int relu;
    const unsigned* xmax_bits; int n_xmax; const unsigned* xmax_bits2;       // max |x|: one word (n_xmax 1) or a slot (PAPR_SLOT_W), and a second slot or null
    int ldo;                                     // row stride of out (>= N: a channel slice of a wider map)
    unsigned* out_max;                           // or null (splits == 1): the slot that receives max |out|
    int splits; float* partial;                  // splits > 1: blockIdx.z sums 9 / splits taps into partial[z] (M, N), no bias / act
};

// ONE: a single f16 product per fp32 product (hi planes only) -- the arithmetic of the reference's fp16 autocast (`use_amp: true`, models/unet.py:212),
// with fp32 accumulation and fp32 maps in and out
template <bool ONE>
__global__ __launch_bounds__(256, 3) void conv3x3_h3_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Ah = reinterpret_cast<_Float16*>(smem);
    _Float16* Al = Ah + CV_PLANE;
    _Float16* Wh = Al + CV_PLANE;
    _Float16* Wl = Wh + CV_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const long M = (long)p.B * p.H * p.W;
    const long m0 = (long)blockIdx.x * CV_BM;
    const int n0 = blockIdx.y * CV_BN;
    const int slabs_per_tap = p.C / CV_BK;
    const int s_begin = blockIdx.z * (9 / p.splits) * slabs_per_tap, s_end = s_begin + (9 / p.splits) * slabs_per_tap;

    // scale of the whole input tensor: max -> [2^13, 2^14)
    unsigned mb = papr_slot_max(p.xmax_bits, p.n_xmax);
    if (p.xmax_bits2) { const unsigned o = papr_slot_max(p.xmax_bits2, PAPR_SLOT_W); mb = o > mb ? o : mb; }
    const int ea = mb ? (int)((mb >> 23) & 0xff) : 127 + 13;
    const float x_scale = pow2_from_biased(127 + 13 - (ea - 127)), x_inv = pow2_from_biased(127 - 13 + (ea - 127));

    // A slab copy: thread t carries rows t/8 + 32 q (q < 4), floats 4 (t % 8) .. + 3 of the 32-channel slab
    const int a_kq = (tid & 7) * 4;
    int py[4], px[4];
    long pbase[4];                                // element offset of the thread's pixel rows (clamped), -1: past the map
    bool prow_ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        long m = m0 + (tid >> 3) + 32 * q;
        prow_ok[q] = m < M;
        m = prow_ok[q] ? m : M - 1;
        const int rem = (int)(m % ((long)p.H * p.W));
        py[q] = rem / p.W; px[q] = rem - py[q] * p.W;
        pbase[q] = m * p.C;
    }
    // W slab copy: chunk c = t + 256 q (q < 2): row c / 4, halfs 8 (c % 4) .. + 7
    struct SlabRegs { float4 a[4]; bool ok[4]; half8 wh[2], wl[2]; };
    // (slabs are asked for in order: the tap and the channel slab of the next one are counters -- a division by the run-time slabs_per_tap per
    //  slab was a third of the loop's 138 scalar instructions, and with one product per fp32 product the loop is bound by them)
    int nx_tap = s_begin / slabs_per_tap, nx_cs = s_begin - nx_tap * slabs_per_tap;
    auto load_slab = [&](int s, SlabRegs& r) {
        const int tap = nx_tap, c0 = nx_cs * CV_BK;
        if (++nx_cs == slabs_per_tap) { nx_cs = 0; ++nx_tap; }
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = py[q] + dy, xx = px[q] + dx;
            r.ok[q] = prow_ok[q] & ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);      // (no short circuits: each was a branch around the next compare)
            const long off = r.ok[q] ? pbase[q] + ((long)dy * p.W + dx) * p.C : pbase[q];
            r.a[q] = *reinterpret_cast<const float4*>(p.x + off + c0 + a_kq);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = tid + 256 * q;
            const long o = (long)(n0 + (c >> 2)) * p.K + (long)s * CV_BK + (c & 3) * 8;
            r.wh[q] = *reinterpret_cast<const half8*>(p.w_hi + o);
            if (!ONE) r.wl[q] = *reinterpret_cast<const half8*>(p.w_lo + o);
        }
    };
    auto store_slab = [&](const SlabRegs& r) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half4 hi, lo;
            split4(r.a[q], r.ok[q] ? x_scale : 0.f, hi, lo);
            const int off = ((tid >> 3) + 32 * q) * CV_HP + a_kq;
            *reinterpret_cast<half4*>(Ah + off) = hi;
            if (!ONE) *reinterpret_cast<half4*>(Al + off) = lo;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = tid + 256 * q;
            const int off = (c >> 2) * CV_HP + (c & 3) * 8;
            *reinterpret_cast<half8*>(Wh + off) = r.wh[q];
            if (!ONE) *reinterpret_cast<half8*>(Wl + off) = r.wl[q];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag = (lane & 31) * CV_HP + 8 * (lane >> 5);
    auto multiply = [&]() {
#pragma unroll
        for (int ks = 0; ks < CV_BK; ks += 16) {
            half8 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = (wm * 64 + i * 32) * CV_HP + frag + ks;
                ah[i] = *reinterpret_cast<const half8*>(Ah + o);
                if (!ONE) al[i] = *reinterpret_cast<const half8*>(Al + o);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = (wn * 64 + j * 32) * CV_HP + frag + ks;
                const half8 wh = *reinterpret_cast<const half8*>(Wh + o);
                half8 wl;
                if (!ONE) wl = *reinterpret_cast<const half8*>(Wl + o);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (!ONE) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al[i], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah[i], acc[i][j], 0, 0, 0);
                }
            }
        }
    };
    // One register set: slab s+1 is requested before slab s is multiplied.  (A second set -- two slabs ahead -- measured
    // no gain on small maps and -5 % on large ones, where it costs the third workgroup of the CU its registers.)
    SlabRegs r0;
    load_slab(s_begin, r0);
    store_slab(r0);
    lds_barrier();
    for (int s = s_begin; s < s_end; ++s) {
        if (s + 1 < s_end) load_slab(s + 1, r0);
        multiply();
        lds_barrier();                               // everybody has read slab s
        if (s + 1 < s_end) store_slab(r0);
        lds_barrier();
    }

    // epilogue: lane = pixel, registers = runs of four channels
    const int hh = lane >> 5;
    float omax = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long m = m0 + wm * 64 + i * 32 + (lane & 31);
        if (m >= M) continue;
        float* orow = p.splits > 1 ? p.partial + ((long)blockIdx.z * M + m) * p.N : p.out + m * p.ldo;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wn * 64 + j * 32 + 8 * g + 4 * hh;
                if (col >= p.N) continue;
                float4 b4 = (p.bias && p.splits == 1) ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 r = make_float4(__builtin_fmaf(acc[i][j][4 * g], x_inv, b4.x), __builtin_fmaf(acc[i][j][4 * g + 1], x_inv, b4.y),
                                       __builtin_fmaf(acc[i][j][4 * g + 2], x_inv, b4.z), __builtin_fmaf(acc[i][j][4 * g + 3], x_inv, b4.w));
                if (p.relu && p.splits == 1) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
                omax = fmaxf(omax, fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w))));
                *reinterpret_cast<float4*>(orow + col) = r;
            }
    }
    if (p.out_max && p.splits == 1) papr_wg_max_to_slot(p.out_max, omax);
}

// planes hi / lo (N_pad, 9 C) <- weight element (n, ky, kx, c) at w[n sn + c sc + ky sky + kx skx]; flip: taps mirrored
// (the data-gradient's weight); rows N .. N_pad-1 zero.  One thread per four consecutive c.
struct ConvSplitArgs { const float* w; int N, C; long sn, sc, sky, skx; int flip; long total4; _Float16* hi; _Float16* lo; };
__device__ __forceinline__ void conv_split_weight_block(const ConvSplitArgs& a, long block) {
    const float* __restrict__ w = a.w; const int N = a.N, C = a.C, flip = a.flip; const long sn = a.sn, sc = a.sc, sky = a.sky, skx = a.skx, total4 = a.total4;
    _Float16* __restrict__ hi = a.hi; _Float16* __restrict__ lo = a.lo;
    const long e = block * 256 + threadIdx.x;
    if (e >= total4) return;
    const long K = 9L * C;
    const long n = e * 4 / K;
    const int k = (int)(e * 4 - n * K), tap = k / C, c = k - tap * C;
    int ky = tap / 3, kx = tap - ky * 3;
    if (flip) { ky = 2 - ky; kx = 2 - kx; }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) {
        const float* q = w + n * sn + ky * sky + kx * skx + c * sc;
        v = sc == 1 ? *reinterpret_cast<const float4*>(q) : make_float4(q[0], q[sc], q[2 * sc], q[3 * sc]);
    }
    half4 h, l;
    split4(v, 1.0f, h, l);
    reinterpret_cast<half4*>(hi)[e] = h;
    reinterpret_cast<half4*>(lo)[e] = l;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the same layer: dW[n][tap][c] = sum over pixels p of dY[p][n] * X[p + off(tap)][c].
// The reduction runs over pixels, so both operands go into LDS TRANSPOSED (rows = channels, k = pixels): a thread loads
// a 4-pixel x 4-channel block (four float4, one per pixel) and writes its columns as 4-pixel runs (the trick of
// gemm_tn_h3).  Channel 4 q + j of the 128-channel block lives in LDS row q + 32 j, so the lanes of a write hit
// consecutive rows; the epilogue undoes the permutation.  One workgroup = one tap, one 128 x 128 (n, c) tile and one
// CHUNK of the pixels (few output tiles exist: 9 ... 72 per layer); the chunks' partial tiles meet in a fixed order in
// conv_wgrad_reduce_kernel.  Scales: one power of two per tensor for dY and for X.
struct ConvWArgs {
    const float* dy; const float* x; int B, H, W, N, C;
    const unsigned* dymax_bits; const unsigned* xmax_bits; const unsigned* xmax_bits2; int n_max;      // one word each (n_max 1) or slots (PAPR_SLOT_W); xmax_bits2
                                                                                                        // or null: x concatenated from two producers
    float* partial;                              // [chunk][N][9][C]
    float* partial_b;                            // [chunk][N] column sums of dY (bias gradient), or null
    long px_per_chunk;                           // multiple of 32
};

// CT = channels of x per tile: 128, or 32 for the network's first layer (32 input channels: a 128-wide tile would multiply
// three quarters of zeros) -- then the four waves stack along n, 32 x 32 each.
template <int CT, bool ONE>
__global__ __launch_bounds__(256, 3) void conv3x3_wgrad_h3_kernel(ConvWArgs p) {
    constexpr int NI = CT == 128 ? 2 : 1, NJ = NI, CQ = CT / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Gh = reinterpret_cast<_Float16*>(smem);
    _Float16* Gl = Gh + CV_PLANE;
    _Float16* Xh = Gl + CV_PLANE;
    _Float16* Xl = Xh + CV_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = CT == 128 ? wave >> 1 : wave, wn = CT == 128 ? wave & 1 : 0;
    const long M = (long)p.B * p.H * p.W;
    const int cb = (p.C + CT - 1) / CT;
    const int n0 = (blockIdx.y / cb) * 128, c0 = (blockIdx.y % cb) * CT;
    const int tap = blockIdx.z, dy_ = tap / 3 - 1, dx_ = tap - (tap / 3) * 3 - 1;
    const long pbeg = (long)blockIdx.x * p.px_per_chunk;
    long pend = pbeg + p.px_per_chunk;
    if (pend > M) pend = M;

    auto scale_of = [](unsigned mb, float& inv) {
        const int ea = mb ? (int)((mb >> 23) & 0xff) : 127 + 13;
        inv = pow2_from_biased(127 - 13 + (ea - 127));
        return pow2_from_biased(127 + 13 - (ea - 127));
    };
    float g_inv, x_inv;
    unsigned xmb = papr_slot_max(p.xmax_bits, p.n_max);
    if (p.xmax_bits2) { const unsigned o = papr_slot_max(p.xmax_bits2, p.n_max); xmb = o > xmb ? o : xmb; }
    const float g_scale = scale_of(papr_slot_max(p.dymax_bits, p.n_max), g_inv), x_scale = scale_of(xmb, x_inv);

    // block of the thread: channels 4 q .. 4 q + 3 (q = tid % 32), pixels 4 r .. 4 r + 3 of the 32-pixel slab (r = tid / 32)
    const int q = tid & 31, r = tid >> 5;
    const bool n_ok = n0 + 4 * q < p.N, c_ok = q < CQ && c0 + 4 * q < p.C;
    const int ncol = n_ok ? n0 + 4 * q : 0, ccol = c_ok ? c0 + 4 * q : 0;
    float4 rg[4], rx[4];
    bool okg[4], okx[4];
    // (y, x) of the thread's four pixels of the current slab, kept up to date by steps of 32 pixels (no division per slab)
    int py[4], px[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        long pix = pbeg + 4 * r + j;
        pix = pix < M ? pix : M - 1;
        const int rem = (int)(pix % ((long)p.H * p.W));
        py[j] = rem / p.W; px[j] = rem - py[j] * p.W;
    }
    auto load_slab = [&](long ps) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long pix = ps + 4 * r + j;
            okg[j] = n_ok & (pix < pend);
            const long pc = pix < M ? pix : M - 1;
            rg[j] = *reinterpret_cast<const float4*>(p.dy + pc * p.N + ncol);
            const int yy = py[j] + dy_, xx = px[j] + dx_;
            okx[j] = c_ok & (pix < pend) & ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);
            const long src = okx[j] ? pc + (long)dy_ * p.W + dx_ : pc;
            if (CT == 128 || q < CQ) rx[j] = *reinterpret_cast<const float4*>(p.x + src * p.C + ccol);
            px[j] += 32;                                  // the same thread's pixel of the next slab
            while (px[j] >= p.W) { px[j] -= p.W; if (++py[j] == p.H) py[j] = 0; }
        }
    };
    auto put = [&](const float4 (&v)[4], const bool (&ok)[4], float sc, _Float16* hi_plane, _Float16* lo_plane, int rq) {
        const float s0 = ok[0] ? sc : 0.f, s1 = ok[1] ? sc : 0.f, s2 = ok[2] ? sc : 0.f, s3 = ok[3] ? sc : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                    // channel 4 q + j: its four pixels
            const float4 col = make_float4(comp4c(v[0], j) * s0, comp4c(v[1], j) * s1, comp4c(v[2], j) * s2, comp4c(v[3], j) * s3);
            half4 hi, lo;
            split4(col, 1.0f, hi, lo);
            const int off = (q + rq * j) * CV_HP + 4 * r;
            *reinterpret_cast<half4*>(hi_plane + off) = hi;
            if (!ONE) *reinterpret_cast<half4*>(lo_plane + off) = lo;
        }
    };
    // bias gradient = column sums of dY: the workgroups of the centre tap and the first c tile add up their pixels
    const bool do_bias = p.partial_b != nullptr && tap == 4 && c0 == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto store_slab = [&]() {
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (okg[j]) { bsum.x += rg[j].x; bsum.y += rg[j].y; bsum.z += rg[j].z; bsum.w += rg[j].w; }
        }
        put(rg, okg, g_scale, Gh, Gl, 32);
        if (CT == 128 || q < CQ) put(rx, okx, x_scale, Xh, Xl, CQ);
    };

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frag = (lane & 31) * CV_HP + 8 * (lane >> 5);
    if (pbeg < pend) {
        load_slab(pbeg);
        store_slab();
        lds_barrier();
        for (long ps = pbeg; ps < pend; ps += 32) {
            if (ps + 32 < pend) load_slab(ps + 32);
#pragma unroll
            for (int ks = 0; ks < CV_BK; ks += 16) {
                half8 xh[NJ], xl[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int o = (wn * 64 + j * 32) * CV_HP + frag + ks;
                    xh[j] = *reinterpret_cast<const half8*>(Xh + o);
                    if (!ONE) xl[j] = *reinterpret_cast<const half8*>(Xl + o);
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int o = (wm * 32 * NI + i * 32) * CV_HP + frag + ks;
                    const half8 gh = *reinterpret_cast<const half8*>(Gh + o);
                    half8 gl;
                    if (!ONE) gl = *reinterpret_cast<const half8*>(Gl + o);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (!ONE) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, xh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xh[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
            lds_barrier();
            if (ps + 32 < pend) store_slab();
            lds_barrier();
        }
    }
    if (do_bias) {                                   // (the slab buffers are idle: the loop ended with a barrier)
        float4* bs = reinterpret_cast<float4*>(smem);
        bs[r * 32 + q] = bsum;
        __syncthreads();
        if (r == 0 && n_ok) {
            float4 t = bs[q];
#pragma unroll
            for (int g = 1; g < 8; ++g) { const float4 v = bs[g * 32 + q]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            *reinterpret_cast<float4*>(p.partial_b + (long)blockIdx.x * p.N + n0 + 4 * q) = t;
        }
    }
    // acc[i][j][e]: LDS rows (n) 64 wm + 32 i + (e & 3) + 8 (e >> 2) + 4 (lane >> 5), LDS row (c) 64 wn + 32 j + (lane & 31);
    // LDS row R holds channel 4 (R % 32) + R / 32 of the n block, 4 (R % CQ) + R / CQ of the c block
    float* out = p.partial + (long)blockIdx.x * p.N * 9 * p.C;
    const float inv = g_inv * x_inv;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int rc = 64 * wn + 32 * j + (lane & 31);
        const int c = c0 + 4 * (rc % CQ) + rc / CQ;
        if (c >= p.C) continue;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rn = 32 * NI * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int n = n0 + 4 * (rn & 31) + (rn >> 5);
                if (n < p.N) out[((long)n * 9 + tap) * p.C + c] = acc[i][j][e] * inv;
            }
    }
}

// d_w (N, 3, 3, C) contiguous = sum of the chunks' partial tiles, in chunk order
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float4* __restrict__ partial, int chunks, long n4, float4* __restrict__ out,
                                                                const float4* __restrict__ partial_b, int nb4, float4* __restrict__ out_b,
                                                                unsigned* __restrict__ stale2) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    // The two maximum slots this call's successors (16 calls on) will atomicMax into are cleared HERE, whatever maxima the
    // caller supplied: clearing only as a side effect of the absmax launches left a slot uncleared whenever a supplied
    // maximum spared its launch, and the scale of a later call would become a running maximum over history.
    if (stale2 && blockIdx.x == 0 && threadIdx.x == 0) { stale2[0] = 0u; stale2[1] = 0u; }
    if (out_b && e < nb4) {                          // bias gradient: the chunks' column sums
        float4 r = partial_b[e];
#pragma unroll 8
        for (int z = 1; z < chunks; ++z) { const float4 v = partial_b[(long)z * nb4 + e]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        out_b[e] = r;
    }
    if (e >= n4) return;
    float4 r = partial[e];
#pragma unroll 8                                   // (the loads of eight chunks in flight together: 34 -> 67 chunks in a row were a latency chain of as many round trips)
    for (int z = 1; z < chunks; ++z) {
        const float4 v = partial[(long)z * n4 + e];
        r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
    }
    out[e] = r;
}

// out = act(bias + partial[0] + partial[1] + ...): the tap groups of a split launch meet in a fixed order
__global__ __launch_bounds__(256) void conv_reduce_kernel(const float4* __restrict__ partial, int splits, long mn4, int n4,
                                                          const float4* __restrict__ bias, int relu, float4* __restrict__ out, int ldo4,
                                                          unsigned* __restrict__ out_max) {
    float mx = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < mn4; e += (long)gridDim.x * 256) {
        const long m = e / n4;
        const int c = (int)(e - m * n4);
        float4 r = bias ? bias[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 3
        for (int z = 0; z < splits; ++z) {
            const float4 v = partial[(long)z * mn4 + e];
            r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
        }
        if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w))));
        out[m * ldo4 + c] = r;
    }
    if (out_max) papr_wg_max_to_slot(out_max, mx);          // (uniform per launch)
}

__device__ __forceinline__ void tensor_absmax_block(const float4* __restrict__ x, long n4, unsigned* __restrict__ out, unsigned* __restrict__ stale,
                                                    long block, long blocks) {
    __shared__ float part[4];
    if (block == 0 && threadIdx.x == 0) *stale = 0u;       // the slot of 32 calls ago, for its next turn (no memset launch)
    float m = 0.f;
    for (long i = block * 256 + threadIdx.x; i < n4; i += blocks * 256) {
        const float4 v = x[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)                                   // |x| bit patterns order like unsigned ints
        atomicMax(out, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}
__global__ __launch_bounds__(256) void tensor_absmax_kernel(const float4* __restrict__ x, long n4, unsigned* __restrict__ out,
                                                            unsigned* __restrict__ stale) {
    tensor_absmax_block(x, n4, out, stale, blockIdx.x, gridDim.x);
}
// What a convolution call does in front of its matrix kernel, in ONE launch: the first nb_abs workgroups take the maximum of the input map,
// the others split the weight into its f16 planes (two independent 6-us launches before: 20 launches of a training step)
__global__ __launch_bounds__(256) void conv_prep_kernel(const float4* __restrict__ x, long n4, unsigned* __restrict__ out, unsigned* __restrict__ stale,
                                                        int nb_abs, ConvSplitArgs sp) {
    if ((int)blockIdx.x < nb_abs) tensor_absmax_block(x, n4, out, stale, blockIdx.x, nb_abs);
    else conv_split_weight_block(sp, (long)blockIdx.x - nb_abs);
}
// What a whole SmallUNet call does in front of its first layer, in ONE launch: partial maxima of the input map (no atomics: the first layer's
// kernel takes the largest of the 64), the call's maximum slots zeroed, and every 3x3 weight of the network -- the data-gradient's mirrored
// forms too when a backward pass will follow -- split into its f16 planes.
struct UnetPrepJobs { ConvSplitArgs job[PAPR_UNET_MAX_JOBS]; int first_block[PAPR_UNET_MAX_JOBS + 1]; int n_jobs; };
__global__ __launch_bounds__(256) void unet_prep_kernel(const float4* __restrict__ x, long n4, unsigned* __restrict__ in_partial, unsigned* __restrict__ slots,
                                                        int n_slots, UnetPrepJobs jobs) {
    const int b = blockIdx.x;
    if (b < PAPR_UNET_IN_PARTS) {
        __shared__ float part[4];
        if (b == 0)                                   // (the input's own slot is written below, words 0 .. 63 by these 64 workgroups, the rest stays zero)
            for (int i = threadIdx.x; i < n_slots * PAPR_SLOT_W; i += 256)
                if (slots + i < in_partial || slots + i >= in_partial + PAPR_UNET_IN_PARTS) slots[i] = 0u;
        float m = 0.f;
        for (long i = (long)b * 256 + threadIdx.x; i < n4; i += (long)PAPR_UNET_IN_PARTS * 256) {
            const float4 v = x[i];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) in_partial[b] = __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
        return;
    }
    const int sb = b - PAPR_UNET_IN_PARTS;
    int j = 0;
    while (j + 1 < jobs.n_jobs && sb >= jobs.first_block[j + 1]) ++j;
    conv_split_weight_block(jobs.job[j], (long)sb - jobs.first_block[j]);
}
// two maxima in one launch (the weight gradient's operands)
__global__ __launch_bounds__(256) void tensor_absmax2_kernel(const float4* __restrict__ x0, long n0, unsigned* __restrict__ out0, unsigned* __restrict__ stale0, int nb0,
                                                             const float4* __restrict__ x1, long n1, unsigned* __restrict__ out1, unsigned* __restrict__ stale1) {
    if ((int)blockIdx.x < nb0) tensor_absmax_block(x0, n0, out0, stale0, blockIdx.x, nb0);
    else tensor_absmax_block(x1, n1, out1, stale1, (long)blockIdx.x - nb0, (long)gridDim.x - nb0);
}

}  // namespace

extern "C" size_t papr_conv3x3_weight_halfs(int32_t c_out, int32_t c_in) {
    const long n_pad = (c_out + CV_BN - 1) / CV_BN * CV_BN;
    return (size_t)2 * n_pad * 9 * c_in;
}

// tap groups per launch: small maps leave most CUs without a tile, so their 9 taps are dealt to 3 or 9 workgroups
static int conv_splits(long M, int c_in, int c_out) {
    const long tiles = ((M + CV_BM - 1) / CV_BM) * ((c_out + CV_BN - 1) / CV_BN);
    if (c_in < 128 || tiles >= 400) return 1;
    return tiles * 3 >= 400 ? 3 : 9;
}
int papr_i_conv_splits(long M, int c_in, int c_out) { return conv_splits(M, c_in, c_out); }

extern "C" size_t papr_conv3x3_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out) {
    const long M = (long)B * H * W;
    const int sp = conv_splits(M, c_in, c_out);
    return 256 + papr_conv3x3_weight_halfs(c_out, c_in) * sizeof(_Float16) + (sp > 1 ? (size_t)sp * M * c_out * sizeof(float) : 0);
}

int papr_i_conv3x3(const PaprConvLaunch& c, hipStream_t s) {
    const long M = (long)c.B * c.H * c.W, K = 9L * c.c_in, n_pad = (c.c_out + CV_BN - 1) / CV_BN * CV_BN;
    ConvArgs a;
    a.x = c.x; a.B = c.B; a.H = c.H; a.W = c.W; a.C = c.c_in;
    a.w_hi = c.w_hi; a.w_lo = c.w_lo; a.K = K;
    a.bias = c.bias; a.out = c.out; a.N = c.c_out; a.relu = c.relu;
    a.xmax_bits = c.xmax; a.n_xmax = c.n_xmax; a.xmax_bits2 = c.xmax2;
    a.ldo = c.ldo; a.out_max = c.out_max;
    a.splits = conv_splits(M, c.c_in, c.c_out);
    a.partial = c.partial;
    PAPR_REQUIRE(a.splits == 1 || c.partial, "conv3x3: a split launch needs its partial buffer");
    if (papr_first_on_device(PAPR_ONCE_CONV)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_h3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CV_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_h3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CV_LDS_BYTES);
    }
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin2(11, M, c.c_out, (int)K, 4LL * M * (c.c_in + c.c_out), 2LL * M * c.c_out * K, s);
    const dim3 grid((unsigned)((M + CV_BM - 1) / CV_BM), (unsigned)(n_pad / CV_BN), (unsigned)a.splits);
    if (c.one_product) conv3x3_h3_kernel<true><<<grid, dim3(256), CV_LDS_BYTES, s>>>(a);
    else conv3x3_h3_kernel<false><<<grid, dim3(256), CV_LDS_BYTES, s>>>(a);
    PAPR_CHECK_LAUNCH("conv3x3_h3");
    if (a.splits > 1) {
        const long mn4 = M * c.c_out / 4;
        const long rb = (mn4 + 255) / 256;             // (with a maximum to leave: a bounded grid, one atomic per workgroup)
        conv_reduce_kernel<<<dim3((unsigned)(c.out_max && rb > 1024 ? 1024 : rb)), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(a.partial), a.splits, mn4, c.c_out / 4,
                                                                                     reinterpret_cast<const float4*>(c.bias), c.relu, reinterpret_cast<float4*>(c.out),
                                                                                     c.ldo / 4, c.out_max);
        PAPR_CHECK_LAUNCH("conv_reduce");
    }
    if (prof) papr_prof_end(s);
    return 0;
}

int papr_i_unet_prep(const float* x, long n4, unsigned* in_partial, unsigned* slots, int n_slots, const PaprSplitJob* jobs, int n_jobs, hipStream_t s) {
    PAPR_REQUIRE(n_jobs >= 1 && n_jobs <= PAPR_UNET_MAX_JOBS, "unet_prep: %d split jobs", n_jobs);
    UnetPrepJobs t;
    t.n_jobs = n_jobs;
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const PaprSplitJob& q = jobs[j];
        const long n_pad = (q.N + CV_BN - 1) / CV_BN * CV_BN, total4 = n_pad * 9L * q.C / 4;
        t.job[j] = ConvSplitArgs{q.w, q.N, q.C, q.sn, q.sc, q.sky, q.skx, q.flip, total4, q.hi, q.lo};
        t.first_block[j] = blocks;
        blocks += (int)((total4 + 255) / 256);
    }
    t.first_block[n_jobs] = blocks;
    unet_prep_kernel<<<dim3((unsigned)(PAPR_UNET_IN_PARTS + blocks)), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(x), n4, in_partial, slots, n_slots, t);
    PAPR_CHECK_LAUNCH("unet_prep");
    return 0;
}

extern "C" int papr_conv3x3_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t c_in, const float* w, int64_t w_stride_n,
                                int64_t w_stride_c, int64_t w_stride_ky, int64_t w_stride_kx, int32_t flip_taps, const float* bias,
                                int32_t c_out, int32_t relu, float* out, void* workspace, int32_t slot, papr_stream_t stream) {
    PAPR_REQUIRE(x && w && out && workspace, "papr_conv3x3_fwd: null pointer");
    PAPR_REQUIRE(slot >= 0 && slot < 64, "papr_conv3x3_fwd: slot %d outside 0 .. 63", slot);
    PAPR_REQUIRE(B >= 1 && H >= 1 && W >= 1 && c_in >= 32 && c_in % 32 == 0 && c_out >= 4 && c_out % 4 == 0,
                 "papr_conv3x3_fwd: B %d, H %d, W %d, c_in %d (multiple of 32), c_out %d (multiple of 4)", B, H, W, c_in, c_out);
    hipStream_t s = as_stream(stream);
    const long M = (long)B * H * W, K = 9L * c_in, n_pad = (c_out + CV_BN - 1) / CV_BN * CV_BN;
    unsigned* xmax = static_cast<unsigned*>(workspace) + slot;
    unsigned* stale = static_cast<unsigned*>(workspace) + ((slot + 32) & 63);
    _Float16* planes = reinterpret_cast<_Float16*>(static_cast<char*>(workspace) + 256);
    const long n4 = M * c_in / 4;
    const long want = (n4 + 256 * 16 - 1) / (256 * 16);
    const long total4 = n_pad * K / 4;
    const int nb_abs = (int)(want < 256 ? want : 256);
    ConvSplitArgs sp = {w, c_out, c_in, w_stride_n, w_stride_c, w_stride_ky, w_stride_kx, flip_taps, total4, planes, planes + n_pad * K};
    conv_prep_kernel<<<dim3((unsigned)(nb_abs + (total4 + 255) / 256)), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(x), n4, xmax, stale, nb_abs, sp);
    PAPR_CHECK_LAUNCH("conv_prep");
    PaprConvLaunch c{};
    c.x = x; c.B = B; c.H = H; c.W = W; c.c_in = c_in;
    c.w_hi = planes; c.w_lo = planes + n_pad * K;
    c.bias = bias; c.c_out = c_out; c.relu = relu; c.out = out; c.ldo = c_out;
    c.xmax = xmax; c.n_xmax = 1;
    c.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256 + papr_conv3x3_weight_halfs(c_out, c_in) * sizeof(_Float16));
    return papr_i_conv3x3(c, s);
}

// pixels per workgroup of the weight-gradient launch: enough chunks for ~600 workgroups, at least 8 slabs each
static long wgrad_px_per_chunk(long M, int c_in, int c_out) {
    const long tiles = (long)((c_out + 127) / 128) * ((c_in + 127) / 128) * 9;
    const long target = papr_switch(PAPR_SW_WGRAD_WGS) > 0 ? papr_switch(PAPR_SW_WGRAD_WGS) : 600;
    long chunks = (target + tiles - 1) / tiles;
    long px = (M + chunks - 1) / chunks;
    px = (px + 31) / 32 * 32;
    return px < 256 ? 256 : px;
}

extern "C" size_t papr_conv3x3_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out) {
    const long M = (long)B * H * W, px = wgrad_px_per_chunk(M, c_in, c_out), chunks = (M + px - 1) / px;
    return 256 + (size_t)chunks * c_out * (9 * c_in + 1) * sizeof(float);
}

size_t papr_i_conv3x3_wgrad_partial_bytes(long M, int c_in, int c_out) {
    const long px = wgrad_px_per_chunk(M, c_in, c_out), chunks = (M + px - 1) / px;
    return (size_t)chunks * c_out * (9 * c_in + 1) * sizeof(float);
}

// stale: the two maximum slots conv_wgrad_reduce_kernel clears for the stand-alone entry point's slot rotation (or null)
static int conv3x3_wgrad_launch(const float* d_y, const float* x, int B, int H, int W, int c_in, int c_out, float* d_w, float* d_b, const unsigned* dymax,
                                const unsigned* xmax, const unsigned* xmax2, int n_max, float* partial, unsigned* stale, bool one_product, hipStream_t s) {
    const long M = (long)B * H * W, px = wgrad_px_per_chunk(M, c_in, c_out), chunks = (M + px - 1) / px;
    ConvWArgs a;
    a.dy = d_y; a.x = x; a.B = B; a.H = H; a.W = W; a.N = c_out; a.C = c_in;
    a.dymax_bits = dymax; a.xmax_bits = xmax; a.xmax_bits2 = xmax2; a.n_max = n_max;
    a.partial = partial;
    a.partial_b = d_b ? a.partial + (size_t)chunks * c_out * 9 * c_in : nullptr;
    a.px_per_chunk = px;
    if (papr_first_on_device(PAPR_ONCE_CONV_WGRAD)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_h3_kernel<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CV_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_h3_kernel<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CV_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_h3_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CV_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_h3_kernel<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CV_LDS_BYTES);
    }
    const int ct = c_in <= 32 ? 32 : 128;
    const int tiles = ((c_out + 127) / 128) * ((c_in + ct - 1) / ct);
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin2(12, M, c_out, 9 * c_in, 4LL * M * (c_in + c_out), 2LL * M * c_out * 9 * c_in, s);
    const dim3 grid((unsigned)chunks, (unsigned)tiles, 9);
    if (ct == 32 && one_product) conv3x3_wgrad_h3_kernel<32, true><<<grid, dim3(256), CV_LDS_BYTES, s>>>(a);
    else if (ct == 32) conv3x3_wgrad_h3_kernel<32, false><<<grid, dim3(256), CV_LDS_BYTES, s>>>(a);
    else if (one_product) conv3x3_wgrad_h3_kernel<128, true><<<grid, dim3(256), CV_LDS_BYTES, s>>>(a);
    else conv3x3_wgrad_h3_kernel<128, false><<<grid, dim3(256), CV_LDS_BYTES, s>>>(a);
    PAPR_CHECK_LAUNCH("conv3x3_wgrad_h3");
    const long n4 = (long)c_out * 9 * c_in / 4;
    conv_wgrad_reduce_kernel<<<dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(a.partial), (int)chunks, n4, reinterpret_cast<float4*>(d_w),
                                                                                      reinterpret_cast<const float4*>(a.partial_b), c_out / 4, reinterpret_cast<float4*>(d_b), stale);
    PAPR_CHECK_LAUNCH("conv_wgrad_reduce");
    if (prof) papr_prof_end(s);
    return 0;
}

int papr_i_conv3x3_wgrad(const float* d_y, const float* x, int B, int H, int W, int c_in, int c_out, float* d_w, float* d_b, const unsigned* dymax,
                         const unsigned* xmax, const unsigned* xmax2, float* partial, bool one_product, hipStream_t s) {
    return conv3x3_wgrad_launch(d_y, x, B, H, W, c_in, c_out, d_w, d_b, dymax, xmax, xmax2, PAPR_SLOT_W, partial, nullptr, one_product, s);
}

extern "C" int papr_conv3x3_wgrad(const float* d_out, const float* x, int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out,
                                  float* d_w, float* d_bias, const uint32_t* d_out_max_bits, const uint32_t* x_max_bits, void* workspace,
                                  int32_t slot, papr_stream_t stream) {
    PAPR_REQUIRE(d_out && x && d_w && workspace, "papr_conv3x3_wgrad: null pointer");
    PAPR_REQUIRE(B >= 1 && H >= 1 && W >= 1 && c_in >= 4 && c_in % 4 == 0 && c_out >= 4 && c_out % 4 == 0,
                 "papr_conv3x3_wgrad: B %d, H %d, W %d, c_in %d, c_out %d (channels must be multiples of 4)", B, H, W, c_in, c_out);
    PAPR_REQUIRE(slot >= 0 && slot < 32, "papr_conv3x3_wgrad: slot %d outside 0 .. 31", slot);
    hipStream_t s = as_stream(stream);
    const long M = (long)B * H * W;
    unsigned* head = static_cast<unsigned*>(workspace);       // 64 slots: two per call (d_out, x), cleared 16 calls later
    unsigned* gmax = head + 2 * slot;
    unsigned* xmax = gmax + 1;
    unsigned* stale = head + ((2 * slot + 32) & 63);
    auto absmax = [&](const float* t, long n4, unsigned* dst, unsigned* st) {
        const long want = (n4 + 256 * 16 - 1) / (256 * 16);
        tensor_absmax_kernel<<<dim3((unsigned)(want < 256 ? want : 256)), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(t), n4, dst, st);
    };
    // (a maximum the caller already has -- the slot a papr_conv3x3_fwd call on the same tensor left behind -- spares its launch)
    if (!d_out_max_bits && !x_max_bits) {
        const long n0 = M * c_out / 4, n1 = M * c_in / 4;
        const long w0 = (n0 + 256 * 16 - 1) / (256 * 16), w1 = (n1 + 256 * 16 - 1) / (256 * 16);
        const int nb0 = (int)(w0 < 256 ? w0 : 256), nb1 = (int)(w1 < 256 ? w1 : 256);
        tensor_absmax2_kernel<<<dim3((unsigned)(nb0 + nb1)), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(d_out), n0, gmax, stale, nb0,
                                                                                reinterpret_cast<const float4*>(x), n1, xmax, stale + 1);
    } else {
        if (!d_out_max_bits) absmax(d_out, M * c_out / 4, gmax, stale);
        if (!x_max_bits) absmax(x, M * c_in / 4, xmax, d_out_max_bits ? stale : stale + 1);
    }
    PAPR_CHECK_LAUNCH("tensor_absmax");
    return conv3x3_wgrad_launch(d_out, x, B, H, W, c_in, c_out, d_w, d_bias, d_out_max_bits ? d_out_max_bits : gmax, x_max_bits ? x_max_bits : xmax, nullptr, 1,
                                reinterpret_cast<float*>(static_cast<char*>(workspace) + 256), stale, false, s);
}
