// K5b: the rest of the U-Net render head -- everything of the reference's SmallUNet (models/unet.py:182-258) that is not
// a 3x3 convolution (conv.hip):
//   * MaxPool2d(2) of the Down stages (models/unet.py:36-49)            papr_maxpool2_fwd / papr_maxpool2_bwd
//   * ConvTranspose2d(c, c/2, kernel 2, stride 2) of the Up stages (:62) papr_upconv2x2_fwd / _dgrad / _wgrad
//   * the 1x1 output convolution (OutConv, :86-93)                       papr_conv1x1_fwd / _bwd
// all over NHWC maps (rows = pixels), like conv.hip.
//
// The transposed convolution with kernel 2 and stride 2 has no overlapping taps: output pixel (2y + dy, 2x + dx) is a
// 1x1 convolution of input pixel (y, x) with the weight slice of tap (dy, dx).  With Wm = the weight as a (C_in, 4 C_out)
// matrix [c][tap][n] (the channels-last layout of the reference's (C_in, C_out, 2, 2) parameter) the three products are
//   forward        out[pix(m, tap)][n] = b[n] + sum_c  x[m][c] Wm[c][tap n]          M x 4 C_out x C_in,   stores scattered
//   data-gradient  d_x[m][c]           = sum_{tap n}   d_out[pix(m, tap)][n] Wm[c][tap n]   M x C_in x 4 C_out,   rows gathered
//   weight-grad    d_Wm[c][tap n]      = sum_m         x[m][c] d_out[pix(m, tap)][n]        C_in x 4 C_out x M,   both transposed
// each ONE launch of the same 64 x 64-tile kernel on the f16 matrix pipe with split fp32 operands (a b ~ hi hi + hi lo +
// lo hi, fp32 accumulation, as everywhere in this library).  The training maps are small (a 160 x 160 patch: the two
// layers see 40 x 40 and 80 x 80 maps, M = 1,600 and 6,400 rows), so the kernel is built for latency, not throughput: no weight pre-split, no absmax
// launch -- a workgroup finds the power-of-two scale of its own operand block in a first pass over it (the second pass
// hits L2), 64-deep k-slabs with the next slab in registers.  The weight-gradient reduces over pixels: they are dealt to
// ~512 workgroups in chunks whose partial tiles a second launch adds in a fixed order.  Operands whose contiguous dimension is not k (the weight in the forward
// product, both operands of the weight-gradient) are transposed on their way into LDS (4-wide runs, rows permuted so
// that the lanes of a write hit different banks); the accumulators leave through LDS, so stores are rows of 256 bytes.
#include "papr_common.h"
#include "h3_common.h"
#include "unet_parts.h"

namespace {

__device__ __forceinline__ float absmax4f(const float4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// ---------------------------------------------------------------- MaxPool2d(2)
// out[b][y][x][c] = max over the 2 x 2 window; which = position (0..3, row-major) of the FIRST maximum in scan order (a
// later element replaces the current one only if it is greater, or NaN: torch's max_pool2d rule, so ties -- frequent
// after a ReLU -- send the gradient where torch sends it).  H and W odd: the last row / column is dropped (floor).
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float4* __restrict__ x, int LD4, int H, int W, int C4, long total, float4* __restrict__ out,
                                                           unsigned* __restrict__ which) {       // (LD4: float4 per input pixel row, >= C4: a channel slice of a wider map)
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int Ho = H >> 1, Wo = W >> 1;
    const int c = (int)(e % C4);
    long r = e / C4;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const long b = r / Ho;
    const float4* src = x + ((b * H + 2 * yo) * W + 2 * xo) * LD4 + c;
    const float4 v[4] = {src[0], src[LD4], src[(long)W * LD4], src[(long)W * LD4 + LD4]};
    float4 m = v[0];
    unsigned w = 0;
#pragma unroll
    for (int t = 1; t < 4; ++t) {
        if (v[t].x > m.x || v[t].x != v[t].x) { m.x = v[t].x; w = (w & ~0xffu) | (unsigned)t; }
        if (v[t].y > m.y || v[t].y != v[t].y) { m.y = v[t].y; w = (w & ~0xff00u) | ((unsigned)t << 8); }
        if (v[t].z > m.z || v[t].z != v[t].z) { m.z = v[t].z; w = (w & ~0xff0000u) | ((unsigned)t << 16); }
        if (v[t].w > m.w || v[t].w != v[t].w) { m.w = v[t].w; w = (w & ~0xff000000u) | ((unsigned)t << 24); }
    }
    out[e] = m;
    if (which) which[e] = w;
}

// d_in (B, H, W, C): every element written (zeros off the maxima and in a dropped last row / column)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float4* __restrict__ d_out, const unsigned* __restrict__ which, int H, int W, int C4,
                                                           long total_in, float4* __restrict__ d_in) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total_in) return;
    const int Ho = H >> 1, Wo = W >> 1;
    const int c = (int)(e % C4);
    long r = e / C4;
    const int xi = (int)(r % W); r /= W;
    const int yi = (int)(r % H);
    const long b = r / H;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    const int yo = yi >> 1, xo = xi >> 1;
    if (yo < Ho && xo < Wo) {
        const long o = ((b * Ho + yo) * Wo + xo) * C4 + c;
        const unsigned w = which[o], me = (unsigned)((yi & 1) * 2 + (xi & 1));
        const float4 d = d_out[o];
        if ((w & 0xff) == me) g.x = d.x;
        if (((w >> 8) & 0xff) == me) g.y = d.y;
        if (((w >> 16) & 0xff) == me) g.z = d.z;
        if ((w >> 24) == me) g.w = d.w;
    }
    d_in[e] = g;
}

// The same at a seam of the whole-network backward pass (small_unet.hip): the pooled map's source was a ReLU output y that also feeds the skip
// concatenation, so  d_in = (pool-backward(d_out) + skip gradient) * (y > 0)  -- pooling, the sum of the two uses and the ReLU mask in one pass
// (three launches and a slice copy otherwise) -- and max |d_in| for the convolutions that consume it.
__global__ __launch_bounds__(256) void maxpool2_bwd_fused_kernel(const float4* __restrict__ d_out, const unsigned* __restrict__ which, int H, int W, int C4,
                                                                 long total_in, const float4* __restrict__ skip, int LDS4, const float4* __restrict__ y, int LDY4,
                                                                 float4* __restrict__ d_in, unsigned* __restrict__ out_max) {
    float mx = 0.f;
    const int Ho = H >> 1, Wo = W >> 1;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total_in; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C4);
        const long pix = e / C4;
        long r = pix;
        const int xi = (int)(r % W); r /= W;
        const int yi = (int)(r % H);
        const long b = r / H;
        float4 g = skip[pix * LDS4 + c];
        const float4 yv = y[pix * LDY4 + c];
        const int yo = yi >> 1, xo = xi >> 1;
        if (yo < Ho && xo < Wo) {
            const long o = ((b * Ho + yo) * Wo + xo) * C4 + c;
            const unsigned w = which[o], me = (unsigned)((yi & 1) * 2 + (xi & 1));
            const float4 d = d_out[o];
            if ((w & 0xff) == me) g.x += d.x;
            if (((w >> 8) & 0xff) == me) g.y += d.y;
            if (((w >> 16) & 0xff) == me) g.z += d.z;
            if ((w >> 24) == me) g.w += d.w;
        }
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        d_in[e] = g;
        mx = fmaxf(mx, absmax4f(g));
    }
    papr_wg_max_to_slot(out_max, mx);
}

// ---------------------------------------------------------------- ConvTranspose2d(kernel 2, stride 2)
constexpr int UP_T = 64;                       // tile edge (i and j)
constexpr int UP_BK = 64;                      // k-slab
constexpr int UP_HP = UP_BK + 8;               // LDS row pitch in halfs (144 bytes)
constexpr int UP_PLANE = UP_T * UP_HP;
constexpr int UP_CP = UP_T + 4;                // pitch of the fp32 output tile in LDS
constexpr int UP_QN = UP_BK / 16;              // float4 per thread and slab, k-contiguous operand (all 256 threads)
constexpr int UP_TN = UP_BK / 8;               // float4 per thread and slab, transposed operand (128 threads)

struct UpArgs {
    const float* x;        // forward / weight-gradient: the layer's input (M, C_in)
    const float* g;        // data- / weight-gradient: d_out (B, 2H, 2W, C_out)
    const float* wm;       // (C_in, 4 C_out) [c][tap][n]
    const float* bias;     // forward: (C_out) or null
    float* out;            // forward: (B, 2H, 2W, C_out); data-gradient: (M, C_in); weight-gradient: (C_in, 4 C_out)
    const float* maxes;    // weight-gradient: 64 partial maxima of |x|, then 64 of |d_out| (up_stats_kernel)
    int B, H, W, C_in, C_out;
    int px_per_chunk;      // weight-gradient: pixels per workgroup along blockIdx.z (multiple of UP_BK); the chunks' tiles go to
                           // out + z * C_in * 4 C_out and meet in upconv_wgrad_reduce_kernel
    int ldo;               // forward: row stride of out (>= C_out: a channel slice of the skip concatenation)
    int ldg;               // data- / weight-gradient: row stride of g (>= C_out: a slice of the concatenation's gradient)
    const float* mask;     // data-gradient, or null: the layer's input was a ReLU output y (M, C_in): d_x *= (y > 0)
    unsigned* out_max;     // forward / data-gradient, or null: atomicMax of max |out|
    const unsigned* pmax_bits; const unsigned* qmax_bits;      // weight-gradient, instead of `maxes`: the slots (papr_common.h) of max |x|, max |g|
    float* partial_b;      // weight-gradient, or null: [chunk][4 C_out] column sums of g (the bias gradient), from the workgroups of the first c_in tile
};

__device__ __forceinline__ float comp4u(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
__device__ __forceinline__ float absmax4(float m, const float4& v) { return fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); }
__device__ __forceinline__ int up_logical(int R) { return 4 * (R & 15) + (R >> 4); }      // LDS row of a transposed operand -> index in the tile

// MODE 0 forward, 1 data-gradient, 2 weight-gradient
// ONE: a single f16 product per fp32 product (hi planes only): the reference's fp16-autocast arithmetic (`use_amp: true`)
template <int MODE, bool ONE>
__global__ __launch_bounds__(256) void upconv2x2_h3_kernel(UpArgs p) {
    __shared__ __attribute__((aligned(16))) _Float16 planes[4 * UP_PLANE];
    __shared__ float red[20];
    _Float16* Ph = planes;
    _Float16* Pl = Ph + UP_PLANE;
    _Float16* Qh = Pl + UP_PLANE;
    _Float16* Ql = Qh + UP_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm_ = wave >> 1, wn_ = wave & 1;
    const int HW = p.H * p.W, W2 = 2 * p.W;
    const int M = p.B * HW, J4 = 4 * p.C_out;
    const int i0 = blockIdx.x * UP_T, j0 = blockIdx.y * UP_T;
    const int pbeg = MODE == 2 ? (int)blockIdx.z * p.px_per_chunk : 0;                    // weight-gradient: this workgroup's pixels
    const int pend = MODE == 2 ? (pbeg + p.px_per_chunk < M ? pbeg + p.px_per_chunk : M) : 0;
    const int K = MODE == 0 ? p.C_in : (MODE == 1 ? J4 : (pend - pbeg + UP_BK - 1) / UP_BK * UP_BK);
    // first output pixel (tap 0) of input pixel m
    auto opix = [&](int m) { const int b = m / HW, r = m - b * HW, y = r / p.W, x = r - y * p.W; return (b * 2 * p.H + 2 * y) * W2 + 2 * x; };
    const int tapJ = j0 / p.C_out, nJ = j0 - tapJ * p.C_out;                 // weight-gradient / forward: the tile's tap and first channel
    const int tapoffJ = (tapJ >> 1) * W2 + (tapJ & 1);

    // ---- per-thread geometry of the two staging patterns
    // k-contiguous: UP_QN rows (kr + 16 q), floats kc .. kc + 3 of the slab
    const int kr = tid >> 4, kc = (tid & 15) * 4;
    // transposed (threads of one half of the workgroup): tile indices 4 ig .. 4 ig + 3, slab k's UP_TN kg .. + UP_TN - 1
    const int tt = tid & 127, ig = tt & 15, kg = tt >> 4;
    const bool firsthalf = tid < 128;

    long prow[UP_QN];              // MODE 0: element offset of the thread's x rows; MODE 1: first output pixel of its rows
    bool prow_ok[UP_QN];
    if (MODE != 2) {
#pragma unroll
        for (int q = 0; q < UP_QN; ++q) {
            int m = i0 + kr + 16 * q;
            prow_ok[q] = m < M;
            m = prow_ok[q] ? m : M - 1;
            prow[q] = MODE == 0 ? (long)m * p.C_in : (long)opix(m);
        }
    }
    float4 rp[MODE == 2 ? UP_TN : UP_QN], rq[MODE == 0 ? UP_TN : UP_QN];
    bool okt[UP_TN];               // weight-gradient: the thread's pixels of the slab exist

    auto load_p = [&](int k0) {
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < UP_QN; ++q) rp[q] = *reinterpret_cast<const float4*>(p.x + prow[q] + k0 + kc);
        } else if (MODE == 1) {
            const int tap = k0 / p.C_out, n = k0 - tap * p.C_out;
            const int tapoff = (tap >> 1) * W2 + (tap & 1);
#pragma unroll
            for (int q = 0; q < UP_QN; ++q) rp[q] = *reinterpret_cast<const float4*>(p.g + (prow[q] + tapoff) * p.ldg + n + kc);
        } else {
#pragma unroll
            for (int u = 0; u < UP_TN; ++u) {
                int m = pbeg + k0 + UP_TN * kg + u;
                okt[u] = m < pend;
                m = okt[u] ? m : pend - 1;
                rp[u] = firsthalf ? *reinterpret_cast<const float4*>(p.x + (long)m * p.C_in + i0 + 4 * ig)
                                  : *reinterpret_cast<const float4*>(p.g + ((long)opix(m) + tapoffJ) * p.ldg + nJ + 4 * ig);
            }
        }
    };
    auto load_q = [&](int k0) {
        if (MODE == 0) {
            if (firsthalf) {
#pragma unroll
                for (int u = 0; u < UP_TN; ++u) rq[u] = *reinterpret_cast<const float4*>(p.wm + (long)(k0 + UP_TN * kg + u) * J4 + j0 + 4 * ig);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < UP_QN; ++q) rq[q] = *reinterpret_cast<const float4*>(p.wm + (long)(j0 + kr + 16 * q) * J4 + k0 + kc);
        }
    };

    // ---- first pass: max |.| of the workgroup's activation block(s) -> power-of-two scales
    // (weight-gradient: one scale per tensor, from the maxima a small launch left -- its 512 workgroups would read every
    //  block sixteen times over for their own)
    float mx = 0.f;
    if (MODE == 2) {
        if (p.pmax_bits) mx = __uint_as_float(papr_slot_max(wave == 0 ? p.pmax_bits : p.qmax_bits, PAPR_SLOT_W));
        else if (tid < 128) mx = p.maxes[tid];
    } else {
        for (int k0 = 0; k0 < K; k0 += UP_BK) {
            load_p(k0);
#pragma unroll
            for (int q = 0; q < UP_QN; ++q) if (prow_ok[q]) mx = absmax4(mx, rp[q]);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    auto scale_of = [](float m, float& inv) {
        const unsigned mb = __float_as_uint(m);
        const int ea = mb ? (int)((mb >> 23) & 0xff) : 127 + 13;
        inv = pow2_from_biased(127 - 13 + (ea - 127));
        return pow2_from_biased(127 + 13 - (ea - 127));
    };
    float p_inv, q_inv = 1.f, q_scale = 1.f;
    const float p_scale = MODE == 2 ? scale_of(red[0], p_inv) : scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), p_inv);
    if (MODE == 2) q_scale = scale_of(red[1], q_inv);

    // ---- staging into LDS
    auto put_rows = [&](const float4* r, const bool* ok, float sc, _Float16* hi_plane, _Float16* lo_plane) {       // k-contiguous
#pragma unroll
        for (int q = 0; q < UP_QN; ++q) {
            half4 hi, lo;
            split4(r[q], (ok == nullptr || ok[q]) ? sc : 0.f, hi, lo);
            const int off = (kr + 16 * q) * UP_HP + kc;
            *reinterpret_cast<half4*>(hi_plane + off) = hi;
            if (!ONE) *reinterpret_cast<half4*>(lo_plane + off) = lo;
        }
    };
    auto put_cols = [&](const float4* r, const bool* ok, float sc, _Float16* hi_plane, _Float16* lo_plane) {       // transposed
#pragma unroll
        for (int j = 0; j < 4; ++j) {                    // tile index 4 ig + j -> LDS row ig + 16 j: its UP_TN k's
#pragma unroll
            for (int h = 0; h < UP_TN / 4; ++h) {
                float4 col;
                col.x = comp4u(r[4 * h + 0], j) * ((ok == nullptr || ok[4 * h + 0]) ? sc : 0.f);
                col.y = comp4u(r[4 * h + 1], j) * ((ok == nullptr || ok[4 * h + 1]) ? sc : 0.f);
                col.z = comp4u(r[4 * h + 2], j) * ((ok == nullptr || ok[4 * h + 2]) ? sc : 0.f);
                col.w = comp4u(r[4 * h + 3], j) * ((ok == nullptr || ok[4 * h + 3]) ? sc : 0.f);
                half4 hi, lo;
                split4(col, 1.0f, hi, lo);
                const int off = (ig + 16 * j) * UP_HP + UP_TN * kg + 4 * h;
                *reinterpret_cast<half4*>(hi_plane + off) = hi;
                if (!ONE) *reinterpret_cast<half4*>(lo_plane + off) = lo;
            }
        }
    };
    const bool do_bias = MODE == 2 && p.partial_b != nullptr && blockIdx.x == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto store_slab = [&]() {
        if (MODE == 0) {
            put_rows(rp, prow_ok, p_scale, Ph, Pl);
            if (firsthalf) put_cols(rq, nullptr, 1.0f, Qh, Ql);
        } else if (MODE == 1) {
            put_rows(rp, prow_ok, p_scale, Ph, Pl);
            put_rows(rq, nullptr, 1.0f, Qh, Ql);
        } else {
            if (firsthalf) put_cols(rp, okt, p_scale, Ph, Pl);
            else {
                if (do_bias) {
#pragma unroll
                    for (int u = 0; u < UP_TN; ++u)
                        if (okt[u]) { bsum.x += rp[u].x; bsum.y += rp[u].y; bsum.z += rp[u].z; bsum.w += rp[u].w; }
                }
                put_cols(rp, okt, q_scale, Qh, Ql);
            }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int frag = (lane & 31) * UP_HP + 8 * (lane >> 5);
    load_p(0); load_q(0);
    store_slab();
    lds_barrier();
    for (int k0 = 0; k0 < K; k0 += UP_BK) {
        const bool more = k0 + UP_BK < K;
        if (more) { load_p(k0 + UP_BK); load_q(k0 + UP_BK); }
#pragma unroll
        for (int ks = 0; ks < UP_BK; ks += 16) {
            const int op = (wm_ * 32) * UP_HP + frag + ks, oq = (wn_ * 32) * UP_HP + frag + ks;
            const half8 ph = *reinterpret_cast<const half8*>(Ph + op), qh = *reinterpret_cast<const half8*>(Qh + oq);
            if (!ONE) {
                const half8 pl = *reinterpret_cast<const half8*>(Pl + op), ql = *reinterpret_cast<const half8*>(Ql + oq);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, qh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, ql, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, qh, acc, 0, 0, 0);
        }
        lds_barrier();                               // everybody has read the slab
        if (more) store_slab();
        lds_barrier();
    }

    // ---- the tile leaves through LDS: acc[e] = C[i-row 32 wm + (e & 3) + 8 (e >> 2) + 4 (lane >> 5)][j-row 32 wn + lane % 32]
    float* Ct = reinterpret_cast<float*>(planes);
    {
        const int jr = wn_ * 32 + (lane & 31);
        const int jl = (MODE == 0 || MODE == 2) ? up_logical(jr) : jr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ir = wm_ * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const int il = MODE == 2 ? up_logical(ir) : ir;
            Ct[il * UP_CP + jl] = acc[e];
        }
    }
    __syncthreads();
    const float inv = p_inv * q_inv;
    const int c4 = (tid & 15) * 4;
    float omax = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (tid >> 4) + 16 * q;
        float4 v = *reinterpret_cast<const float4*>(Ct + row * UP_CP + c4);
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        if (MODE == 0) {
            const int m = i0 + row;
            if (m >= M) continue;
            if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + nJ + c4); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
            omax = fmaxf(omax, absmax4f(v));
            *reinterpret_cast<float4*>(p.out + ((long)opix(m) + tapoffJ) * p.ldo + nJ + c4) = v;
        } else if (MODE == 1) {
            const int m = i0 + row;
            if (m >= M) continue;
            if (p.mask) {
                const float4 yv = *reinterpret_cast<const float4*>(p.mask + (long)m * p.C_in + j0 + c4);
                v.x = yv.x > 0.f ? v.x : 0.f; v.y = yv.y > 0.f ? v.y : 0.f; v.z = yv.z > 0.f ? v.z : 0.f; v.w = yv.w > 0.f ? v.w : 0.f;
            }
            omax = fmaxf(omax, absmax4f(v));
            *reinterpret_cast<float4*>(p.out + (long)m * p.C_in + j0 + c4) = v;
        } else {
            *reinterpret_cast<float4*>(p.out + ((long)blockIdx.z * p.C_in + i0 + row) * J4 + j0 + c4) = v;
        }
    }
    if (MODE != 2 && p.out_max) papr_wg_max_to_slot(p.out_max, omax);
    if (MODE == 2 && do_bias) {                      // the bias gradient's share of this chunk: the eight pixel lanes of a column group meet in order
        __syncthreads();
        float4* bs = reinterpret_cast<float4*>(planes);
        if (!firsthalf) bs[kg * 16 + ig] = bsum;
        __syncthreads();
        if (tid < 16) {
            float4 t = bs[tid];
#pragma unroll
            for (int g = 1; g < UP_TN; ++g) { const float4 v = bs[g * 16 + tid]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            *reinterpret_cast<float4*>(p.partial_b + (long)blockIdx.z * J4 + j0 + 4 * tid) = t;
        }
    }
}

// Statistics of the weight-gradient's two operands, one pass over each: out[x] = max |a| over the slice workgroup x strides
// through (blockIdx.y == 0); out[64 + x] = the same for b, and colsum[x][C] = the column sums of b's rows x, x + 64, ...
// (blockIdx.y == 1; b = d_out (rows, C): the bias gradient's partial sums).  Thread = float4 column t % C4, row lane t / C4.
__global__ __launch_bounds__(256) void up_stats_kernel(const float4* __restrict__ a, long na4, const float4* __restrict__ b, long rows, int C4,
                                                       float* __restrict__ out, float4* __restrict__ colsum) {
    __shared__ float part[4];
    __shared__ float4 cs[256];
    float m = 0.f;
    if (blockIdx.y == 0) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na4; i += 64L * 256) m = absmax4(m, a[i]);
    } else {
        const int RL = 256 / C4, c4 = threadIdx.x % C4, rl = threadIdx.x / C4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < RL) {
#pragma unroll 4
            for (long r = (long)blockIdx.x + 64L * rl; r < rows; r += 64L * RL) {
                const float4 v = b[r * C4 + c4];
                m = absmax4(m, v);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        cs[threadIdx.x] = s;
        __syncthreads();
        if (colsum && rl == 0) {
            float4 t = cs[c4];
            for (int r = 1; r < RL; ++r) { const float4 v = cs[r * C4 + c4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            colsum[(long)blockIdx.x * C4 + c4] = t;
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[64 * blockIdx.y + blockIdx.x] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

// d_wm = sum of the chunks' tiles, in chunk order; d_bias = sum of the 64 column sums of up_stats_kernel
__global__ __launch_bounds__(256) void upconv_wgrad_reduce_kernel(const float4* __restrict__ partial, int chunks, long n4, float4* __restrict__ out,
                                                                  const float4* __restrict__ partial_b, int nb4, float4* __restrict__ out_b) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (out_b && e < nb4) {
        float4 r = partial_b[e];
#pragma unroll 8
        for (int z = 1; z < 64; ++z) { const float4 v = partial_b[(long)z * nb4 + e]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        out_b[e] = r;
    }
    if (e >= n4) return;
    float4 r = partial[e];
#pragma unroll 8
    for (int z = 1; z < chunks; ++z) { const float4 v = partial[(long)z * n4 + e]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
    out[e] = r;
}

// the same with the bias gradient from the weight-gradient launch's own column sums: d_bias[n] = sum over chunks and the four taps of partial_b[chunk][tap][n]
__global__ __launch_bounds__(256) void upconv_wgrad_reduce2_kernel(const float4* __restrict__ partial, int chunks, long n4, float4* __restrict__ out,
                                                                   const float4* __restrict__ partial_b, int nb4, float4* __restrict__ out_b) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (out_b && e < nb4) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = 0; z < chunks; ++z)
#pragma unroll
            for (int t = 0; t < 4; ++t) { const float4 v = partial_b[((long)z * 4 + t) * nb4 + e]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        out_b[e] = r;
    }
    if (e >= n4) return;
    float4 r = partial[e];
#pragma unroll 8
    for (int z = 1; z < chunks; ++z) { const float4 v = partial[(long)z * n4 + e]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
    out[e] = r;
}

// pixels per workgroup of the weight-gradient launch: ~512 workgroups, at least two slabs each
static int up_px_per_chunk(long M, int c_in, int c_out) {
    const long tiles = (long)(c_in / UP_T) * (4 * c_out / UP_T);
    long chunks = (512 + tiles - 1) / tiles;
    long px = (M + chunks - 1) / chunks;
    px = (px + UP_BK - 1) / UP_BK * UP_BK;
    return (int)(px < 2 * UP_BK ? 2 * UP_BK : px);
}

// ---------------------------------------------------------------- 1x1 output convolution (few output channels: HBM-bound, plain fp32)
constexpr int C1_MAXN = 4;
// out[m][n] = b[n] + sum_c x[m][c] w[n][c]: 32 lanes per pixel (a float4 of channels each, C <= 128 per pass), butterfly sum
__global__ __launch_bounds__(256) void conv1x1_fwd_kernel(const float* __restrict__ x, long M, int C, const float* __restrict__ w, const float* __restrict__ bias,
                                                          int N, float* __restrict__ out) {
    const int l32 = threadIdx.x & 31;
    const long m = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const long mc = m < M ? m : M - 1;
    float s[C1_MAXN] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 4 * l32; c < C; c += 128) {
        const float4 v = *reinterpret_cast<const float4*>(x + mc * C + c);
#pragma unroll
        for (int n = 0; n < C1_MAXN; ++n)
            if (n < N) {
                const float4 ww = *reinterpret_cast<const float4*>(w + (long)n * C + c);
                s[n] = __builtin_fmaf(v.w, ww.w, __builtin_fmaf(v.z, ww.z, __builtin_fmaf(v.y, ww.y, __builtin_fmaf(v.x, ww.x, s[n]))));
            }
    }
#pragma unroll
    for (int n = 0; n < C1_MAXN; ++n)
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) s[n] += __shfl_xor(s[n], o, 32);
    if (m < M && l32 < N) {
        const float r = l32 == 0 ? s[0] : (l32 == 1 ? s[1] : (l32 == 2 ? s[2] : s[3]));
        out[m * N + l32] = r + (bias ? bias[l32] : 0.f);
    }
}

// d_x[m][c] = sum_n d_y[m][n] w[n][c]   (one thread per float4 of d_x)
// (mask, or null: x was a ReLU output y (M, C): d_x *= (y > 0); out_max, or null: atomicMax of max |d_x|)
__global__ __launch_bounds__(256) void conv1x1_dgrad_kernel(const float* __restrict__ dy, long M, int C, const float* __restrict__ w, int N, float* __restrict__ dx,
                                                            const float* __restrict__ mask, unsigned* __restrict__ out_max) {
    const int C4 = C >> 2;
    float mx = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < M * C4; e += (long)gridDim.x * 256) {
        const long m = e / C4;
        const int c = (int)(e - m * C4) * 4;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = 0; n < N; ++n) {
            const float g = dy[m * N + n];
            const float4 ww = *reinterpret_cast<const float4*>(w + (long)n * C + c);
            r.x = __builtin_fmaf(g, ww.x, r.x); r.y = __builtin_fmaf(g, ww.y, r.y); r.z = __builtin_fmaf(g, ww.z, r.z); r.w = __builtin_fmaf(g, ww.w, r.w);
        }
        if (mask) {
            const float4 yv = *reinterpret_cast<const float4*>(mask + m * C + c);
            r.x = yv.x > 0.f ? r.x : 0.f; r.y = yv.y > 0.f ? r.y : 0.f; r.z = yv.z > 0.f ? r.z : 0.f; r.w = yv.w > 0.f ? r.w : 0.f;
        }
        mx = fmaxf(mx, absmax4f(r));
        *reinterpret_cast<float4*>(dx + m * C + c) = r;
    }
    if (out_max) papr_wg_max_to_slot(out_max, mx);
}

// partial[chunk][n][c] = sum over the chunk's pixels of d_y[m][n] x[m][c]; partial_b[chunk][n] = sum d_y[m][n].
// Workgroup = one chunk of pixels; thread = one float4 of channels (c4 = tid % C4) and one pixel lane (tid / C4).
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, long M, int C, int N, long px_per_chunk,
                                                            float* __restrict__ partial, float* __restrict__ partial_b) {
    __shared__ float4 sm[256];
    const int C4 = C >> 2, lanes = 256 / C4;                 // C4 in {8, 16, 32, 64}
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
    const long beg = (long)blockIdx.x * px_per_chunk;
    long end = beg + px_per_chunk;
    if (end > M) end = M;
    float4 s[C1_MAXN];
    float sb[C1_MAXN];
#pragma unroll
    for (int n = 0; n < C1_MAXN; ++n) { s[n] = make_float4(0.f, 0.f, 0.f, 0.f); sb[n] = 0.f; }
#pragma unroll 4
    for (long m = beg + pl; m < end; m += lanes) {
        const float4 v = *reinterpret_cast<const float4*>(x + m * C + 4 * c4);
#pragma unroll
        for (int n = 0; n < C1_MAXN; ++n)
            if (n < N) {
                const float g = dy[m * N + n];
                s[n].x = __builtin_fmaf(g, v.x, s[n].x); s[n].y = __builtin_fmaf(g, v.y, s[n].y);
                s[n].z = __builtin_fmaf(g, v.z, s[n].z); s[n].w = __builtin_fmaf(g, v.w, s[n].w);
                sb[n] += g;
            }
    }
    for (int n = 0; n < N; ++n) {                            // the pixel lanes meet in a fixed order
        __syncthreads();
        sm[threadIdx.x] = n == 0 ? s[0] : (n == 1 ? s[1] : (n == 2 ? s[2] : s[3]));
        __syncthreads();
        if (pl == 0) {
            float4 t = sm[c4];
            for (int r = 1; r < lanes; ++r) { const float4 v = sm[r * C4 + c4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            *reinterpret_cast<float4*>(partial + ((long)blockIdx.x * N + n) * C + 4 * c4) = t;
        }
    }
    if (partial_b) {
        __syncthreads();
        float* sf = reinterpret_cast<float*>(sm);
        if (c4 == 0) {
#pragma unroll
            for (int n = 0; n < C1_MAXN; ++n) sf[pl * C1_MAXN + n] = sb[n];
        }
        __syncthreads();
        if ((int)threadIdx.x < N) {
            float t = 0.f;
            for (int r = 0; r < lanes; ++r) t += sf[r * C1_MAXN + threadIdx.x];
            partial_b[(long)blockIdx.x * N + threadIdx.x] = t;
        }
    }
}

// One wave per output (d_w element, then d_bias element): lane l adds chunks l, l + 64, ..., the lanes meet in a butterfly
// (a fixed order, so the sum is reproducible).
__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ partial_b, int chunks, int NC, int N,
                                                                   float* __restrict__ dw, float* __restrict__ db) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), zl = threadIdx.x & 63;
    const int total = NC + (db ? N : 0);
    if (o >= total) return;
    const bool isb = o >= NC;
    const float* src = isb ? partial_b + (o - NC) : partial + o;
    const long stride = isb ? N : NC;
    float t = 0.f;
    for (int z = zl; z < chunks; z += 64) t += src[(long)z * stride];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
    if (zl == 0) { if (isb) db[o - NC] = t; else dw[o] = t; }
}

static long c1_px_per_chunk(long M) {
    long chunks = (M + 47) / 48;
    if (chunks > 1024) chunks = 1024;
    if (chunks < 1) chunks = 1;
    return (M + chunks - 1) / chunks;
}

}  // namespace

int papr_i_maxpool2_fwd(const float* x, int ld_in, int B, int H, int W, int C, float* out, unsigned* which, hipStream_t s) {
    const long total = (long)B * (H / 2) * (W / 2) * (C / 4);
    maxpool2_fwd_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(x), ld_in / 4, H, W, C / 4, total,
                                                                                  reinterpret_cast<float4*>(out), which);
    PAPR_CHECK_LAUNCH("maxpool2_fwd");
    return 0;
}

extern "C" int papr_maxpool2_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float* out, uint32_t* which, papr_stream_t stream) {
    PAPR_REQUIRE(x && out, "papr_maxpool2_fwd: null pointer");
    PAPR_REQUIRE(B >= 1 && H >= 2 && W >= 2 && C >= 4 && C % 4 == 0, "papr_maxpool2_fwd: B %d, H %d, W %d, C %d (multiple of 4)", B, H, W, C);
    return papr_i_maxpool2_fwd(x, C, B, H, W, C, out, which, as_stream(stream));
}

int papr_i_maxpool2_bwd_fused(const float* d_out, const unsigned* which, int B, int H, int W, int C, const float* skip, int ld_skip, const float* y, int ld_y,
                              float* d_in, unsigned* out_max, hipStream_t s) {
    const long total = (long)B * H * W * (C / 4);
    const long nb = (total + 255) / 256;
    maxpool2_bwd_fused_kernel<<<dim3((unsigned)(nb > 1024 ? 1024 : nb)), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(d_out), which, H, W, C / 4, total,
                                                                                        reinterpret_cast<const float4*>(skip), ld_skip / 4,
                                                                                        reinterpret_cast<const float4*>(y), ld_y / 4, reinterpret_cast<float4*>(d_in), out_max);
    PAPR_CHECK_LAUNCH("maxpool2_bwd_fused");
    return 0;
}

extern "C" int papr_maxpool2_bwd(const float* d_out, const uint32_t* which, int32_t B, int32_t H, int32_t W, int32_t C, float* d_in, papr_stream_t stream) {
    PAPR_REQUIRE(d_out && which && d_in, "papr_maxpool2_bwd: null pointer");
    PAPR_REQUIRE(B >= 1 && H >= 2 && W >= 2 && C >= 4 && C % 4 == 0, "papr_maxpool2_bwd: B %d, H %d, W %d, C %d (multiple of 4)", B, H, W, C);
    const long total = (long)B * H * W * (C / 4);
    maxpool2_bwd_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(reinterpret_cast<const float4*>(d_out), which, H, W, C / 4, total,
                                                                                                  reinterpret_cast<float4*>(d_in));
    PAPR_CHECK_LAUNCH("maxpool2_bwd");
    return 0;
}

static int up_check(const char* who, int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out) {
    PAPR_REQUIRE(B >= 1 && H >= 1 && W >= 1 && c_in >= 64 && c_in % 64 == 0 && c_out >= 64 && c_out % 64 == 0,
                 "%s: B %d, H %d, W %d, c_in %d, c_out %d (channels must be multiples of 64)", who, B, H, W, c_in, c_out);
    PAPR_REQUIRE((long)B * H * W * 4 * (c_in > c_out ? c_in : c_out) < (1L << 31), "%s: map too large for 32-bit pixel arithmetic", who);
    return 0;
}

int papr_i_upconv_fwd(const float* x, int B, int H, int W, int c_in, const float* wm, const float* bias, int c_out, float* out, int ldo, unsigned* out_max,
                      bool one_product, hipStream_t s) {
    UpArgs a{};
    a.x = x; a.wm = wm; a.bias = bias; a.out = out; a.B = B; a.H = H; a.W = W; a.C_in = c_in; a.C_out = c_out; a.ldo = ldo; a.ldg = c_out; a.out_max = out_max;
    const long M = (long)B * H * W;
    const dim3 grid((unsigned)((M + UP_T - 1) / UP_T), (unsigned)(4 * c_out / UP_T));
    if (one_product) upconv2x2_h3_kernel<0, true><<<grid, dim3(256), 0, s>>>(a);
    else upconv2x2_h3_kernel<0, false><<<grid, dim3(256), 0, s>>>(a);
    PAPR_CHECK_LAUNCH("upconv2x2_h3<fwd>");
    return 0;
}

extern "C" int papr_upconv2x2_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t c_in, const float* wm, const float* bias, int32_t c_out,
                                  float* out, papr_stream_t stream) {
    PAPR_REQUIRE(x && wm && out, "papr_upconv2x2_fwd: null pointer");
    if (int rc = up_check("papr_upconv2x2_fwd", B, H, W, c_in, c_out)) return rc;
    return papr_i_upconv_fwd(x, B, H, W, c_in, wm, bias, c_out, out, c_out, nullptr, false, as_stream(stream));
}

int papr_i_upconv_dgrad(const float* g, int ldg, int B, int H, int W, int c_in, const float* wm, int c_out, const float* mask_y, float* d_x, unsigned* out_max,
                        bool one_product, hipStream_t s) {
    UpArgs a{};
    a.g = g; a.wm = wm; a.out = d_x; a.B = B; a.H = H; a.W = W; a.C_in = c_in; a.C_out = c_out; a.ldo = c_out; a.ldg = ldg; a.mask = mask_y; a.out_max = out_max;
    const long M = (long)B * H * W;
    const dim3 grid((unsigned)((M + UP_T - 1) / UP_T), (unsigned)(c_in / UP_T));
    if (one_product) upconv2x2_h3_kernel<1, true><<<grid, dim3(256), 0, s>>>(a);
    else upconv2x2_h3_kernel<1, false><<<grid, dim3(256), 0, s>>>(a);
    PAPR_CHECK_LAUNCH("upconv2x2_h3<dgrad>");
    return 0;
}

extern "C" int papr_upconv2x2_dgrad(const float* d_out, int32_t B, int32_t H, int32_t W, int32_t c_in, const float* wm, int32_t c_out, float* d_x,
                                    papr_stream_t stream) {
    PAPR_REQUIRE(d_out && wm && d_x, "papr_upconv2x2_dgrad: null pointer");
    if (int rc = up_check("papr_upconv2x2_dgrad", B, H, W, c_in, c_out)) return rc;
    return papr_i_upconv_dgrad(d_out, c_out, B, H, W, c_in, wm, c_out, nullptr, d_x, nullptr, false, as_stream(stream));
}

extern "C" size_t papr_upconv2x2_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out) {
    const long M = (long)B * H * W, px = up_px_per_chunk(M, c_in, c_out), chunks = (M + px - 1) / px;
    return 512 + (size_t)64 * c_out * sizeof(float) + (size_t)chunks * c_out * 4 * c_in * sizeof(float);
}

extern "C" int papr_upconv2x2_wgrad(const float* d_out, const float* x, int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out, float* d_wm,
                                    float* d_bias, void* workspace, papr_stream_t stream) {
    PAPR_REQUIRE(d_out && x && d_wm && workspace, "papr_upconv2x2_wgrad: null pointer");
    if (int rc = up_check("papr_upconv2x2_wgrad", B, H, W, c_in, c_out)) return rc;
    const long M = (long)B * H * W, px = up_px_per_chunk(M, c_in, c_out), chunks = (M + px - 1) / px;
    hipStream_t s = as_stream(stream);
    UpArgs a{};
    a.x = x; a.g = d_out; a.B = B; a.H = H; a.W = W; a.C_in = c_in; a.C_out = c_out; a.px_per_chunk = (int)px; a.ldg = c_out; a.ldo = c_out;
    float* maxes = static_cast<float*>(workspace);
    float* colsum = maxes + 128;
    PAPR_REQUIRE(c_out <= 1024, "papr_upconv2x2_wgrad: c_out %d > 1024", c_out);
    up_stats_kernel<<<dim3(64, 2), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(x), M * c_in / 4, reinterpret_cast<const float4*>(d_out), 4 * M, c_out / 4,
                                                      maxes, d_bias ? reinterpret_cast<float4*>(colsum) : nullptr);
    PAPR_CHECK_LAUNCH("up_stats");
    a.maxes = maxes;
    a.out = colsum + (size_t)64 * c_out;
    upconv2x2_h3_kernel<2, false><<<dim3((unsigned)(c_in / UP_T), (unsigned)(4 * c_out / UP_T), (unsigned)chunks), dim3(256), 0, s>>>(a);
    PAPR_CHECK_LAUNCH("upconv2x2_h3<wgrad>");
    const long n4 = (long)c_in * c_out;
    upconv_wgrad_reduce_kernel<<<dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(a.out), (int)chunks, n4,
                                                                                         reinterpret_cast<float4*>(d_wm), reinterpret_cast<const float4*>(colsum),
                                                                                         c_out / 4, reinterpret_cast<float4*>(d_bias));
    PAPR_CHECK_LAUNCH("upconv_wgrad_reduce");
    return 0;
}

size_t papr_i_upconv_wgrad_bytes(long M, int c_in, int c_out) {
    const long px = up_px_per_chunk(M, c_in, c_out), chunks = (M + px - 1) / px;
    return (size_t)chunks * 4 * c_out * sizeof(float) + (size_t)chunks * c_out * 4 * c_in * sizeof(float);
}

// (maxima from the producers, the bias gradient from the launch's own column sums: no statistics launch)
int papr_i_upconv_wgrad(const float* g, int ldg, const float* x, int B, int H, int W, int c_in, int c_out, const unsigned* xmax, const unsigned* gmax, float* d_wm,
                        float* d_bias, void* ws, bool one_product, hipStream_t s) {
    const long M = (long)B * H * W, px = up_px_per_chunk(M, c_in, c_out), chunks = (M + px - 1) / px;
    UpArgs a{};
    a.x = x; a.g = g; a.B = B; a.H = H; a.W = W; a.C_in = c_in; a.C_out = c_out; a.px_per_chunk = (int)px; a.ldg = ldg; a.ldo = c_out;
    a.pmax_bits = xmax; a.qmax_bits = gmax;
    float* partial_b = static_cast<float*>(ws);
    a.partial_b = d_bias ? partial_b : nullptr;
    a.out = partial_b + (size_t)chunks * 4 * c_out;
    const dim3 grid((unsigned)(c_in / UP_T), (unsigned)(4 * c_out / UP_T), (unsigned)chunks);
    if (one_product) upconv2x2_h3_kernel<2, true><<<grid, dim3(256), 0, s>>>(a);
    else upconv2x2_h3_kernel<2, false><<<grid, dim3(256), 0, s>>>(a);
    PAPR_CHECK_LAUNCH("upconv2x2_h3<wgrad>");
    const long n4 = (long)c_in * c_out;
    upconv_wgrad_reduce2_kernel<<<dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s>>>(reinterpret_cast<const float4*>(a.out), (int)chunks, n4,
                                                                                          reinterpret_cast<float4*>(d_wm), reinterpret_cast<const float4*>(a.partial_b),
                                                                                          c_out / 4, reinterpret_cast<float4*>(d_bias));
    PAPR_CHECK_LAUNCH("upconv_wgrad_reduce");
    return 0;
}

extern "C" int papr_conv1x1_fwd(const float* x, int64_t M, int32_t c_in, const float* w, const float* bias, int32_t c_out, float* out, papr_stream_t stream) {
    PAPR_REQUIRE(x && w && out, "papr_conv1x1_fwd: null pointer");
    PAPR_REQUIRE(M >= 1 && c_in >= 4 && c_in % 4 == 0 && c_out >= 1 && c_out <= C1_MAXN, "papr_conv1x1_fwd: M %lld, c_in %d (multiple of 4), c_out %d (1 .. 4)",
                 (long long)M, c_in, c_out);
    conv1x1_fwd_kernel<<<dim3((unsigned)((M + 7) / 8)), dim3(256), 0, as_stream(stream)>>>(x, M, c_in, w, bias, c_out, out);
    PAPR_CHECK_LAUNCH("conv1x1_fwd");
    return 0;
}

extern "C" size_t papr_conv1x1_bwd_workspace_bytes(int64_t M, int32_t c_in, int32_t c_out) {
    const long px = c1_px_per_chunk(M), chunks = (M + px - 1) / px;
    return (size_t)chunks * c_out * (c_in + 1) * sizeof(float);
}

int papr_i_conv1x1_bwd(const float* d_out, const float* x, long M, int c_in, const float* w, int c_out, const float* mask_y, float* d_x, unsigned* out_max,
                       float* d_w, float* d_b, void* ws, hipStream_t s) {
    if (d_x) {
        const long n4 = M * (c_in / 4);
        const long nb = (n4 + 255) / 256;
        conv1x1_dgrad_kernel<<<dim3((unsigned)(out_max && nb > 1024 ? 1024 : nb)), dim3(256), 0, s>>>(d_out, M, c_in, w, c_out, d_x, mask_y, out_max);
        PAPR_CHECK_LAUNCH("conv1x1_dgrad");
    }
    if (d_w) {
        const long px = c1_px_per_chunk(M), chunks = (M + px - 1) / px;
        float* partial = static_cast<float*>(ws);
        float* partial_b = d_b ? partial + (size_t)chunks * c_out * c_in : nullptr;
        conv1x1_wgrad_kernel<<<dim3((unsigned)chunks), dim3(256), 0, s>>>(d_out, x, M, c_in, c_out, px, partial, partial_b);
        PAPR_CHECK_LAUNCH("conv1x1_wgrad");
        const int NC = c_out * c_in;
        conv1x1_wgrad_reduce_kernel<<<dim3((unsigned)((NC + c_out + 3) / 4)), dim3(256), 0, s>>>(partial, partial_b, (int)chunks, NC, c_out, d_w, d_b);
        PAPR_CHECK_LAUNCH("conv1x1_wgrad_reduce");
    }
    return 0;
}

extern "C" int papr_conv1x1_bwd(const float* d_out, const float* x, int64_t M, int32_t c_in, const float* w, int32_t c_out, float* d_x, float* d_w,
                                float* d_bias, void* workspace, papr_stream_t stream) {
    PAPR_REQUIRE(d_out && x && w, "papr_conv1x1_bwd: null pointer");
    PAPR_REQUIRE(M >= 1 && c_in >= 32 && c_in <= 256 && (c_in & (c_in - 1)) == 0 && c_out >= 1 && c_out <= C1_MAXN,
                 "papr_conv1x1_bwd: M %lld, c_in %d (32, 64, 128 or 256), c_out %d (1 .. 4)", (long long)M, c_in, c_out);
    PAPR_REQUIRE(!d_w || workspace, "papr_conv1x1_bwd: the weight gradient needs the workspace");
    return papr_i_conv1x1_bwd(d_out, x, M, c_in, w, c_out, nullptr, d_x, nullptr, d_w, d_bias, workspace, as_stream(stream));
}
