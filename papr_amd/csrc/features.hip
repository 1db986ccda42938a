// K2: gather the k selected points of every ray, decompose p - o into the along-ray part s and the
// perpendicular part u, and write the positionally-encoded key / query / value input rows.
//
// Replaces (reference): points[idx] / pc_feats[idx] (models/model.py:330,431-435),
// _calculate_distances + normalize_vector (models/model.py:285-310, models/utils.py:255-257),
// the feature lists of _get_kqv (models/model.py:396-437), posenc (models/utils.py:232-242) and the
// concatenations of Embeddings.forward (models/attn.py:173-191) -- about forty eager kernels and
// ~0.8 GB of temporaries per 25,600-ray step -- with one pass that reads 12+4 B per pair and writes
// each row exactly once.
//
// One thread owns one (ray, neighbour) pair.  sin/cos use the full-range ocml sincosf: arguments
// reach 2^5 * |x| ~ 1.6e3 rad, so the fast hardware approximations are not an option for parity.
// The backward pass recomputes the encodings (cheaper than saving 2 x M x 78 floats) and folds
// d/dx [x, sin(f x), cos(f x)] = [1, f cos(f x), -f sin(f x)] and the geometry Jacobian into a
// single atomic scatter onto the point gradients.
#include "papr_common.h"
#include <algorithm>

namespace {

struct FeatParams {
    papr_feature_desc d;
    int key_w, qry_w, val_w;
};

__device__ __forceinline__ int pe_width(int L, int with_self) { return 3 * (with_self + 2 * L); }

// write pe(x) for one 3-vector at dst (stride 1), returns number of floats written; s1 / s2 gather the sum of the values written and of their squares
// The sums behind the one-pass statistics of a row: the sines and cosines (|v| <= 1, ~2 L per component) add up in fp32 without a rounding that
// matters; the RAW coordinates (with_self; +-50 and more in the Tanks&Temples scenes, squares of 2,500 beside 108 values of size one) go into
// doubles, where their squares are exact -- nine double operations per vector instead of ~240 for the whole row (which cost features_fwd 26 us)
struct PeSums { float t1, t2; double r1, r2; };
__device__ __forceinline__ int write_pe(float* __restrict__ dst, const float x[3], int L, int with_self,
                                        float factor, float mult, PeSums& a) {
    const int per = with_self + 2 * L;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float* o = dst + c * per;
        if (with_self) { *o++ = x[c]; a.r1 += (double)x[c]; a.r2 += (double)x[c] * (double)x[c]; }
        float f = 1.0f;
        for (int i = 0; i < L; ++i) {
            float s, co;
            sincosf((f * x[c]) * mult, &s, &co);
            o[2 * i] = s;
            o[2 * i + 1] = co;
            a.t1 += s + co;
            a.t2 += s * s + co * co;
            f *= factor;
        }
    }
    return 3 * per;
}
__device__ __forceinline__ int write_pe(float* __restrict__ dst, const float x[3], int L, int with_self, float factor, float mult) {
    PeSums a = {0.f, 0.f, 0.0, 0.0};
    return write_pe(dst, x, L, with_self, factor, mult, a);
}

struct RayGeom {
    float ox, oy, oz;   // origin
    float rx, ry, rz;   // re-normalised direction r = d / (|d| + eps)
    float rr;           // r.r + eps
};

__device__ __forceinline__ RayGeom load_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                            long r, long rays_per_image, float eps) {
    RayGeom g;
    long n = r / rays_per_image;
    g.ox = rays_o[n * 3 + 0]; g.oy = rays_o[n * 3 + 1]; g.oz = rays_o[n * 3 + 2];
    float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
    // torch.norm: sqrt(fma(z,z, fma(y,y, x*x))) (measured against torch CPU)
    float nrm = sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx))) + eps;
    g.rx = dx / nrm; g.ry = dy / nrm; g.rz = dz / nrm;
    g.rr = ((g.rx * g.rx + g.ry * g.ry) + g.rz * g.rz) + eps;
    return g;
}

// Rows are 117-142 floats: a thread that owns a row and reads or writes it directly touches 64 different cache lines
// per instruction (PMC: 2.5-3x the algorithmic traffic, 4-byte pieces).  So each wave stages one encoding block
// (39 floats for L = 6) of its 64 rows in LDS -- odd row pitch, conflict-free for "same column, 64 rows" -- and moves
// it to / from memory with consecutive lanes on consecutive addresses.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// buf[r][c], r < 64, c < w  ->  dst[(m0 + r) * ld + c0 + c]
// (four floats per store instruction: the block's place in a row is only 4-byte aligned, which a global dwordx4 store accepts)
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void wave_flush(const float* buf, int pitch, int w, float* __restrict__ dst, long ld, int c0, long m0, long M) {
    const int lane = threadIdx.x & 63, wq = (w + 3) >> 2, total = 64 * wq;
    const float inv_wq = 1.0f / (float)wq;
    for (int e = lane; e < total; e += 64) {
        const int r = (int)(((float)e + 0.5f) * inv_wq), c = 4 * (e - r * wq);
        if (m0 + r >= M) continue;
        const float* b = buf + r * pitch + c;
        float* d = dst + (m0 + r) * ld + c0 + c;
        if (c + 3 < w) *reinterpret_cast<f32x4_a4*>(d) = f32x4_a4{b[0], b[1], b[2], b[3]};
        else for (int j = 0; c + j < w; ++j) d[j] = b[j];
    }
}
// src[(m0 + r) * ld + c0 + c]  ->  buf[r][c]   (rows beyond M read row M-1)
// (four floats per load instruction, like wave_flush, and several instructions' loads in flight before their LDS writes: a float per lane and
//  iteration was 39 dependent load -> write steps per block.  The last group of a block may read up to three floats of the next block or of the
//  row's padding: inside the row while c0 + 4 ceil(w / 4) <= ld, which the caller's rows satisfy; otherwise float by float.)
__device__ __forceinline__ void wave_fetch(float* buf, int pitch, int w, const float* __restrict__ src, long ld, int c0, long m0, long M) {
    const int lane = threadIdx.x & 63;
    const int wq = (w + 3) >> 2, total = 64 * wq;
    if (c0 + 4 * wq <= ld) {
        const float inv_wq = 1.0f / (float)wq;
#pragma unroll 4
        for (int e = lane; e < total; e += 64) {
            const int r = (int)(((float)e + 0.5f) * inv_wq), c = 4 * (e - r * wq);
            const long m = m0 + r < M ? m0 + r : M - 1;
            const f32x4_a4 v = *reinterpret_cast<const f32x4_a4*>(src + m * ld + c0 + c);
            float* b = buf + r * pitch + c;
            b[0] = v.x;
            if (c + 1 < w) b[1] = v.y;
            if (c + 2 < w) b[2] = v.z;
            if (c + 3 < w) b[3] = v.w;
        }
        return;
    }
    const float inv_w = 1.0f / (float)w;
    for (int e = lane; e < 64 * w; e += 64) {
        const int r = (int)(((float)e + 0.5f) * inv_w), c = e - r * w;
        const long m = m0 + r < M ? m0 + r : M - 1;
        buf[r * pitch + c] = src[m * ld + c0 + c];
    }
}
// feature rows of the 64 selected points -> dst[(m0 + r) * ld + c0 + 0..feat_dim)
__device__ __forceinline__ void wave_copy_feats(const float* __restrict__ pc_feats, int feat_dim, int pi, float* __restrict__ dst, long ld, int c0,
                                                long m0, long M) {
    const int lane = threadIdx.x & 63;
    const int q4 = feat_dim >> 2;                    // 16-byte pieces per row
    if ((feat_dim & 3) == 0 && q4 >= 1 && q4 <= 64 && (64 % q4) == 0) {
        // 64 / q4 rows per instruction, four instructions' loads in flight before their stores (a row per instruction, load then store, was a
        // chain of 64 round trips per wave: ~half of what features_fwd spent outside its encoding blocks)
        const int rp = 64 / q4, c4 = lane % q4, rl = lane / q4;
        for (int r0 = 0; r0 < 64; r0 += 4 * rp) {
            float4 v[4];
            int row[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                row[u] = r0 + u * rp + rl;
                const int pr = __shfl(pi, row[u] & 63, 64);
                v[u] = *reinterpret_cast<const float4*>(pc_feats + (long)pr * feat_dim + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (row[u] < 64 && m0 + row[u] < M)
                    *reinterpret_cast<f32x4_a4*>(dst + (m0 + row[u]) * ld + c0 + 4 * c4) = f32x4_a4{v[u].x, v[u].y, v[u].z, v[u].w};
        }
        return;
    }
    for (int r = 0; r < 64; ++r) {
        const int pr = __builtin_amdgcn_readlane(pi, r);
        if (m0 + r >= M) break;
        for (int c = lane; c < feat_dim; c += 64) dst[(m0 + r) * ld + c0 + c] = pc_feats[(long)pr * feat_dim + c];
    }
}
// zero columns [c0, c1) of the 64 rows
__device__ __forceinline__ void wave_zero_cols(float* __restrict__ dst, long ld, int c0, int c1, long m0, long M) {
    const int lane = threadIdx.x & 63, w = c1 - c0;
    if (w <= 0) return;
    for (int e = lane; e < 64 * w; e += 64) {
        const int r = e / w, c = e - r * w;
        if (m0 + r < M) dst[(m0 + r) * ld + c0 + c] = 0.f;
    }
}

__global__ __launch_bounds__(256) void features_fwd_kernel(FeatParams fp, const float* __restrict__ points,
                                                           const float* __restrict__ pc_feats,
                                                           const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d, long R,
                                                           long rays_per_image, const int* __restrict__ idx,
                                                           float* __restrict__ key, float* __restrict__ val,
                                                           float* __restrict__ sel_points, int pitch,
                                                           float* __restrict__ key_stats, float* __restrict__ key_mean, float key_eps) {
    extern __shared__ float feat_lds[];
    const papr_feature_desc& d = fp.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* buf = feat_lds + wave * 64 * pitch;
    float* row = buf + lane * pitch;
    const long M = R * d.k;
    const long m0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) - lane;     // first pair of this wave
    if (m0 >= M) return;                                                        // (whole wave)
    const long mreal = m0 + lane;
    const long m = mreal < M ? mreal : M - 1;                                   // idle lanes shadow the last pair: the wave works as one
    long r = m / d.k;
    RayGeom g = load_ray(rays_o, rays_d, r, rays_per_image, d.eps);
    int pi = idx[m];
    float p[3] = {points[pi * 3 + 0], points[pi * 3 + 1], points[pi * 3 + 2]};
    if (sel_points && mreal < M) { sel_points[m * 3 + 0] = p[0]; sel_points[m * 3 + 1] = p[1]; sel_points[m * 3 + 2] = p[2]; }
    float vx = p[0] - g.ox, vy = p[1] - g.oy, vz = p[2] - g.oz;
    float t = ((vx * g.rx + vy * g.ry) + vz * g.rz) / g.rr;
    float s[3] = {g.rx * t, g.ry * t, g.rz * t};
    float u[3] = {vx - s[0], vy - s[1], vz - s[2]};

    PeSums sums = {0.f, 0.f, 0.0, 0.0};             // sum and sum of squares of the row being written (this thread's)
    auto emit = [&](const float x[3], int L, float* dst, long ld, int c0) -> int {
        const int w = write_pe(row, x, L, d.with_self, d.pe_factor, d.pe_mult, sums);
        wave_lds_sync();
        wave_flush(buf, pitch, w, dst, ld, c0, m0, M);
        wave_lds_sync();
        return w;
    };
    // pe(s) and pe(u) stand in the key row AND in the value row: where the two rows ask for the same orders (every shipped configuration), an
    // encoding block is computed once and leaves the staging buffer twice -- 54 instead of 90 full-range sincosf per pair (the kernel was bound
    // by them: 1.2 ms per 160,000-ray chunk of a render against 0.6 ms of row writes)
    const bool shared = d.L_key[1] == d.L_val[0] && d.L_key[2] == d.L_val[1];
    auto emit2 = [&](const float x[3], int L, int c_key, int c_val) -> int {
        const int w = write_pe(row, x, L, d.with_self, d.pe_factor, d.pe_mult, sums);
        wave_lds_sync();
        wave_flush(buf, pitch, w, key, d.ld_key, c_key, m0, M);
        wave_flush(buf, pitch, w, val, d.ld_val, c_val, m0, M);
        wave_lds_sync();
        return w;
    };
    int c = 0, cv = 0;
    c += emit(p, d.L_key[0], key, d.ld_key, c);
    if (shared) {
        const int ws = emit2(s, d.L_key[1], c, 0);
        c += ws;
        cv = ws + emit2(u, d.L_key[2], c, ws);
        c += cv - ws;
    } else {
        c += emit(s, d.L_key[1], key, d.ld_key, c);
        c += emit(u, d.L_key[2], key, d.ld_key, c);
    }
    if (key_stats && mreal < M) {
        // the statistics of the LayerNorm core in front of the key MLP (FeedForward.innorm, models/attn.py:39-42: unbiased std, eps added to it), while
        // the row's values are in this thread's hands: the fused run that stages the rows then applies them (papr_row_norm.given_mean) instead of
        // taking two wave sums, a square root and a division per row in its staging slot -- 8k of that slot's 12-15k cycles.  One pass (PeSums): the
        // difference below is taken in double, where the large raw coordinates were added
        const double nd = (double)c, s1 = sums.r1 + (double)sums.t1, s2 = sums.r2 + (double)sums.t2, mean_d = s1 / nd;
        const double var_d = (s2 - nd * mean_d * mean_d) / (nd - 1.0);
        const float mean = (float)mean_d;
        const float sigma = sqrtf((float)(var_d > 0.0 ? var_d : 0.0));
        key_mean[m] = mean;
        key_stats[2 * m] = 1.0f / (sigma + key_eps);
        key_stats[2 * m + 1] = sigma;
    }
    if (d.key_has_feats) { wave_copy_feats(pc_feats, d.feat_dim, pi, key, d.ld_key, c, m0, M); c += d.feat_dim; }
    wave_zero_cols(key, d.ld_key, c, d.ld_key, m0, M);

    c = cv;
    if (!shared) {
        PeSums unused = sums;                       // (the value row's encodings do not belong to the key row's statistics)
        c += emit(s, d.L_val[0], val, d.ld_val, c);
        c += emit(u, d.L_val[1], val, d.ld_val, c);
        sums = unused;
    }
    if (d.val_has_feats) { wave_copy_feats(pc_feats, d.feat_dim, pi, val, d.ld_val, c, m0, M); c += d.feat_dim; }
    wave_zero_cols(val, d.ld_val, c, d.ld_val, m0, M);
}

__global__ __launch_bounds__(256) void query_fwd_kernel(FeatParams fp, const float* __restrict__ rays_d, long R,
                                                        float* __restrict__ qry) {
    const papr_feature_desc& d = fp.d;
    long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float dv[3] = {rays_d[r * 3 + 0], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
    float* q = qry + r * d.ld_qry;
    int c = write_pe(q, dv, d.L_qry, d.with_self, d.pe_factor, d.pe_mult);
    for (; c < d.ld_qry; ++c) q[c] = 0.f;
}

// gradient of the loss w.r.t. a 3-vector x from the gradient of pe(x)
__device__ __forceinline__ void pe_grad(const float* __restrict__ g, const float x[3], int L, int with_self,
                                        float factor, float mult, float out[3]) {
    const int per = with_self + 2 * L;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* gc = g + c * per;
        float acc = 0.f;
        if (with_self) acc = *gc++;
        float f = 1.0f;
        for (int i = 0; i < L; ++i) {
            float s, co;
            sincosf((f * x[c]) * mult, &s, &co);
            acc += (f * mult) * (gc[2 * i] * co - gc[2 * i + 1] * s);
            f *= factor;
        }
        out[c] += acc;
    }
}

// The same through a LayerNorm core in front (FeedForward.innorm of the key MLP, models/attn.py:39-42): g is the gradient w.r.t. the STANDARDISED
// row y = (pe - mean) * rinv, and what is wanted is the gradient w.r.t. x through  d_pe = rinv (g - mean(g)) - y sum(g y) / ((n - 1) sigma)
// (rownorm_bwd_kernel's formula).  pe_grad is linear in its gradient row, so  out(d_pe) = rinv (F(g) - mean(g) F(1)) - coef F(y)  with F the
// functional above: this routine gathers F(g), F(1), F(y) for one 3-vector and the row's two sums over THIS block -- the caller combines them
// once every block of the row has been seen.  y is recomputed from the sines and cosines the coefficients need anyway (the bits features_fwd
// wrote and the fused run standardised), so the standardised key rows are not read at all.
struct LnSums { float sg, sgy; };
__device__ __forceinline__ void pe_grad_ln(const float* __restrict__ g, const float x[3], int L, int with_self, float factor, float mult,
                                           float mean, float rinv, LnSums& sums, float Fg[3], float F1[3], float Fy[3]) {
    const int per = with_self + 2 * L;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* gc = g + c * per;
        float ag = 0.f, a1 = 0.f, ay = 0.f;
        if (with_self) {
            const float g0 = *gc++, y0 = (x[c] - mean) * rinv;
            sums.sg += g0; sums.sgy += g0 * y0;
            ag = g0; a1 = 1.f; ay = y0;
        }
        float f = 1.0f;
        for (int i = 0; i < L; ++i) {
            float sn, co;
            sincosf((f * x[c]) * mult, &sn, &co);
            const float ys = (sn - mean) * rinv, yc = (co - mean) * rinv, gsn = gc[2 * i], gco = gc[2 * i + 1], kf = f * mult;
            sums.sg += gsn + gco;
            sums.sgy += gsn * ys + gco * yc;
            ag += kf * (gsn * co - gco * sn);
            a1 += kf * (co - sn);
            ay += kf * (ys * co - yc * sn);
            f *= factor;
        }
        Fg[c] += ag; F1[c] += a1; Fy[c] += ay;
    }
}

__global__ __launch_bounds__(256) void features_bwd_kernel(FeatParams fp, const float* __restrict__ points,
                                                           const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d, long R,
                                                           long rays_per_image, const int* __restrict__ idx,
                                                           const float* __restrict__ d_key,
                                                           const float* __restrict__ d_val,
                                                           float* __restrict__ d_points, float4* __restrict__ d_pair, int pitch,
                                                           const float* __restrict__ key_mean, const float* __restrict__ key_stats) {
    extern __shared__ float feat_lds[];
    const papr_feature_desc& d = fp.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* buf = feat_lds + wave * 64 * pitch;
    const float* row = buf + lane * pitch;
    const long M = R * d.k;
    const long m0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) - lane;
    if (m0 >= M) return;
    const long mreal = m0 + lane;
    const long m = mreal < M ? mreal : M - 1;
    long r = m / d.k;
    RayGeom g = load_ray(rays_o, rays_d, r, rays_per_image, d.eps);
    int pi = idx[m];
    float vx = points[pi * 3 + 0] - g.ox, vy = points[pi * 3 + 1] - g.oy, vz = points[pi * 3 + 2] - g.oz;
    float t = ((vx * g.rx + vy * g.ry) + vz * g.rz) / g.rr;
    float s[3] = {g.rx * t, g.ry * t, g.rz * t};
    float u[3] = {vx - s[0], vy - s[1], vz - s[2]};
    float gs[3] = {0.f, 0.f, 0.f}, gu[3] = {0.f, 0.f, 0.f};
    // one encoding block of the 64 gradient rows at a time through LDS (see features_fwd_kernel)
    auto absorb = [&](const float* src, long ld, int c0, const float x[3], int L, float out[3]) -> int {
        const int w = pe_width(L, d.with_self);
        wave_fetch(buf, pitch, w, src, ld, c0, m0, M);
        wave_lds_sync();
        pe_grad(row, x, L, d.with_self, d.pe_factor, d.pe_mult, out);
        wave_lds_sync();
        return w;
    };
    if (d_key && key_mean) {
        // d_key is the gradient w.r.t. the STANDARDISED key rows: the LayerNorm core's backward pass rides here (before: papr_rownorm_bwd, a pass that
        // read the gradient rows and the standardised rows and wrote the gradient rows back: 720 MB per step).  The pe(p) block gets no gradient
        // (points.detach(), models/model.py:405) but its columns belong to the row's two sums.
        const float mean = key_mean[m], rinv = key_stats[2 * m], sigma = key_stats[2 * m + 1];
        const float p3[3] = {points[pi * 3 + 0], points[pi * 3 + 1], points[pi * 3 + 2]};
        LnSums sums = {0.f, 0.f};
        float Fg[3][3] = {}, F1[3][3] = {}, Fy[3][3] = {};
        auto absorb_ln = [&](int c0, const float x[3], int L, int b) -> int {
            const int w = pe_width(L, d.with_self);
            wave_fetch(buf, pitch, w, d_key, d.ld_key, c0, m0, M);
            wave_lds_sync();
            pe_grad_ln(row, x, L, d.with_self, d.pe_factor, d.pe_mult, mean, rinv, sums, Fg[b], F1[b], Fy[b]);
            wave_lds_sync();
            return w;
        };
        int c = absorb_ln(0, p3, d.L_key[0], 0);
        c += absorb_ln(c, s, d.L_key[1], 1);
        c += absorb_ln(c, u, d.L_key[2], 2);
        const float n = (float)c, gbar = sums.sg / n;
        const float coef = sigma > 0.f ? sums.sgy / ((n - 1.f) * sigma) : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            gs[j] += rinv * (Fg[1][j] - gbar * F1[1][j]) - coef * Fy[1][j];
            gu[j] += rinv * (Fg[2][j] - gbar * F1[2][j]) - coef * Fy[2][j];
        }
    } else if (d_key) {
        int c = pe_width(d.L_key[0], d.with_self);                       // skip pe(p): detached
        c += absorb(d_key, d.ld_key, c, s, d.L_key[1], gs);
        absorb(d_key, d.ld_key, c, u, d.L_key[2], gu);
    }
    if (d_val) {
        int c = absorb(d_val, d.ld_val, 0, s, d.L_val[0], gs);
        absorb(d_val, d.ld_val, c, u, d.L_val[1], gu);
    }
    if (mreal >= M) return;
    // s = r t, u = v - r t, t = (v.r)/(r.r+eps)  =>  dL/dv = gu + r * (r.(gs - gu)) / (r.r+eps)
    float w = (g.rx * (gs[0] - gu[0]) + g.ry * (gs[1] - gu[1]) + g.rz * (gs[2] - gu[2])) / g.rr;
    if (d_pair) {           // per-pair rows for the deterministic segmented reduction (papr_segment_reduce)
        d_pair[m] = make_float4(gu[0] + g.rx * w, gu[1] + g.ry * w, gu[2] + g.rz * w, 0.f);
    } else {
        unsafeAtomicAdd(d_points + pi * 3 + 0, gu[0] + g.rx * w);
        unsafeAtomicAdd(d_points + pi * 3 + 1, gu[1] + g.ry * w);
        unsafeAtomicAdd(d_points + pi * 3 + 2, gu[2] + g.rz * w);
    }
}

// scatter-add of the un-encoded per-point feature block: 4 columns per thread
__global__ __launch_bounds__(256) void feats_scatter_kernel(const float* __restrict__ grad, int ld, int col0,
                                                            int feat_dim, const int* __restrict__ idx, long M,
                                                            float* __restrict__ d_pc_feats) {
    const int groups = feat_dim / 4;
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * groups) return;
    long m = e / groups;
    int c = (int)(e - m * groups) * 4;
    const float* src = grad + m * ld + col0 + c;
    float* dst = d_pc_feats + (long)idx[m] * feat_dim + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) unsafeAtomicAdd(dst + j, src[j]);
}

// Gradient of per-point parameters as a segmented sum over the pairs that selected each point.
// `order` lists pair ids grouped by point (stable sort of idx), `sorted_pts` the point of each entry,
// seg[p]..seg[p+1] point p's group.  Work is cut into fixed chunks of SEG_CH sorted entries, one wave
// per chunk (a wave-per-point split is hopelessly unbalanced: a few points are selected by thousands
// of rays).  Lanes span the feature columns, lanes 0-2 also carry xyz and lane 3 the influence term.
// A group that lies inside one chunk is stored directly; a group that straddles chunks is finished
// by a second small kernel that adds the chunks' shares in chunk order (no atomics: bit-reproducible).  Outputs must be zeroed by the caller.
constexpr int SEG_CH = 128;
constexpr int SEG_PW = 4 + 128;                   // floats of one parked share: point gradient (3) + influence (1) + up to 128 feature columns

__global__ __launch_bounds__(256) void segment_reduce_kernel(const long* __restrict__ order, const int* __restrict__ sorted_pts,
                                                             const long* __restrict__ seg, long M,
                                                             const float* __restrict__ pair_points, const float* __restrict__ pair_influ,
                                                             const float* __restrict__ rows, int ld, int col0, int ncols,
                                                             float* __restrict__ d_points, float* __restrict__ d_influ,
                                                             float* __restrict__ d_feats, int accumulate, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const long chunk = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long e0 = chunk * SEG_CH;
    if (e0 >= M) return;
    const int n = (int)(M - e0 < SEG_CH ? M - e0 : SEG_CH);
    // this chunk's entries: lane holds entries lane and lane+64
    int ma = e0 + lane < M ? (int)order[e0 + lane] : 0, mb = e0 + 64 + lane < M ? (int)order[e0 + 64 + lane] : 0;
    int pa = e0 + lane < M ? sorted_pts[e0 + lane] : 0, pb = e0 + 64 + lane < M ? sorted_pts[e0 + 64 + lane] : 0;
    const bool has_small = (pair_points && lane < 3) || (pair_influ && lane == 3);
    const bool c0_ok = rows && lane < ncols, c1_ok = rows && lane + 64 < ncols;
    float small = 0.f, acc0 = 0.f, acc1 = 0.f;
    int cur = __builtin_amdgcn_readlane(pa, 0);

    auto flush = [&](int p) {
        const bool interior = seg[p] >= e0 && seg[p + 1] <= e0 + n;
        if (interior) {               // the group lies wholly in this chunk: this wave alone writes the point's row
            if (accumulate) {         // (a second pass over the same outputs: features that feed both the key and the value branch)
                if (pair_points && lane < 3) small += d_points[p * 3 + lane];
                if (pair_influ && lane == 3) small += d_influ[p];
                if (c0_ok) acc0 += d_feats[(long)p * ncols + lane];
                if (c1_ok) acc1 += d_feats[(long)p * ncols + lane + 64];
            }
            if (pair_points && lane < 3) d_points[p * 3 + lane] = small;
            if (pair_influ && lane == 3) d_influ[p] = small;
            if (c0_ok) d_feats[(long)p * ncols + lane] = acc0;
            if (c1_ok) d_feats[(long)p * ncols + lane + 64] = acc1;
        } else {
            // the group goes on in a neighbouring chunk: this chunk's share is parked (slot 0: the group came in from the chunk before, slot 1: it
            // starts here and runs on) and segment_finish_kernel adds the shares in chunk order -- with atomic adds three or more shares of a
            // popular point arrived in any order and training was not bit-reproducible from run to run
            float* const ps = partial + (chunk * 2 + (seg[p] < e0 ? 0 : 1)) * SEG_PW;
            if (lane < 4) ps[lane] = small;
            ps[4 + lane] = acc0;
            ps[4 + 64 + lane] = acc1;
        }
        small = acc0 = acc1 = 0.f;
    };
    auto entry = [&](int j, int& m, int& pt) {       // wave-uniform (pair id, point) of entry j
        m = j < 64 ? __builtin_amdgcn_readlane(ma, j) : __builtin_amdgcn_readlane(mb, j - 64);
        pt = j < 64 ? __builtin_amdgcn_readlane(pa, j) : __builtin_amdgcn_readlane(pb, j - 64);
    };
    constexpr int SEG_U = 8;                         // entries whose rows are in flight together (the rows are a gather over ~130 MB: a round trip each;
                                                     // four at a time: 62 us per step, eight: see DESIGN section 6)
    for (int j0 = 0; j0 < n; j0 += SEG_U) {
        int m[SEG_U], pt[SEG_U];
        float vs[SEG_U], v0[SEG_U], v1[SEG_U];
#pragma unroll
        for (int u = 0; u < SEG_U; ++u) {            // issue the loads of the batch's entries before using any
            entry(j0 + u < n ? j0 + u : n - 1, m[u], pt[u]);
            vs[u] = v0[u] = v1[u] = 0.f;
            if (has_small) vs[u] = lane < 3 ? pair_points[(long)m[u] * 4 + lane] : pair_influ[m[u]];
            if (c0_ok) v0[u] = rows[(long)m[u] * ld + col0 + lane];
            if (c1_ok) v1[u] = rows[(long)m[u] * ld + col0 + lane + 64];
        }
#pragma unroll
        for (int u = 0; u < SEG_U; ++u) {
            if (j0 + u >= n) break;
            if (pt[u] != cur) { flush(cur); cur = pt[u]; }
            small += vs[u]; acc0 += v0[u]; acc1 += v1[u];
        }
    }
    flush(cur);
}

// one wave per point whose group of pairs spans several chunks: the chunks' parked shares, added in chunk order
__global__ __launch_bounds__(256) void segment_finish_kernel(const long* __restrict__ seg, long P, int has_points, int has_influ, int ncols,
                                                             float* __restrict__ d_points, float* __restrict__ d_influ,
                                                             float* __restrict__ d_feats, int accumulate, const float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const long a = seg[p], b = seg[p + 1];
    if (b <= a) return;
    const long c0 = a / SEG_CH, c1 = (b - 1) / SEG_CH;
    if (c0 == c1) return;                           // (wholly inside one chunk: segment_reduce_kernel wrote it)
    float small = 0.f, acc0 = 0.f, acc1 = 0.f;
    for (long c = c0; c <= c1; ++c) {
        const float* ps = partial + (c * 2 + (a < c * SEG_CH ? 0 : 1)) * SEG_PW;
        if (lane < 4) small += ps[lane];
        acc0 += ps[4 + lane];
        acc1 += ps[4 + 64 + lane];
    }
    const bool c0_ok = ncols > 0 && lane < ncols, c1_ok = ncols > 0 && lane + 64 < ncols;
    if (accumulate) {
        if (has_points && lane < 3) small += d_points[p * 3 + lane];
        if (has_influ && lane == 3) small += d_influ[p];
        if (c0_ok) acc0 += d_feats[p * ncols + lane];
        if (c1_ok) acc1 += d_feats[p * ncols + lane + 64];
    }
    if (has_points && lane < 3) d_points[p * 3 + lane] = small;
    if (has_influ && lane == 3) d_influ[p] = small;
    if (c0_ok) d_feats[p * ncols + lane] = acc0;
    if (c1_ok) d_feats[p * ncols + lane + 64] = acc1;
}

// LDS row pitch of the per-wave staging buffer: the widest encoding block, made odd
int stage_pitch(const papr_feature_desc* d) {
    int w = 1;
    for (int i = 0; i < 3; ++i) w = std::max(w, 3 * (d->with_self + 2 * d->L_key[i]));
    for (int i = 0; i < 2; ++i) w = std::max(w, 3 * (d->with_self + 2 * d->L_val[i]));
    return w | 1;
}

int fill_params(const papr_feature_desc* d, FeatParams* fp) {
    fp->d = *d;
    fp->key_w = 0;
    for (int i = 0; i < 3; ++i) fp->key_w += 3 * (d->with_self + 2 * d->L_key[i]);
    if (d->key_has_feats) fp->key_w += d->feat_dim;
    fp->qry_w = 3 * (d->with_self + 2 * d->L_qry);
    fp->val_w = 0;
    for (int i = 0; i < 2; ++i) fp->val_w += 3 * (d->with_self + 2 * d->L_val[i]);
    if (d->val_has_feats) fp->val_w += d->feat_dim;
    return 0;
}

}  // namespace

extern "C" int papr_feature_widths(const papr_feature_desc* d, int32_t* key_w, int32_t* qry_w, int32_t* val_w) {
    PAPR_REQUIRE(d, "papr_feature_widths: null descriptor");
    FeatParams fp;
    fill_params(d, &fp);
    if (key_w) *key_w = fp.key_w;
    if (qry_w) *qry_w = fp.qry_w;
    if (val_w) *val_w = fp.val_w;
    return 0;
}

static int check_desc(const papr_feature_desc* d, const FeatParams& fp, const char* who) {
    PAPR_REQUIRE(d->k >= 1, "%s: k must be >= 1", who);
    PAPR_REQUIRE(d->ld_key >= fp.key_w && d->ld_qry >= fp.qry_w && d->ld_val >= fp.val_w,
                 "%s: row strides (%d,%d,%d) smaller than widths (%d,%d,%d)", who, d->ld_key, d->ld_qry, d->ld_val,
                 fp.key_w, fp.qry_w, fp.val_w);
    PAPR_REQUIRE(d->ld_key % 4 == 0 && d->ld_qry % 4 == 0 && d->ld_val % 4 == 0, "%s: row strides must be multiples of 4", who);
    PAPR_REQUIRE(!(d->key_has_feats || d->val_has_feats) || d->feat_dim % 4 == 0, "%s: feat_dim must be a multiple of 4", who);
    return 0;
}

extern "C" int papr_build_features_fwd(const papr_feature_desc* d, const float* points, const float* pc_feats,
                                       const float* rays_o, const float* rays_d, int64_t R, int64_t rays_per_image,
                                       const int32_t* idx, float* key, float* qry, float* val, float* sel_points,
                                       float* key_stats, float* key_mean, float key_norm_eps, papr_stream_t stream) {
    PAPR_REQUIRE(d && points && rays_o && rays_d && idx && key && qry && val, "papr_build_features_fwd: null pointer");
    PAPR_REQUIRE((key_stats == nullptr) == (key_mean == nullptr), "papr_build_features_fwd: key_stats and key_mean come together");
    PAPR_REQUIRE(!key_stats || !d->key_has_feats, "papr_build_features_fwd: key statistics are taken over the encoded columns only (no point features in the key)");
    FeatParams fp;
    fill_params(d, &fp);
    if (int e = check_desc(d, fp, "papr_build_features_fwd")) return e;
    PAPR_REQUIRE(pc_feats || !(d->key_has_feats || d->val_has_feats), "papr_build_features_fwd: pc_feats required");
    if (R <= 0) return 0;
    hipStream_t s = as_stream(stream);
    long M = R * d->k;
    const int pitch = stage_pitch(d);
    PAPR_REQUIRE(pitch <= 129, "papr_build_features_fwd: encoding orders too large for the staging buffer");
    features_fwd_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), (size_t)4 * 64 * pitch * sizeof(float), s>>>(
        fp, points, pc_feats, rays_o, rays_d, R, rays_per_image, idx, key, val, sel_points, pitch, key_stats, key_mean, key_norm_eps);
    PAPR_CHECK_LAUNCH("features_fwd");
    query_fwd_kernel<<<dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s>>>(fp, rays_d, R, qry);
    PAPR_CHECK_LAUNCH("query_fwd");
    return 0;
}

extern "C" int papr_build_features_bwd(const papr_feature_desc* d, const float* points, const float* rays_o,
                                       const float* rays_d, int64_t R, int64_t rays_per_image, const int32_t* idx,
                                       const float* d_key, const float* d_val, float* d_points, float* d_pc_feats,
                                       papr_stream_t stream) {
    PAPR_REQUIRE(d && points && rays_o && rays_d && idx && d_points, "papr_build_features_bwd: null pointer");
    FeatParams fp;
    fill_params(d, &fp);
    if (int e = check_desc(d, fp, "papr_build_features_bwd")) return e;
    if (R <= 0) return 0;
    hipStream_t s = as_stream(stream);
    long M = R * d->k;
    const int pitch = stage_pitch(d);
    PAPR_REQUIRE(pitch <= 129, "papr_build_features_bwd: encoding orders too large for the staging buffer");
    features_bwd_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), (size_t)4 * 64 * pitch * sizeof(float), s>>>(
        fp, points, rays_o, rays_d, R, rays_per_image, idx, d_key, d_val, d_points, nullptr, pitch, nullptr, nullptr);
    PAPR_CHECK_LAUNCH("features_bwd");
    if (d_pc_feats) {
        long n = M * (d->feat_dim / 4);
        if (d->val_has_feats && d_val) {
            feats_scatter_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
                d_val, d->ld_val, fp.val_w - d->feat_dim, d->feat_dim, idx, M, d_pc_feats);
            PAPR_CHECK_LAUNCH("feats_scatter(val)");
        }
        if (d->key_has_feats && d_key) {
            feats_scatter_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
                d_key, d->ld_key, fp.key_w - d->feat_dim, d->feat_dim, idx, M, d_pc_feats);
            PAPR_CHECK_LAUNCH("feats_scatter(key)");
        }
    }
    return 0;
}

extern "C" int papr_build_features_bwd_pairs(const papr_feature_desc* d, const float* points, const float* rays_o,
                                             const float* rays_d, int64_t R, int64_t rays_per_image, const int32_t* idx,
                                             const float* d_key, const float* d_val, float* d_pair_points,
                                             const float* key_mean, const float* key_stats, papr_stream_t stream) {
    PAPR_REQUIRE(d && points && rays_o && rays_d && idx && d_pair_points, "papr_build_features_bwd_pairs: null pointer");
    PAPR_REQUIRE((key_mean == nullptr) == (key_stats == nullptr) && !(key_mean && (d->key_has_feats || !d_key)),
                 "papr_build_features_bwd_pairs: key_mean and key_stats come together, with d_key, and not with key_has_feats");
    FeatParams fp;
    fill_params(d, &fp);
    if (int e = check_desc(d, fp, "papr_build_features_bwd_pairs")) return e;
    if (R <= 0) return 0;
    long M = R * d->k;
    const int pitch = stage_pitch(d);
    PAPR_REQUIRE(pitch <= 129, "papr_build_features_bwd_pairs: encoding orders too large for the staging buffer");
    features_bwd_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), (size_t)4 * 64 * pitch * sizeof(float), as_stream(stream)>>>(
        fp, points, rays_o, rays_d, R, rays_per_image, idx, d_key, d_val, nullptr, reinterpret_cast<float4*>(d_pair_points), pitch, key_mean, key_stats);
    PAPR_CHECK_LAUNCH("features_bwd(pairs)");
    return 0;
}

extern "C" size_t papr_segment_reduce_workspace_bytes(int64_t M) {
    return (size_t)((M + SEG_CH - 1) / SEG_CH) * 2 * SEG_PW * sizeof(float);
}

extern "C" int papr_segment_reduce(const int64_t* order, const int32_t* sorted_pts, const int64_t* seg, int64_t M, int64_t P,
                                   const float* pair_points, const float* pair_influ, const float* rows, int ld, int col0,
                                   int ncols, float* d_points, float* d_influ, float* d_feats, int accumulate, void* workspace,
                                   papr_stream_t stream) {
    PAPR_REQUIRE(order && sorted_pts && seg && workspace, "papr_segment_reduce: null index arrays / workspace");
    PAPR_REQUIRE(M < ((int64_t)1 << 31), "papr_segment_reduce: more than 2^31 pairs");
    PAPR_REQUIRE(!rows || (d_feats && ncols >= 1 && ncols <= 128), "papr_segment_reduce: 1 <= ncols <= 128 and d_feats required");
    PAPR_REQUIRE(!pair_points || d_points, "papr_segment_reduce: d_points required");
    PAPR_REQUIRE(!pair_influ || d_influ, "papr_segment_reduce: d_influ required");
    if (P <= 0 || M <= 0) return 0;
    const long chunks = (M + SEG_CH - 1) / SEG_CH;
    segment_reduce_kernel<<<dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
        reinterpret_cast<const long*>(order), sorted_pts, reinterpret_cast<const long*>(seg), M, pair_points, pair_influ, rows, ld,
        col0, ncols, d_points, d_influ, d_feats, accumulate, static_cast<float*>(workspace));
    PAPR_CHECK_LAUNCH("segment_reduce");
    segment_finish_kernel<<<dim3((unsigned)((P + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
        reinterpret_cast<const long*>(seg), P, pair_points ? 1 : 0, pair_influ ? 1 : 0, rows ? ncols : 0, d_points, d_influ, d_feats, accumulate,
        static_cast<const float*>(workspace));
    PAPR_CHECK_LAUNCH("segment_finish");
    return 0;
}
