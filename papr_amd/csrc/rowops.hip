// Row-wise kernels of the render path: the LayerNorm core and the fused attention tail (K4).
// Both are HBM-streaming: one wave64 per row / per ray, 16-byte lane loads, shuffle reductions.
#include "papr_common.h"
#include "h3_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// y = (x - mean) / (std_unbiased + eps)      reference LayerNorm core, models/attn.py:39-42
// One wave per row; a lane holds up to VPL = 16 elements (width <= 1024), strided by 64 so that
// global accesses are coalesced.
constexpr int MAX_VPL = 16;

template <int VPL>
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* x, long rows, int width, int ld, float eps,
                                                          float* y, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ld;
    float v[VPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        int c = lane + 64 * i;
        v[i] = c < width ? xr[c] : 0.f;
        sum += v[i];
    }
    float mean = wave_sum(sum) / (float)width;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        int c = lane + 64 * i;
        float dlt = c < width ? v[i] - mean : 0.f;
        v[i] = dlt;
        ss += dlt * dlt;
    }
    float sigma = sqrtf(wave_sum(ss) / (float)(width - 1));
    float rinv = 1.0f / (sigma + eps);
    float* yr = y + row * ld;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        int c = lane + 64 * i;
        if (c < width) yr[c] = v[i] * rinv;
    }
    if (lane == 0) { stats[row * 2 + 0] = rinv; stats[row * 2 + 1] = sigma; }
}

// dx_i = rinv (dy_i - mean(dy)) - y_i * (sum_j dy_j y_j) / ((n-1) sigma)
template <int VPL>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* dy, const float* __restrict__ y,
                                                          const float* __restrict__ stats, long rows, int width,
                                                          int ld, float* dx) {
    const int lane = threadIdx.x & 63;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* gr = dy + row * ld;
    const float* yr = y + row * ld;
    float g[VPL], yy[VPL];
    float sg = 0.f, sgy = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        int c = lane + 64 * i;
        g[i] = c < width ? gr[c] : 0.f;
        yy[i] = c < width ? yr[c] : 0.f;
        sg += g[i];
        sgy += g[i] * yy[i];
    }
    sg = wave_sum(sg) / (float)width;
    sgy = wave_sum(sgy);
    float rinv = stats[row * 2 + 0], sigma = stats[row * 2 + 1];
    float coef = sigma > 0.f ? sgy / ((float)(width - 1) * sigma) : 0.f;
    float* dr = dx + row * ld;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        int c = lane + 64 * i;
        if (c < width) dr[c] = rinv * (g[i] - sg) - yy[i] * coef;
    }
}

// ------------------------------------------------------------------------------------------------
// K4 attention tail.  One wave per ray.  Lane j < k owns neighbour j's score / attention weight,
// lane k owns the background token; the d_model-long dot products are 16-byte-per-lane row reads
// reduced with wave shuffles; the C-wide weighted sum puts one output channel on each lane.
// Reference: models/attn.py:217-225 (+:54) and models/model.py:519-534.

// value of lane `src` (wave-uniform index) in every lane.  v_readlane ignores EXEC, so this is safe
// inside divergent code where a ds_bpermute shuffle would return 0 for masked-off source lanes.
__device__ __forceinline__ float bcast(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

__global__ __launch_bounds__(256) void tail_fwd_kernel(papr_tail_desc d, const float* __restrict__ kp,
                                                       const float* __restrict__ qp, const float* __restrict__ v,
                                                       const float* __restrict__ score_bias,
                                                       const float* __restrict__ influ, const int* __restrict__ idx,
                                                       long R, float* __restrict__ scores, float* __restrict__ attn,
                                                       float* __restrict__ fused) {
    const int lane = threadIdx.x & 63;
    long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int k = d.k;
    const float inv_sqrt_d = 1.0f / sqrtf((float)(d.scale_dim > 0 ? d.scale_dim : d.d_model));
    const float* q = qp + r * d.ld_qp;
    float my_score = 0.f;
    if (d.precomputed_dots) {
        if (lane < k) my_score = kp[r * k + lane];
    } else {
    // TG key rows per round: their 1 KB loads are in flight together (one row at a time is a chain of load latencies;
    // 1 row: 202 us, 4: 178 us, 10: 175 us per 25,600 x 20 x 256 launch)
    constexpr int TG = 4;
    for (int j0 = 0; j0 < k; j0 += TG) {
        float part[TG];
#pragma unroll
        for (int u = 0; u < TG; ++u) part[u] = 0.f;
        for (int c = lane * 4; c < d.d_model; c += 256) {
            const float4 a = *reinterpret_cast<const float4*>(q + c);
            float4 b[TG];
#pragma unroll
            for (int u = 0; u < TG; ++u) {
                const int j = j0 + u < k ? j0 + u : k - 1;
                b[u] = *reinterpret_cast<const float4*>(kp + (r * k + j) * d.ld_kp + c);
            }
#pragma unroll
            for (int u = 0; u < TG; ++u) part[u] += (a.x * b[u].x + a.y * b[u].y) + (a.z * b[u].z + a.w * b[u].w);
        }
#pragma unroll
        for (int u = 0; u < TG; ++u) {
            const float dot = wave_sum(part[u]);
            if (lane == j0 + u) my_score = dot;          // (j0 + u >= k repeats row k-1 into a lane that is masked below)
        }
    }
    }
    if (score_bias) my_score += score_bias[r];
    float z = -INFINITY;
    if (lane < k) {
        float sc = papr_act(my_score * inv_sqrt_d, d.score_act);
        scores[r * k + lane] = sc;
        z = sc * influ[idx[r * k + lane]];
    } else if (lane == k) {
        z = d.bkg_score;
    }
    float zmax = wave_max(z);
    float e = lane <= k ? expf(z - zmax) : 0.f;
    float denom = wave_sum(e);
    float a = e / denom;
    if (lane <= k) attn[r * (k + 1) + lane] = a;
    float top = lane < k ? a : 0.f;
    if (d.normalize) top = top / wave_sum(top);
    const int g4 = d.C >> 2;                         // 16-byte pieces per value row
    if ((d.C & 3) == 0 && (d.ld_v & 3) == 0 && g4 >= 1 && g4 <= 32 && (64 % g4) == 0) {
        // 64 / g4 value rows per instruction (a 32-channel row is 128 bytes: one row at a time left half the wave idle and was a chain of k round
        // trips), every lane's rows summed in ascending order, then the row groups met by a butterfly: a fixed order
        const int rp = 64 / g4, c4 = lane % g4, jl = lane / g4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int j0 = 0; j0 < k; j0 += rp) {
            const int j = j0 + jl, jc = j < k ? j : k - 1;
            const float ts = __shfl(top, jc, 64);   // (every lane takes part: the lanes a cross-lane read takes from must be active)
            const float tj = j < k ? ts : 0.f;
            const float4 vv = *reinterpret_cast<const float4*>(v + (r * k + jc) * d.ld_v + 4 * c4);
            acc.x += tj * vv.x; acc.y += tj * vv.y; acc.z += tj * vv.z; acc.w += tj * vv.w;
        }
        for (int off = g4; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
            acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
        }
        if (jl == 0) *reinterpret_cast<float4*>(fused + r * d.C + 4 * c4) = acc;
        return;
    }
    for (int c = lane; c < d.C; c += 64) {
        float acc = 0.f;
        for (int j = 0; j < k; ++j) acc += bcast(top, j) * v[(r * k + j) * d.ld_v + c];
        fused[r * d.C + c] = acc;
    }
}

__global__ __launch_bounds__(256) void tail_bwd_kernel(papr_tail_desc d, const float* __restrict__ kp,
                                                       const float* __restrict__ qp, const float* __restrict__ v,
                                                       const float* __restrict__ influ, const int* __restrict__ idx,
                                                       long R, const float* __restrict__ scores,
                                                       const float* __restrict__ attn, const float* __restrict__ d_fused,
                                                       const float* __restrict__ d_attn, float* __restrict__ d_kp,
                                                       float* __restrict__ d_qp, float* __restrict__ d_v,
                                                       float* __restrict__ d_influ, float* __restrict__ d_score_bias,
                                                       float* __restrict__ d_pair_influ, const float* __restrict__ kp_stats,
                                                       const float* __restrict__ score_bias, const float* __restrict__ kp_mean,
                                                       papr_f16_rows kp16, papr_f16_rows v16) {
    const int lane = threadIdx.x & 63;
    long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int k = d.k;
    // a gradient row in the one-product data-gradient run's input format (papr_f16_rows; chain4.hip's staging arithmetic: the row times the power of two of
    // its maximum, one rounding to f16): four values of this lane, the row's maximum `mx` known to every lane that holds a piece of it
    auto put16 = [&](const papr_f16_rows& o, long row, int col, const float4& val, float mx, bool first) {
        float inv;
        if (o.lo) {                                     // the parity runs' split form (gemm.hip: split_rows_kernel)
            const float sc = h3_scale_from_max(__float_as_uint(mx), inv);
            half4 hi, lo;
            split4(val, sc, hi, lo);
            *reinterpret_cast<half4*>(o.hi + row * o.ld + col) = hi;
            *reinterpret_cast<half4*>(o.lo + row * o.ld + col) = lo;
            if (first) { o.inv[row] = inv; o.scale[row] = sc; o.max[row] = mx; }
            return;
        }
        const float sc = one_scale_from_max(__float_as_uint(mx), ONE_EMIN_DGRAD, inv);
        unsigned h01, h23;                              // (the staging code's own instructions: chain4.hip, write_planes)
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h01) : "v"(val.x), "v"(sc));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h23) : "v"(val.z), "v"(sc));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(val.y), "v"(sc));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(val.w), "v"(sc));
        *reinterpret_cast<uint2*>(o.hi + row * o.ld + col) = make_uint2(h01, h23);
        if (first) { o.inv[row] = inv; o.scale[row] = sc; o.max[row] = mx; }
    };
    const float inv_sqrt_d = 1.0f / sqrtf((float)(d.scale_dim > 0 ? d.scale_dim : d.d_model));
    float a = lane <= k ? attn[r * (k + 1) + lane] : 0.f;
    float top = lane < k ? a : 0.f;
    float tsum = 1.f;
    if (d.normalize) { tsum = wave_sum(top); top = top / tsum; }
    // d_top_j = d_fused . v_j ;  d_v_j = top_j d_fused     (channels across lanes)
    float my_dtop = 0.f;
    const int g4 = d.C >> 2;
    if ((d.C & 3) == 0 && d.ld_v == d.C && g4 >= 1 && g4 <= 32 && (64 % g4) == 0) {
        // (as in tail_fwd_kernel: 64 / g4 rows per instruction; a row's dot product meets over its g4 neighbouring lanes)
        const int rp = 64 / g4, c4 = lane % g4, jl = lane / g4;
        const float4 gf = *reinterpret_cast<const float4*>(d_fused + r * d.C + 4 * c4);
#pragma unroll 4
        for (int j0 = 0; j0 < k; j0 += rp) {
            const int j = j0 + jl, jc = j < k ? j : k - 1;
            const float tj = __shfl(top, jc, 64);
            const float4 vv = *reinterpret_cast<const float4*>(v + (r * k + jc) * d.ld_v + 4 * c4);
            float part = (gf.x * vv.x + gf.y * vv.y) + (gf.z * vv.z + gf.w * vv.w);
            const float4 dv = make_float4(tj * gf.x, tj * gf.y, tj * gf.z, tj * gf.w);
            if (v16.hi) {                                       // (the row's g4 lanes meet for its maximum)
                float mx = fmaxf(fmaxf(fabsf(dv.x), fabsf(dv.y)), fmaxf(fabsf(dv.z), fabsf(dv.w)));
                for (int off = 1; off < g4; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
                if (j < k) put16(v16, r * k + j, 4 * c4, dv, mx, c4 == 0);
            } else if (j < k) *reinterpret_cast<float4*>(d_v + (r * k + j) * d.ld_v + 4 * c4) = dv;
            for (int off = 1; off < g4; off <<= 1) part += __shfl_xor(part, off, 64);
            const float got = __shfl(part, (lane % rp) * g4, 64);        // lane L = j0 + jl' wants the row of group jl' = L % rp
            if (lane >= j0 && lane < j0 + rp && lane < k) my_dtop = got;
        }
    } else
    for (int j = 0; j < k; ++j) {
        float part = 0.f;
        float tj = bcast(top, j);
        for (int c = lane; c < d.ld_v; c += 64) {   // padding columns of d_v are written as zero
            float gf = c < d.C ? d_fused[r * d.C + c] : 0.f;
            float vv = c < d.C ? v[(r * k + j) * d.ld_v + c] : 0.f;
            part += gf * vv;
            d_v[(r * k + j) * d.ld_v + c] = tj * gf;
        }
        float dt = wave_sum(part);
        if (lane == j) my_dtop = dt;
    }
    float da = (d_attn && lane <= k) ? d_attn[r * (k + 1) + lane] : 0.f;
    if (d.normalize) {
        float corr = wave_sum(my_dtop * top);  // lanes >= k contribute 0
        if (lane < k) da += (my_dtop - corr) / tsum;
    } else if (lane < k) {
        da += my_dtop;
    }
    float dz = a * (da - wave_sum(da * a));  // softmax backward over the k+1 tokens
    float sc = lane < k ? scores[r * k + lane] : 0.f;
    int pi = lane < k ? idx[r * k + lane] : 0;
    float w = lane < k ? influ[pi] : 0.f;
    float ddot = 0.f;
    if (lane < k) {
        if (d_pair_influ) d_pair_influ[r * k + lane] = dz * sc;   // summed per point by papr_segment_reduce
        else unsafeAtomicAdd(d_influ + pi, dz * sc);
        ddot = dz * w * papr_act_grad(sc, d.score_act) * inv_sqrt_d;
    }
    if (d_score_bias) {
        float sb = wave_sum(ddot);
        if (lane == 0) d_score_bias[r] = sb;
    }
    // d_qp = sum_j ddot_j kp_j ;  d_kp_j = ddot_j qp
    const float* q = qp + r * d.ld_qp;
    // kp rows standardised (y = (x - mean) / (sigma + eps)): hand back the gradient w.r.t. x (rownorm_bwd_kernel's
    // formula).  dy_j = ddot_j qp, so mean(dy_j) = ddot_j mean(qp) and sum_c dy_jc y_jc = ddot_j (qp . y_j), which is
    // the score before bias, scaling and activation: where the activation's derivative is not zero it is recovered
    // from the saved score (and where it is zero, ddot_j is).
    //   dx_jc = ddot_j rinv_j (qp_c - mean(qp)) - y_jc ddot_j (qp . y_j) / ((n - 1) sigma_j)
    float ca = ddot, cb = 0.f, qmean = 0.f;
    // kp_mean: the kp rows are RAW (papr_row_norm.raw_mean): y = (x - mean) * rinv per element as it is read -- the subtraction and the product the
    // fused run would have applied before storing the row, so every value below is the one a standardised row would have held
    float kmean = 0.f, krinv = 1.f;
    if (kp_mean && lane < k) { kmean = kp_mean[r * k + lane]; krinv = kp_stats[(r * k + lane) * 2]; }
    if (kp_stats) {
        float qs = 0.f;
        for (int c = lane * 4; c < d.d_model; c += 256) {
            const float4 qv = *reinterpret_cast<const float4*>(q + c);
            qs += (qv.x + qv.y) + (qv.z + qv.w);
        }
        qmean = wave_sum(qs) / (float)d.d_model;
        if (lane < k) {
            const float pre = (d.score_act == PAPR_ACT_LEAKY_RELU && sc < 0.f) ? sc * 5.0f : sc;      // activation undone
            const float dot = pre / inv_sqrt_d - (score_bias ? score_bias[r] : 0.f);
            const float rinv = kp_stats[(r * k + lane) * 2], sigma = kp_stats[(r * k + lane) * 2 + 1];
            ca = ddot * rinv;
            cb = (sigma > 0.f && ddot != 0.f) ? ddot * dot / ((float)(d.d_model - 1) * sigma) : 0.f;
        }
    }
    for (int c = lane * 4; c < d.d_model; c += 256) {
        float4 qv = *reinterpret_cast<const float4*>(q + c);
        const float4 qc = make_float4(qv.x - qmean, qv.y - qmean, qv.z - qmean, qv.w - qmean);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < k; ++j) {               // (requesting the rows of a round ahead of its stores was slower: 335 vs 229 us)
            float gj = bcast(ddot, j);
            float4 kv = *reinterpret_cast<const float4*>(kp + (r * k + j) * d.ld_kp + c);
            if (kp_mean) {
                const float mj = bcast(kmean, j), rj = bcast(krinv, j);
                kv = make_float4((kv.x - mj) * rj, (kv.y - mj) * rj, (kv.z - mj) * rj, (kv.w - mj) * rj);
            }
            acc.x += gj * kv.x; acc.y += gj * kv.y; acc.z += gj * kv.z; acc.w += gj * kv.w;
            float4 o;
            if (kp_stats) {
                const float aj = bcast(ca, j), bj = bcast(cb, j);
                o = make_float4(aj * qc.x - bj * kv.x, aj * qc.y - bj * kv.y, aj * qc.z - bj * kv.z, aj * qc.w - bj * kv.w);
            } else {
                o = make_float4(gj * qv.x, gj * qv.y, gj * qv.z, gj * qv.w);
            }
            if (kp16.hi) {                                      // (d_model = 256: the wave holds the whole row)
                const float mx = wave_max(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
                put16(kp16, r * k + j, c, o, mx, lane == 0);
            } else *reinterpret_cast<float4*>(d_kp + (r * k + j) * d.ld_kp + c) = o;
        }
        *reinterpret_cast<float4*>(d_qp + r * d.ld_qp + c) = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// The attention tail for 64 <= k <= 255 neighbours per ray (round 5; the reference takes any k, models/model.py:281, 519-534): token t = 64 s + lane
// lives in slot s of its lane (NS slots), tokens 0 .. k-1 the neighbours, token k the background.  The formulas, the order of every per-ray sum over
// the tokens excepted, are tail_fwd_kernel's / tail_bwd_kernel's; the kernels above stay what every shipped configuration (k = 20, 30) runs.
template <int NS>
struct Tok {
    float v[NS];
    __device__ __forceinline__ float at(int t) const {          // token t's value in every lane (t wave-uniform)
        float r = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) if ((t >> 6) == s) r = bcast(v[s], t & 63);
        return r;
    }
    __device__ __forceinline__ float sum() const { float a = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) a += v[s];
        return wave_sum(a); }
    __device__ __forceinline__ float max() const { float a = v[0];
#pragma unroll
        for (int s = 1; s < NS; ++s) a = fmaxf(a, v[s]);
        return wave_max(a); }
};

template <int NS>
__global__ __launch_bounds__(256) void tail_fwd_wide_kernel(papr_tail_desc d, const float* __restrict__ kp, const float* __restrict__ qp,
                                                            const float* __restrict__ v, const float* __restrict__ score_bias,
                                                            const float* __restrict__ influ, const int* __restrict__ idx, long R,
                                                            float* __restrict__ scores, float* __restrict__ attn, float* __restrict__ fused) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int k = d.k;
    const float inv_sqrt_d = 1.0f / sqrtf((float)(d.scale_dim > 0 ? d.scale_dim : d.d_model));
    Tok<NS> sc, z, a, top;
#pragma unroll
    for (int s = 0; s < NS; ++s) sc.v[s] = 0.f;
    if (d.precomputed_dots) {
#pragma unroll
        for (int s = 0; s < NS; ++s) if (64 * s + lane < k) sc.v[s] = kp[r * k + 64 * s + lane];
    } else {
        const float* q = qp + r * d.ld_qp;
        for (int j = 0; j < k; ++j) {
            float part = 0.f;
            for (int c = lane * 4; c < d.d_model; c += 256) {
                const float4 x = *reinterpret_cast<const float4*>(q + c), y = *reinterpret_cast<const float4*>(kp + (r * k + j) * d.ld_kp + c);
                part += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
            }
            const float dot = wave_sum(part);
#pragma unroll
            for (int s = 0; s < NS; ++s) if (64 * s + lane == j) sc.v[s] = dot;
        }
    }
    const float sb = score_bias ? score_bias[r] : 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int t = 64 * s + lane;
        z.v[s] = -INFINITY;
        if (t < k) {
            const float x = papr_act((sc.v[s] + sb) * inv_sqrt_d, d.score_act);
            scores[r * k + t] = x;
            z.v[s] = x * influ[idx[r * k + t]];
        } else if (t == k) {
            z.v[s] = d.bkg_score;
        }
    }
    const float zmax = z.max();
    Tok<NS> e;
#pragma unroll
    for (int s = 0; s < NS; ++s) e.v[s] = 64 * s + lane <= k ? expf(z.v[s] - zmax) : 0.f;
    const float denom = e.sum();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int t = 64 * s + lane;
        a.v[s] = e.v[s] / denom;
        if (t <= k) attn[r * (k + 1) + t] = a.v[s];
        top.v[s] = t < k ? a.v[s] : 0.f;
    }
    if (d.normalize) {
        const float ts = top.sum();
#pragma unroll
        for (int s = 0; s < NS; ++s) top.v[s] = top.v[s] / ts;
    }
    for (int c = lane; c < d.C; c += 64) {
        float acc = 0.f;
        for (int j = 0; j < k; ++j) acc += top.at(j) * v[(r * k + j) * d.ld_v + c];
        fused[r * d.C + c] = acc;
    }
}

template <int NS>
__global__ __launch_bounds__(256) void tail_bwd_wide_kernel(papr_tail_desc d, const float* __restrict__ kp, const float* __restrict__ qp,
                                                            const float* __restrict__ v, const float* __restrict__ influ, const int* __restrict__ idx,
                                                            long R, const float* __restrict__ scores, const float* __restrict__ attn,
                                                            const float* __restrict__ d_fused, const float* __restrict__ d_attn,
                                                            float* __restrict__ d_kp, float* __restrict__ d_qp, float* __restrict__ d_v,
                                                            float* __restrict__ d_influ, float* __restrict__ d_score_bias,
                                                            float* __restrict__ d_pair_influ, const float* __restrict__ kp_stats,
                                                            const float* __restrict__ score_bias, const float* __restrict__ kp_mean) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int k = d.k;
    const float inv_sqrt_d = 1.0f / sqrtf((float)(d.scale_dim > 0 ? d.scale_dim : d.d_model));
    Tok<NS> a, top, dtop, da, ddot, ca, cb, kmean, krinv;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int t = 64 * s + lane;
        a.v[s] = t <= k ? attn[r * (k + 1) + t] : 0.f;
        top.v[s] = t < k ? a.v[s] : 0.f;
        dtop.v[s] = 0.f; ddot.v[s] = 0.f; cb.v[s] = 0.f; kmean.v[s] = 0.f; krinv.v[s] = 1.f;
    }
    float tsum = 1.f;
    if (d.normalize) {
        tsum = top.sum();
#pragma unroll
        for (int s = 0; s < NS; ++s) top.v[s] = top.v[s] / tsum;
    }
    for (int j = 0; j < k; ++j) {
        float part = 0.f;
        const float tj = top.at(j);
        for (int c = lane; c < d.ld_v; c += 64) {
            const float gf = c < d.C ? d_fused[r * d.C + c] : 0.f;
            const float vv = c < d.C ? v[(r * k + j) * d.ld_v + c] : 0.f;
            part += gf * vv;
            d_v[(r * k + j) * d.ld_v + c] = tj * gf;
        }
        const float dt = wave_sum(part);
#pragma unroll
        for (int s = 0; s < NS; ++s) if (64 * s + lane == j) dtop.v[s] = dt;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) da.v[s] = (d_attn && 64 * s + lane <= k) ? d_attn[r * (k + 1) + 64 * s + lane] : 0.f;
    if (d.normalize) {
        Tok<NS> w;
#pragma unroll
        for (int s = 0; s < NS; ++s) w.v[s] = dtop.v[s] * top.v[s];
        const float corr = w.sum();
#pragma unroll
        for (int s = 0; s < NS; ++s) if (64 * s + lane < k) da.v[s] += (dtop.v[s] - corr) / tsum;
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) if (64 * s + lane < k) da.v[s] += dtop.v[s];
    }
    Tok<NS> w2;
#pragma unroll
    for (int s = 0; s < NS; ++s) w2.v[s] = da.v[s] * a.v[s];
    const float dasum = w2.sum();
    Tok<NS> scv;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int t = 64 * s + lane;
        const float dz = a.v[s] * (da.v[s] - dasum);
        scv.v[s] = t < k ? scores[r * k + t] : 0.f;
        if (t < k) {
            const int pi = idx[r * k + t];
            if (d_pair_influ) d_pair_influ[r * k + t] = dz * scv.v[s];
            else unsafeAtomicAdd(d_influ + pi, dz * scv.v[s]);
            ddot.v[s] = dz * influ[pi] * papr_act_grad(scv.v[s], d.score_act) * inv_sqrt_d;
        }
        ca.v[s] = ddot.v[s];
    }
    if (d_score_bias) {
        const float sbs = ddot.sum();
        if (lane == 0) d_score_bias[r] = sbs;
    }
    const float* q = qp + r * d.ld_qp;
    float qmean = 0.f;
    if (kp_stats) {
        float qs = 0.f;
        for (int c = lane * 4; c < d.d_model; c += 256) {
            const float4 qv = *reinterpret_cast<const float4*>(q + c);
            qs += (qv.x + qv.y) + (qv.z + qv.w);
        }
        qmean = wave_sum(qs) / (float)d.d_model;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int t = 64 * s + lane;
            if (t < k) {
                const float pre = (d.score_act == PAPR_ACT_LEAKY_RELU && scv.v[s] < 0.f) ? scv.v[s] * 5.0f : scv.v[s];
                const float dot = pre / inv_sqrt_d - (score_bias ? score_bias[r] : 0.f);
                const float rinv = kp_stats[(r * k + t) * 2], sigma = kp_stats[(r * k + t) * 2 + 1];
                ca.v[s] = ddot.v[s] * rinv;
                cb.v[s] = (sigma > 0.f && ddot.v[s] != 0.f) ? ddot.v[s] * dot / ((float)(d.d_model - 1) * sigma) : 0.f;
                if (kp_mean) { kmean.v[s] = kp_mean[r * k + t]; krinv.v[s] = rinv; }
            }
        }
    }
    for (int c = lane * 4; c < d.d_model; c += 256) {
        const float4 qv = *reinterpret_cast<const float4*>(q + c);
        const float4 qc = make_float4(qv.x - qmean, qv.y - qmean, qv.z - qmean, qv.w - qmean);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < k; ++j) {
            const float gj = ddot.at(j);
            float4 kv = *reinterpret_cast<const float4*>(kp + (r * k + j) * d.ld_kp + c);
            if (kp_mean) {
                const float mj = kmean.at(j), rj = krinv.at(j);
                kv = make_float4((kv.x - mj) * rj, (kv.y - mj) * rj, (kv.z - mj) * rj, (kv.w - mj) * rj);
            }
            acc.x += gj * kv.x; acc.y += gj * kv.y; acc.z += gj * kv.z; acc.w += gj * kv.w;
            float4 o;
            if (kp_stats) {
                const float aj = ca.at(j), bj = cb.at(j);
                o = make_float4(aj * qc.x - bj * kv.x, aj * qc.y - bj * kv.y, aj * qc.z - bj * kv.z, aj * qc.w - bj * kv.w);
            } else {
                o = make_float4(gj * qv.x, gj * qv.y, gj * qv.z, gj * qv.w);
            }
            *reinterpret_cast<float4*>(d_kp + (r * k + j) * d.ld_kp + c) = o;
        }
        *reinterpret_cast<float4*>(d_qp + r * d.ld_qp + c) = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm affine folded into the following Linear layer (see papr_hip.h).  One wave per output row.
__device__ __forceinline__ void ln_fold_fwd_block(const float* __restrict__ w, int n_out, int n_in, int ldw,
                                                  const float* __restrict__ c, const float* __restrict__ a2,
                                                  const float* __restrict__ b2, float* __restrict__ eff_w, int ld_eff,
                                                  float* __restrict__ eff_b) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= n_out) return;
    float acc = 0.f;
    for (int i = lane; i < ld_eff; i += 64) {
        const float wv = i < n_in ? w[(long)o * ldw + i] : 0.f;
        eff_w[(long)o * ld_eff + i] = i < n_in ? wv * a2[i] : 0.f;
        acc += i < n_in ? wv * b2[i] : 0.f;
    }
    acc = wave_sum(acc);
    if (lane == 0) eff_b[o] = (c ? c[o] : 0.f) + acc;
}
__global__ __launch_bounds__(256) void ln_fold_fwd_kernel(const float* __restrict__ w, int n_out, int n_in, int ldw,
                                                          const float* __restrict__ c, const float* __restrict__ a2,
                                                          const float* __restrict__ b2, float* __restrict__ eff_w, int ld_eff,
                                                          float* __restrict__ eff_b) {
    ln_fold_fwd_block(w, n_out, n_in, ldw, c, a2, b2, eff_w, ld_eff, eff_b);
}
// all folds of a model in one launch (blockIdx.y = job): a PAPR step folds four LayerNorm affines -- in front of the key and the query MLP, behind
// them into w_k and w_q -- 5 us of launch each for a microsecond of work
struct LnFoldJobs { papr_ln_fold_job job[PAPR_LN_FOLD_MAX_JOBS]; };
__global__ __launch_bounds__(256) void ln_fold_fwd_batch_kernel(LnFoldJobs t) {
    const papr_ln_fold_job& j = t.job[blockIdx.y];
    if ((int)blockIdx.x * 4 >= j.n_out) return;
    ln_fold_fwd_block(j.w, j.n_out, j.n_in, j.ldw, j.c, j.a2, j.b2, j.eff_w, j.ld_eff, j.eff_b);
}

// a workgroup of sixteen waves owns 64 columns over all rows: wave q takes the rows q, q + 16, ...; the sixteen partial column
// sums meet in LDS in a fixed order (the matrices are tiny -- 256 x 128 -- and the launch is latency: four waves took 22 us)
__device__ __forceinline__ void ln_fold_bwd_block(const float* __restrict__ w, int n_out, int n_in, int ldw,
                                                  const float* __restrict__ a2, const float* __restrict__ b2,
                                                  const float* __restrict__ d_eff_w, int ld_eff,
                                                  const float* __restrict__ d_eff_b, float* __restrict__ d_w,
                                                  float* __restrict__ d_a2, float* __restrict__ d_b2) {
    __shared__ float part[2][16][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const bool ok = i < n_in;
    const float av = ok ? a2[i] : 0.f, bv = ok ? b2[i] : 0.f;
    float sa = 0.f, sb = 0.f;
    for (int o0 = q; o0 < n_out; o0 += 128) {           // eight rows in flight per wave (one at a time: 64 dependent round trips, 46 us)
        float de[8], wv[8], db[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + 16 * u;
            const bool live = ok && o < n_out;
            de[u] = live ? d_eff_w[(long)o * ld_eff + i] : 0.f;
            wv[u] = live ? w[(long)o * ldw + i] : 0.f;
            db[u] = o < n_out ? d_eff_b[o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + 16 * u;
            if (ok && o < n_out) d_w[(long)o * ldw + i] = de[u] * av + db[u] * bv;
            sa += de[u] * wv[u];
            sb += db[u] * wv[u];
        }
    }
    part[0][q][lane] = sa;
    part[1][q][lane] = sb;
    __syncthreads();
    if (q == 0 && ok) {
        float ta = part[0][0][lane], tb = part[1][0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) { ta += part[0][w][lane]; tb += part[1][w][lane]; }
        d_a2[i] = ta;
        d_b2[i] = tb;
    }
}
__global__ __launch_bounds__(1024) void ln_fold_bwd_kernel(const float* __restrict__ w, int n_out, int n_in, int ldw,
                                                          const float* __restrict__ a2, const float* __restrict__ b2,
                                                          const float* __restrict__ d_eff_w, int ld_eff,
                                                          const float* __restrict__ d_eff_b, float* __restrict__ d_w,
                                                          float* __restrict__ d_a2, float* __restrict__ d_b2) {
    ln_fold_bwd_block(w, n_out, n_in, ldw, a2, b2, d_eff_w, ld_eff, d_eff_b, d_w, d_a2, d_b2);
}
__global__ __launch_bounds__(1024) void ln_fold_bwd_batch_kernel(LnFoldJobs t) {
    const papr_ln_fold_job& j = t.job[blockIdx.y];
    if ((int)blockIdx.x * 64 >= j.n_in) return;       // (uniform per workgroup: before the block's barrier)
    ln_fold_bwd_block(j.w, j.n_out, j.n_in, j.ldw, j.a2, j.b2, j.d_eff_w, j.ld_eff, j.d_eff_b, j.d_w, j.d_a2, j.d_b2);
}

}  // namespace

static int ln_fold_check(const char* who, const papr_ln_fold_job* jobs, int n, bool bwd) {
    PAPR_REQUIRE(jobs && n >= 1 && n <= PAPR_LN_FOLD_MAX_JOBS, "%s: %d jobs (1 .. %d)", who, n, PAPR_LN_FOLD_MAX_JOBS);
    for (int i = 0; i < n; ++i) {
        const papr_ln_fold_job& j = jobs[i];
        PAPR_REQUIRE(j.w && j.a2 && j.b2, "%s: job %d: null pointer", who, i);
        PAPR_REQUIRE(bwd ? (j.d_eff_w && j.d_eff_b && j.d_w && j.d_a2 && j.d_b2) : (j.eff_w && j.eff_b), "%s: job %d: null %s pointer", who, i, bwd ? "gradient" : "output");
        PAPR_REQUIRE(j.n_out >= 1 && j.n_in >= 1 && j.ldw >= j.n_in && j.ld_eff >= j.n_in && j.ld_eff <= 1024, "%s: job %d: n_out %d, n_in %d, ldw %d, ld_eff %d", who, i,
                     j.n_out, j.n_in, j.ldw, j.ld_eff);
    }
    return 0;
}

extern "C" int papr_ln_fold_fwd_batch(const papr_ln_fold_job* jobs, int32_t n, papr_stream_t stream) {
    if (int rc = ln_fold_check("papr_ln_fold_fwd_batch", jobs, n, false)) return rc;
    LnFoldJobs t;
    int most = 0;
    for (int i = 0; i < n; ++i) { t.job[i] = jobs[i]; most = jobs[i].n_out > most ? jobs[i].n_out : most; }
    ln_fold_fwd_batch_kernel<<<dim3((unsigned)((most + 3) / 4), (unsigned)n), dim3(256), 0, as_stream(stream)>>>(t);
    PAPR_CHECK_LAUNCH("ln_fold_fwd_batch");
    return 0;
}

extern "C" int papr_ln_fold_bwd_batch(const papr_ln_fold_job* jobs, int32_t n, papr_stream_t stream) {
    if (int rc = ln_fold_check("papr_ln_fold_bwd_batch", jobs, n, true)) return rc;
    LnFoldJobs t;
    int most = 0;
    for (int i = 0; i < n; ++i) { t.job[i] = jobs[i]; most = jobs[i].n_in > most ? jobs[i].n_in : most; }
    ln_fold_bwd_batch_kernel<<<dim3((unsigned)((most + 63) / 64), (unsigned)n), dim3(1024), 0, as_stream(stream)>>>(t);
    PAPR_CHECK_LAUNCH("ln_fold_bwd_batch");
    return 0;
}

extern "C" int papr_ln_fold_fwd(const float* w, int32_t n_out, int32_t n_in, int32_t ldw, const float* c, const float* a2,
                                const float* b2, float* eff_w, int32_t ld_eff, float* eff_b, papr_stream_t stream) {
    PAPR_REQUIRE(w && a2 && b2 && eff_w && eff_b, "papr_ln_fold_fwd: null pointer");
    PAPR_REQUIRE(n_out >= 1 && n_in >= 1 && ldw >= n_in && ld_eff >= n_in && ld_eff <= 1024, "papr_ln_fold_fwd: n_out %d, n_in %d, ldw %d, ld_eff %d", n_out, n_in, ldw, ld_eff);
    ln_fold_fwd_kernel<<<dim3((unsigned)((n_out + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(w, n_out, n_in, ldw, c, a2, b2, eff_w, ld_eff, eff_b);
    PAPR_CHECK_LAUNCH("ln_fold_fwd");
    return 0;
}

extern "C" int papr_ln_fold_bwd(const float* w, int32_t n_out, int32_t n_in, int32_t ldw, const float* a2, const float* b2,
                                const float* d_eff_w, int32_t ld_eff, const float* d_eff_b, float* d_w, float* d_a2, float* d_b2,
                                papr_stream_t stream) {
    PAPR_REQUIRE(w && a2 && b2 && d_eff_w && d_eff_b && d_w && d_a2 && d_b2, "papr_ln_fold_bwd: null pointer");
    PAPR_REQUIRE(n_out >= 1 && n_in >= 1 && ldw >= n_in && ld_eff >= n_in && ld_eff <= 1024, "papr_ln_fold_bwd: n_out %d, n_in %d, ldw %d, ld_eff %d", n_out, n_in, ldw, ld_eff);
    ln_fold_bwd_kernel<<<dim3((unsigned)((n_in + 63) / 64)), dim3(1024), 0, as_stream(stream)>>>(w, n_out, n_in, ldw, a2, b2, d_eff_w, ld_eff, d_eff_b, d_w, d_a2, d_b2);
    PAPR_CHECK_LAUNCH("ln_fold_bwd");
    return 0;
}

extern "C" int papr_rownorm_fwd(const float* x, int64_t rows, int width, int ld, float eps, float* y, float* stats,
                                papr_stream_t stream) {
    PAPR_REQUIRE(x && y && stats, "papr_rownorm_fwd: null pointer");
    PAPR_REQUIRE(width >= 2 && width <= 64 * MAX_VPL && ld >= width, "papr_rownorm_fwd: width %d / ld %d unsupported", width, ld);
    if (rows <= 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (width <= 256) rownorm_fwd_kernel<4><<<grid, block, 0, as_stream(stream)>>>(x, rows, width, ld, eps, y, stats);
    else rownorm_fwd_kernel<MAX_VPL><<<grid, block, 0, as_stream(stream)>>>(x, rows, width, ld, eps, y, stats);
    PAPR_CHECK_LAUNCH("rownorm_fwd");
    return 0;
}

// the statistics of rownorm_fwd_kernel alone (x untouched): stats[2m], [2m+1] = 1 / (std + eps), std; mean[m] -- for a fused run that applies them while
// it stages the rows (ChainArgs::in_norm_mean), when the caller of papr_mlp_fwd has none to give
template <int VPL>
__global__ __launch_bounds__(256) void rownorm_stats_kernel(const float* __restrict__ x, long rows, int width, int ld, float eps,
                                                            float* __restrict__ stats, float* __restrict__ mean_out) {
    const int lane = threadIdx.x & 63;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ld;
    float v[VPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        int c = lane + 64 * i;
        v[i] = c < width ? xr[c] : 0.f;
        sum += v[i];
    }
    float mean = wave_sum(sum) / (float)width;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        int c = lane + 64 * i;
        float dlt = c < width ? v[i] - mean : 0.f;
        ss += dlt * dlt;
    }
    float sigma = sqrtf(wave_sum(ss) / (float)(width - 1));
    if (lane == 0) { stats[row * 2 + 0] = 1.0f / (sigma + eps); stats[row * 2 + 1] = sigma; mean_out[row] = mean; }
}
int papr_rownorm_stats(const float* x, int64_t rows, int width, int ld, float eps, float* stats, float* mean, papr_stream_t stream) {
    PAPR_REQUIRE(width >= 2 && width <= 64 * MAX_VPL && ld >= width, "rownorm_stats: width %d / ld %d unsupported", width, ld);
    if (rows <= 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (width <= 256) rownorm_stats_kernel<4><<<grid, block, 0, as_stream(stream)>>>(x, rows, width, ld, eps, stats, mean);
    else rownorm_stats_kernel<MAX_VPL><<<grid, block, 0, as_stream(stream)>>>(x, rows, width, ld, eps, stats, mean);
    PAPR_CHECK_LAUNCH("rownorm_stats");
    return 0;
}

// x <- (x - mean[m]) * stats[2m] over the first `width` columns of every row: the LayerNorm core with GIVEN statistics (papr_row_norm.given_mean), for
// the MLP whose first layers no fused run stages
__global__ __launch_bounds__(256) void rownorm_apply_kernel(float* __restrict__ x, long rows, int width, int ld, const float* __restrict__ stats,
                                                            const float* __restrict__ mean) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long m = e / width;
    if (m >= rows) return;
    const int c = (int)(e - m * width);
    x[m * ld + c] = (x[m * ld + c] - mean[m]) * stats[2 * m];
}
int papr_rownorm_apply(float* x, int64_t rows, int width, int ld, const float* stats, const float* mean, papr_stream_t stream) {
    if (rows <= 0) return 0;
    const long n = rows * width;
    rownorm_apply_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(x, rows, width, ld, stats, mean);
    PAPR_CHECK_LAUNCH("rownorm_apply");
    return 0;
}

// dots[m] = rows[m] . dot_rows[m / rows_per_dot]: one wave per row (the stand-alone form of what a fused run's last row phase does)
__global__ __launch_bounds__(256) void row_dots_kernel(const float* __restrict__ rows, long M, int width, int ld,
                                                       const float* __restrict__ dot_rows, int ld_dot, int rows_per_dot,
                                                       float* __restrict__ dots) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* a = rows + m * ld;
    const float* g = dot_rows + (m / rows_per_dot) * ld_dot;
    float part = 0.f;
    for (int c = lane; c < width; c += 64) part = fmaf(a[c], g[c], part);
    part = wave_sum(part);
    if (lane == 0) dots[m] = part;
}

extern "C" int papr_row_dots(const float* rows, int64_t M, int width, int ld, const float* dot_rows, int ld_dot,
                             int rows_per_dot, float* dots, papr_stream_t stream) {
    PAPR_REQUIRE(rows && dot_rows && dots, "papr_row_dots: null pointer");
    PAPR_REQUIRE(width >= 1 && ld >= width && ld_dot >= width && rows_per_dot >= 1, "papr_row_dots: width %d / ld %d / ld_dot %d / rows_per_dot %d", width, ld, ld_dot, rows_per_dot);
    if (M <= 0) return 0;
    row_dots_kernel<<<dim3((unsigned)((M + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(rows, M, width, ld, dot_rows, ld_dot, rows_per_dot, dots);
    PAPR_CHECK_LAUNCH("row_dots");
    return 0;
}

// The score bias c0 = q'.b_k, q' = W_q Q + b_q (models/attn.py:217-225 computes it inside the R k x d_model score product; here one number per
// ray) is linear in the query embedding Q itself: c0 = Q.(W_q^T b_k) + b_q.b_k.  With u = Q^T d_c0 (dq) and s = sum(d_c0) its gradient is
//   d_Q += d_c0 (x) (W_q^T b_k)      d_W_q += b_k (x) u      d_b_q += s b_k      d_b_k = W_q u + s b_q
// -- two launches (ten torch launches of 5 ... 26 us before, which also read the R x d_model rows q'): qk_bias_rows_kernel walks the R rows of Q
// once, every workgroup a chunk of rows, and updates d_Q on the way; qk_bias_finish_kernel adds the chunks in a fixed order and applies the rest.
namespace {
constexpr int QKB_CHUNKS = 256;                  // row chunks = workgroups of the first launch
// Thread = (row lane, float4 column group of [Q | 1]): the workgroup's rows r0 + lane, r0 + lane + RL, ... eight at a time (all loads of a batch
// in flight together: the kernel is a latency chain otherwise).  partial[chunk][dq + 1].
__global__ __launch_bounds__(256) void qk_bias_rows_kernel(const float* __restrict__ Q, int ldq, int dq, int dm, const float* __restrict__ d_c0, long R,
                                                           const float* __restrict__ wq, int ldwq, const float* __restrict__ bk, float* __restrict__ d_Q,
                                                           float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // vq = W_q^T b_k (4 gq, zero beyond dq), then the lanes' sums (256 float4)
    const int gq = (dq + 3) >> 2, ng = gq + 1;
    float* vq = sm;
    float4* lsum = reinterpret_cast<float4*>(sm + 4 * gq);
    {   // vq: thread = (n lane, column group); the n lanes meet in LDS in a fixed order
        const int nl_count = 256 / gq > 0 ? 256 / gq : 1, span = gq < 256 ? gq : 256;
        for (int cb = 0; cb < gq; cb += 256) {
            const int cg = cb + (int)threadIdx.x % span, nl = (int)threadIdx.x / span;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cg < gq && nl < nl_count) {
#pragma unroll 8
                for (int n = nl; n < dm; n += nl_count) {        // (whole float4 groups: ldwq is a multiple of 4; columns beyond dq are zeroed below)
                    const float b = bk[n];
                    const float4 w = *reinterpret_cast<const float4*>(wq + (long)n * ldwq + 4 * cg);
                    t.x = fmaf(w.x, b, t.x); t.y = fmaf(w.y, b, t.y); t.z = fmaf(w.z, b, t.z); t.w = fmaf(w.w, b, t.w);
                }
                if (4 * cg + 1 >= dq) t.y = 0.f;
                if (4 * cg + 2 >= dq) t.z = 0.f;
                if (4 * cg + 3 >= dq) t.w = 0.f;
            }
            lsum[threadIdx.x] = t;
            __syncthreads();
            if (nl == 0 && cg < gq) {
                float4 r = lsum[threadIdx.x];
                for (int j = 1; j < nl_count; ++j) { const float4 v = lsum[threadIdx.x + j * gq]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
                *reinterpret_cast<float4*>(vq + 4 * cg) = r;
            }
            __syncthreads();
        }
    }
    const long per = (R + gridDim.x - 1) / gridDim.x, r0 = (long)blockIdx.x * per, r1 = r0 + per < R ? r0 + per : R;
    for (int gb = 0; gb < ng; gb += 256) {
        const int span = ng - gb < 256 ? ng - gb : 256, RL = 256 / span;
        const int g = gb + (int)threadIdx.x % span, rl = (int)threadIdx.x / span;
        const bool ones = g >= gq;                                            // the last group: sum(d_c0)
        const int gs = ones ? 0 : g;                                          // (it loads group 0 like everybody else and ignores it: no branch around the loads)
        const float4 v = *reinterpret_cast<const float4*>(vq + 4 * gs);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < RL) {
            for (long rb = r0 + rl; rb < r1; rb += 8L * RL) {
                float gg[8];
                float4 x[8], dqv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long r = rb + (long)u * RL, rc = r < r1 ? r : r1 - 1;     // (clamped: every load of the batch is issued unconditionally)
                    gg[u] = d_c0[rc];
                    x[u] = *reinterpret_cast<const float4*>(Q + rc * ldq + 4 * gs);
                    dqv[u] = *reinterpret_cast<const float4*>(d_Q + rc * ldq + 4 * gs);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long r = rb + (long)u * RL;
                    const float gu = r < r1 ? gg[u] : 0.f;
                    const float4 xv = ones ? make_float4(1.f, 0.f, 0.f, 0.f) : x[u];
                    acc.x = fmaf(gu, xv.x, acc.x); acc.y = fmaf(gu, xv.y, acc.y); acc.z = fmaf(gu, xv.z, acc.z); acc.w = fmaf(gu, xv.w, acc.w);
                    if (!ones && r < r1) {
                        float4 d = dqv[u];
                        d.x = fmaf(gu, v.x, d.x); d.y = fmaf(gu, v.y, d.y); d.z = fmaf(gu, v.z, d.z); d.w = fmaf(gu, v.w, d.w);
                        *reinterpret_cast<float4*>(d_Q + r * ldq + 4 * g) = d;
                    }
                }
            }
        }
        lsum[threadIdx.x] = acc;
        __syncthreads();
        if (rl == 0) {
            float4 t = lsum[threadIdx.x];
            for (int j = 1; j < RL; ++j) { const float4 w = lsum[threadIdx.x + j * span]; t.x += w.x; t.y += w.y; t.z += w.z; t.w += w.w; }
            float* out = partial + (long)blockIdx.x * (dq + 1);
            if (ones) out[dq] = t.x;
            else {
                out[4 * g] = t.x;
                if (4 * g + 1 < dq) out[4 * g + 1] = t.y;
                if (4 * g + 2 < dq) out[4 * g + 2] = t.z;
                if (4 * g + 3 < dq) out[4 * g + 3] = t.w;
            }
        }
        __syncthreads();
    }
}
// Every workgroup adds the chunks for all dq + 1 columns (thread = (chunk lane, column): 1024 / columns lanes stride through the chunks, the lanes
// meet in LDS in lane order), then handles its share of the d_model rows: d_W_q += b_k (x) u, d_b_q += s b_k, d_b_k = W_q u + s b_q (a wave per row).
__global__ __launch_bounds__(1024) void qk_bias_finish_kernel(const float* __restrict__ partial, int chunks, int dq, int dm, const float* __restrict__ wq,
                                                              int ldwq, const float* __restrict__ bk, const float* __restrict__ bq, float* __restrict__ d_wq,
                                                              const float* __restrict__ d_bq_in, float* __restrict__ d_bq, float* __restrict__ d_bk) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];      // red [Q^T d_c0 (dq) | sum d_c0], then the lanes' sums (1024)
    const int cols = dq + 1;
    float* red = fsm;
    float* lane_sum = fsm + cols;
    for (int cb = 0; cb < cols; cb += 1024) {
        const int span = cols - cb < 1024 ? cols - cb : 1024, ZL = 1024 / span;
        const int c = cb + (int)threadIdx.x % span, zl = (int)threadIdx.x / span;
        float t = 0.f;
        if (zl < ZL)
#pragma unroll 8
            for (int z = zl; z < chunks; z += ZL) t += partial[(long)z * cols + c];
        lane_sum[threadIdx.x] = t;
        __syncthreads();
        if (zl == 0) {
            float r = lane_sum[threadIdx.x];
            for (int j = 1; j < ZL; ++j) r += lane_sum[threadIdx.x + j * span];
            red[c] = r;
        }
        __syncthreads();
    }
    const float ssum = red[dq];
    const int rows_per = (dm + gridDim.x - 1) / gridDim.x, n0 = blockIdx.x * rows_per, n1 = n0 + rows_per < dm ? n0 + rows_per : dm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = n0 + wave; n < n1; n += 16) {
        const float b = bk[n];
        float dot = 0.f;
        for (int c = lane; c < dq; c += 64) {
            const long o = (long)n * ldwq + c;
            dot = fmaf(wq[o], red[c], dot);
            d_wq[o] = fmaf(b, red[c], d_wq[o]);
        }
        dot = wave_sum(dot);
        if (lane == 0) {
            d_bk[n] = fmaf(ssum, bq[n], dot);
            d_bq[n] = fmaf(b, ssum, d_bq_in[n]);
        }
    }
}
}  // namespace

extern "C" size_t papr_qk_bias_bwd_workspace_bytes(int dq) { return (size_t)QKB_CHUNKS * (dq + 1) * sizeof(float); }

extern "C" int papr_qk_bias_bwd(const float* Q, int ldq, int dq, int dm, const float* d_c0, int64_t R, const float* wq, int ldwq, const float* bk, const float* bq,
                                float* d_Q, float* d_wq, const float* d_bq_in, float* d_bq, float* d_bk, void* workspace, papr_stream_t stream) {
    PAPR_REQUIRE(Q && d_c0 && wq && bk && bq && d_Q && d_wq && d_bq_in && d_bq && d_bk && workspace, "papr_qk_bias_bwd: null pointer");
    PAPR_REQUIRE(dq >= 1 && dm >= 1 && dq <= 1024 && dm <= 1024 && ldq >= dq && ldwq >= dq && R >= 1 && ldq % 4 == 0 && ldwq % 4 == 0,
                 "papr_qk_bias_bwd: dq %d (ld %d, weight ld %d: multiples of 4), dm %d, R %lld", dq, ldq, ldwq, dm, (long long)R);
    hipStream_t s = as_stream(stream);
    float* partial = static_cast<float*>(workspace);
    const int chunks = R < QKB_CHUNKS ? (int)R : QKB_CHUNKS;
    qk_bias_rows_kernel<<<dim3((unsigned)chunks), dim3(256), (4 * ((dq + 3) / 4) + 4 * 256) * sizeof(float), s>>>(Q, ldq, dq, dm, d_c0, R, wq, ldwq, bk, d_Q, partial);
    PAPR_CHECK_LAUNCH("qk_bias_rows");
    qk_bias_finish_kernel<<<dim3(16), dim3(1024), (dq + 1 + 1024) * sizeof(float), s>>>(partial, chunks, dq, dm, wq, ldwq, bk, bq, d_wq, d_bq_in, d_bq, d_bk);
    PAPR_CHECK_LAUNCH("qk_bias_finish");
    return 0;
}

// The MSE training loss (torch.nn.MSELoss(), the reference's models/__init__.py:8-52 `mse` term) forward and its gradient direction in ONE launch:
// loss = mean((pred - target)^2), grad[i] = 2 (pred[i] - target[i]) / n (the backward pass multiplies it by the incoming scalar).  A patch of rays
// is 76,800 numbers: torch spends an elementwise launch, a two-stage 16-us reduction and three more launches backwards on it.  Up to 64 workgroups
// leave fp64 partial sums; the one that arrives last (a ticket counter in the workspace, which it resets) adds them in index order.
namespace {
constexpr int MSE_MAX_WGS = 64;
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, long n, float* __restrict__ loss,
                                                      float* __restrict__ grad, unsigned* __restrict__ ticket, double* __restrict__ partial) {
    __shared__ double part[4];
    __shared__ bool last;
    const float two_inv = 2.0f / (float)n;
    double acc = 0.0;
    const long n4 = (n % 4 == 0 && (reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(grad)) % 16 == 0) ? n / 4 : 0;
    const long stride = (long)gridDim.x * 256, t0 = (long)blockIdx.x * 256 + threadIdx.x;
    for (long i = t0; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(pred)[i], b = reinterpret_cast<const float4*>(target)[i];
        const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        acc += (double)(d.x * d.x) + (double)(d.y * d.y) + (double)(d.z * d.z) + (double)(d.w * d.w);
        if (grad) reinterpret_cast<float4*>(grad)[i] = make_float4(d.x * two_inv, d.y * two_inv, d.z * two_inv, d.w * two_inv);
    }
    for (long i = 4 * n4 + t0; i < n; i += stride) {
        const float d = pred[i] - target[i];
        acc += (double)(d * d);
        if (grad) grad[i] = d * two_inv;
    }
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(partial + blockIdx.x, (part[0] + part[1]) + (part[2] + part[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        double t = 0.0;
        for (unsigned w = 0; w < gridDim.x; ++w) t += __hip_atomic_load(partial + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *loss = (float)(t / (double)n);
        *ticket = 0u;                                // (for the next call on this workspace)
    }
}
}  // namespace

extern "C" size_t papr_mse_workspace_bytes(void) { return 16 + MSE_MAX_WGS * sizeof(double); }

extern "C" int papr_mse_fwd(const float* pred, const float* target, int64_t n, float* loss, float* grad, void* workspace, papr_stream_t stream) {
    PAPR_REQUIRE(pred && target && loss && workspace && n >= 1, "papr_mse_fwd: null pointer or empty input");
    long wgs = (n / 4 + 255) / 256;
    wgs = wgs < 1 ? 1 : (wgs > MSE_MAX_WGS ? MSE_MAX_WGS : wgs);
    // the ticket is zeroed IN the stream in front of every launch (4 bytes): the last workgroup's reset alone would leave a launch that was aborted or
    // faulted mid-flight behind as a ticket that never reaches the count again -- every later loss on the stream unfinalised, silently
    PAPR_REQUIRE(hipMemsetAsync(workspace, 0, 4, as_stream(stream)) == hipSuccess, "papr_mse_fwd: memset of the ticket failed");
    mse_fwd_kernel<<<dim3((unsigned)wgs), dim3(256), 0, as_stream(stream)>>>(pred, target, n, loss, grad, static_cast<unsigned*>(workspace),
                                                                            reinterpret_cast<double*>(static_cast<char*>(workspace) + 16));
    PAPR_CHECK_LAUNCH("mse_fwd");
    return 0;
}

extern "C" int papr_rownorm_bwd(const float* dy, const float* y, const float* stats, int64_t rows, int width, int ld,
                                float eps, float* dx, papr_stream_t stream) {
    (void)eps;
    PAPR_REQUIRE(dy && y && stats && dx, "papr_rownorm_bwd: null pointer");
    PAPR_REQUIRE(width >= 2 && width <= 64 * MAX_VPL && ld >= width, "papr_rownorm_bwd: width %d / ld %d unsupported", width, ld);
    if (rows <= 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (width <= 256) rownorm_bwd_kernel<4><<<grid, block, 0, as_stream(stream)>>>(dy, y, stats, rows, width, ld, dx);
    else rownorm_bwd_kernel<MAX_VPL><<<grid, block, 0, as_stream(stream)>>>(dy, y, stats, rows, width, ld, dx);
    PAPR_CHECK_LAUNCH("rownorm_bwd");
    return 0;
}

static int check_tail(const papr_tail_desc* d, const char* who) {
    PAPR_REQUIRE(d, "%s: null descriptor", who);
    PAPR_REQUIRE(d->k >= 1 && d->k <= 255, "%s: k=%d outside [1,255]", who, d->k);
    PAPR_REQUIRE(d->d_model % 4 == 0 && d->ld_kp % 4 == 0 && d->ld_qp % 4 == 0, "%s: d_model and strides must be multiples of 4", who);
    PAPR_REQUIRE(d->ld_kp >= d->d_model && d->ld_qp >= d->d_model && d->ld_v >= d->C, "%s: strides too small", who);
    return 0;
}

extern "C" int papr_attn_tail_fwd(const papr_tail_desc* d, const float* kp, const float* qp, const float* score_bias,
                                  const float* v, const float* influ, const int32_t* idx, int64_t R, float* scores, float* attn,
                                  float* fused, papr_stream_t stream) {
    if (int e = check_tail(d, "papr_attn_tail_fwd")) return e;
    PAPR_REQUIRE(kp && (qp || d->precomputed_dots) && v && influ && idx && scores && attn && fused, "papr_attn_tail_fwd: null pointer");
    if (R <= 0) return 0;
    const dim3 grid((unsigned)((R + 3) / 4)), block(256);
    if (d->k <= 63) tail_fwd_kernel<<<grid, block, 0, as_stream(stream)>>>(*d, kp, qp, v, score_bias, influ, idx, R, scores, attn, fused);
    else if (d->k <= 127) tail_fwd_wide_kernel<2><<<grid, block, 0, as_stream(stream)>>>(*d, kp, qp, v, score_bias, influ, idx, R, scores, attn, fused);
    else tail_fwd_wide_kernel<4><<<grid, block, 0, as_stream(stream)>>>(*d, kp, qp, v, score_bias, influ, idx, R, scores, attn, fused);
    PAPR_CHECK_LAUNCH("tail_fwd");
    return 0;
}

extern "C" int papr_attn_tail_bwd(const papr_tail_desc* d, const float* kp, const float* qp, const float* v,
                                  const float* influ, const int32_t* idx, int64_t R, const float* scores,
                                  const float* attn, const float* d_fused, const float* d_attn, float* d_kp,
                                  float* d_qp, float* d_v, float* d_influ, float* d_score_bias, float* d_pair_influ,
                                  const float* kp_norm_stats, const float* score_bias, const float* kp_mean,
                                  const papr_f16_rows* d_kp_f16, const papr_f16_rows* d_v_f16, papr_stream_t stream) {
    if (int e = check_tail(d, "papr_attn_tail_bwd")) return e;
    PAPR_REQUIRE(!kp_mean || kp_norm_stats, "papr_attn_tail_bwd: kp_mean (raw kp rows) needs kp_norm_stats");
    PAPR_REQUIRE(kp && qp && v && influ && idx && scores && attn && d_fused && (d_kp || d_kp_f16) && d_qp && (d_v || d_v_f16) && (d_influ || d_pair_influ),
                 "papr_attn_tail_bwd: null pointer");
    papr_f16_rows kp16 = {}, v16 = {};
    if (d_kp_f16) {
        kp16 = *d_kp_f16;
        PAPR_REQUIRE(kp16.hi && kp16.inv && kp16.scale && kp16.max && d->k <= 63 && d->d_model == 256 && kp16.ld >= 256 && kp16.ld % 32 == 0,
                     "papr_attn_tail_bwd: d_kp_f16 needs k <= 63, d_model = 256 and rows of at least 256 halfs (a multiple of 32)");
    }
    if (d_v_f16) {
        v16 = *d_v_f16;
        const int g4 = d->C >> 2;
        PAPR_REQUIRE(v16.hi && v16.inv && v16.scale && v16.max && d->k <= 63 && (d->C & 3) == 0 && d->ld_v == d->C && g4 >= 1 && g4 <= 32 && 64 % g4 == 0 &&
                     v16.ld >= d->C && v16.ld % 32 == 0,
                     "papr_attn_tail_bwd: d_v_f16 needs k <= 63, C = ld_v a multiple of 4 that divides 256, rows of a multiple of 32 halfs");
    }
    if (R <= 0) return 0;
    const dim3 grid((unsigned)((R + 3) / 4)), block(256);
    if (d->k <= 63)
        tail_bwd_kernel<<<grid, block, 0, as_stream(stream)>>>(*d, kp, qp, v, influ, idx, R, scores, attn, d_fused, d_attn, d_kp, d_qp, d_v, d_influ,
                                                               d_score_bias, d_pair_influ, kp_norm_stats, score_bias, kp_mean, kp16, v16);
    else if (d->k <= 127)
        tail_bwd_wide_kernel<2><<<grid, block, 0, as_stream(stream)>>>(*d, kp, qp, v, influ, idx, R, scores, attn, d_fused, d_attn, d_kp, d_qp, d_v, d_influ,
                                                                       d_score_bias, d_pair_influ, kp_norm_stats, score_bias, kp_mean);
    else
        tail_bwd_wide_kernel<4><<<grid, block, 0, as_stream(stream)>>>(*d, kp, qp, v, influ, idx, R, scores, attn, d_fused, d_attn, d_kp, d_qp, d_v, d_influ,
                                                                       d_score_bias, d_pair_influ, kp_norm_stats, score_bias, kp_mean);
    PAPR_CHECK_LAUNCH("tail_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Background compositing, the last line of the attention tail (reference models/model.py:536-545):
//     rgb = fg * (1 - a) + bkg * a      (normalize_topk_attn)          rgb = fg + bkg * a      (otherwise)
// with a = the background token's attention (column k of attn), fg the render head's output (R, C <= 8), bkg (C).  In torch
// ops the line and its autograd are ~20 launches of 4-27 us over 25,600 x 3 values (0.13 ms per step); here one launch forward
// and two backward (the second adds the workgroups' partial sums of d_bkg in a fixed order).
namespace {

constexpr int COMP_MAXC = 8;

__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ fg, const float* __restrict__ attn, int ld_attn, int col,
                                                            const float* __restrict__ bkg, long R, int Cn, int normalize, float* __restrict__ rgb) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= R) return;
    const float a = attn[p * ld_attn + col], keep = normalize ? 1.0f - a : 1.0f;
    for (int c = 0; c < Cn; ++c) rgb[p * Cn + c] = fg[p * Cn + c] * keep + bkg[c] * a;
}

// d_fg = d_rgb * keep; d_attn row = zeros except column `col`: sum_c d_rgb (bkg - fg) (normalize) or sum_c d_rgb bkg;
// partial[block][c] = sum over the block's pixels of d_rgb a
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ d_rgb, const float* __restrict__ fg, const float* __restrict__ attn,
                                                            int ld_attn, int col, const float* __restrict__ bkg, long R, int Cn, int normalize,
                                                            float* __restrict__ d_fg, float* __restrict__ d_attn, float* __restrict__ partial) {
    __shared__ float red[4][COMP_MAXC];
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    float sb[COMP_MAXC];
#pragma unroll
    for (int c = 0; c < COMP_MAXC; ++c) sb[c] = 0.f;
    if (p < R) {
        const float a = attn[p * ld_attn + col], keep = normalize ? 1.0f - a : 1.0f;
        float da = 0.f;
#pragma unroll
        for (int c = 0; c < COMP_MAXC; ++c)
            if (c < Cn) {
                const float g = d_rgb[p * Cn + c], f = fg[p * Cn + c];
                if (d_fg) d_fg[p * Cn + c] = g * keep;
                da += g * (normalize ? bkg[c] - f : bkg[c]);
                sb[c] = g * a;
            }
        if (d_attn) {
            for (int j = 0; j < ld_attn; ++j) d_attn[p * ld_attn + j] = j == col ? da : 0.f;
        }
    }
    if (partial) {
#pragma unroll
        for (int c = 0; c < COMP_MAXC; ++c) {
            float v = sb[c];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);      // (a butterfly: every lane ends with the same sum, in a fixed order)
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = v;
        }
        __syncthreads();
        if ((int)threadIdx.x < Cn) partial[(long)blockIdx.x * COMP_MAXC + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void composite_bkg_reduce_kernel(const float* __restrict__ partial, int blocks, int Cn, float* __restrict__ d_bkg) {
    __shared__ float red[4];
    for (int c = 0; c < Cn; ++c) {
        float v = 0.f;
        for (int b = threadIdx.x; b < blocks; b += 256) v += partial[(long)b * COMP_MAXC + c];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) d_bkg[c] = ((red[0] + red[1]) + red[2]) + red[3];
        __syncthreads();
    }
}

}  // namespace

extern "C" int papr_composite_fwd(const float* fg, const float* attn, int32_t ld_attn, int32_t col, const float* bkg, int64_t R, int32_t C,
                                  int32_t normalize, float* rgb, papr_stream_t stream) {
    PAPR_REQUIRE(fg && attn && bkg && rgb, "papr_composite_fwd: null pointer");
    PAPR_REQUIRE(C >= 1 && C <= COMP_MAXC && col >= 0 && col < ld_attn, "papr_composite_fwd: C %d (1 .. %d), column %d of %d", C, COMP_MAXC, col, ld_attn);
    if (R <= 0) return 0;
    composite_fwd_kernel<<<dim3((unsigned)((R + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(fg, attn, ld_attn, col, bkg, R, C, normalize, rgb);
    PAPR_CHECK_LAUNCH("composite_fwd");
    return 0;
}

extern "C" size_t papr_composite_bwd_workspace_bytes(int64_t R) { return (size_t)((R + 255) / 256) * COMP_MAXC * sizeof(float); }

extern "C" int papr_composite_bwd(const float* d_rgb, const float* fg, const float* attn, int32_t ld_attn, int32_t col, const float* bkg, int64_t R,
                                  int32_t C, int32_t normalize, float* d_fg, float* d_attn, float* d_bkg, void* workspace, papr_stream_t stream) {
    PAPR_REQUIRE(d_rgb && fg && attn && bkg, "papr_composite_bwd: null pointer");
    PAPR_REQUIRE(C >= 1 && C <= COMP_MAXC && col >= 0 && col < ld_attn, "papr_composite_bwd: C %d (1 .. %d), column %d of %d", C, COMP_MAXC, col, ld_attn);
    PAPR_REQUIRE(!d_bkg || workspace, "papr_composite_bwd: d_bkg needs the workspace");
    if (R <= 0) return 0;
    const int blocks = (int)((R + 255) / 256);
    hipStream_t s = as_stream(stream);
    composite_bwd_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(d_rgb, fg, attn, ld_attn, col, bkg, R, C, normalize, d_fg, d_attn,
                                                                      d_bkg ? static_cast<float*>(workspace) : nullptr);
    PAPR_CHECK_LAUNCH("composite_bwd");
    if (d_bkg) {
        composite_bkg_reduce_kernel<<<dim3(1), dim3(256), 0, s>>>(static_cast<const float*>(workspace), blocks, C, d_bkg);
        PAPR_CHECK_LAUNCH("composite_bkg_reduce");
    }
    return 0;
}
