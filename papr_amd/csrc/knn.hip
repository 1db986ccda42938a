// K1: for every ray, the k points of the cloud nearest to the ray (perpendicular distance).
//
// Replaces PAPR._calculate_global_distances (reference models/model.py:258-283), which materialises
// five R x P fp32 tensors and runs torch.topk over them.  Here nothing of size R x P ever exists.
//
// Work decomposition (wave64, one wave = one tile of T rays against ALL points):
//   * the 64 lanes of a wave each hold PPL points of the current batch in VGPRs (coalesced 16-byte
//     loads of a {x,y,z,index} stream); the batch is reused for all T rays of the tile, so the point
//     stream is read once per T rays;
//   * the ray constants (origin, direction, d.d+eps and its reciprocal) are wave-uniform and come
//     in through scalar loads (s_load -> SGPRs), so every VALU op has one VGPR and one SGPR operand;
//   * each ray's running k-set lives ACROSS the lanes of one VGPR pair (lanes 0..k-1, unordered);
//     its largest member (the k-th distance so far) and the lane holding it sit in SGPRs.  A
//     candidate test is one v_cmp + ballot against that SGPR; an insertion overwrites the max lane
//     and re-derives the max with a 6-step DPP reduction -- pure VALU, no LDS, no divergence.  The
//     set is ranked once at the end so the output is ascending in (distance, index);
//   * the rays of a tile are neighbouring pixels, so only the first one pays for a cold start: the
//     others start from its neighbour set (see ray_knn_kernel).
//
// The stream is a scattered copy of the cloud (position i holds point (i*A) mod P, A coprime to P,
// written by a tiny pre-pass): an index-ordered walk over a lattice-initialised (or otherwise
// spatially sorted) cloud approaches every ray monotonically and turns most points into
// insertions; a scattered walk tightens the k-th distance after a few hundred points, leaving about
// k (1 + ln(P/k)) insertions per ray.
//
// Arithmetic follows the reference formulation operation by operation (no FMA contraction, IEEE
// division, torch.norm's fma chain) so that near-tie neighbours resolve the same way:
// v = p - o; t = (v.d)/(d.d+eps); D = v - d t; select on |D|^2 (sqrt is monotone).  Exact ties at
// the k-th distance are unordered in the reference (topk sorted=False); here the first one met in
// the (fixed) stream order is kept, so results are reproducible run to run.
#include "papr_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(4))) const float cfloat;

// ray record: ox oy oz dx dy dz den rcp
__global__ __launch_bounds__(256) void pack_rays_kernel(const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d, long R,
                                                        long rays_per_image, float eps,
                                                        float* __restrict__ rec) {
    long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    long n = r / rays_per_image;
    float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
    float den = ((dx * dx + dy * dy) + dz * dz) + eps;
    float4 a = make_float4(rays_o[n * 3 + 0], rays_o[n * 3 + 1], rays_o[n * 3 + 2], dx);
    float4 b = make_float4(dy, dz, den, 1.0f / den);
    reinterpret_cast<float4*>(rec)[r * 2 + 0] = a;
    reinterpret_cast<float4*>(rec)[r * 2 + 1] = b;
}

// stream[i] = {xyz of point (i*mul) mod P, bit pattern of that index}
__global__ __launch_bounds__(256) void scatter_points_kernel(const float* __restrict__ points, int P, int mul,
                                                             float4* __restrict__ stream) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    int src = (int)(((long)i * mul) % P);
    stream[i] = make_float4(points[src * 3 + 0], points[src * 3 + 1], points[src * 3 + 2], __int_as_float(src));
}

__device__ __forceinline__ float read_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Squared distances are >= 0, so their bit patterns order like unsigned integers: the k-set is kept as
// raw bits and reduced with v_max_u32, which takes a DPP operand directly (a float max would drag in
// canonicalisation moves).  Identity 0 + bound_ctrl lets the compiler fold each step into one VALU op.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax(unsigned v) {
    unsigned moved = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
    return v > moved ? v : moved;
}

// maximum over the 64 lanes, returned wave-uniform (gfx9 DPP: quad_perm, row_half_mirror, row_mirror,
// row_bcast:15, row_bcast:31; the total lands in lane 63)
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = dpp_umax<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_umax<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_umax<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_umax<0x140, 0xf>(v);   // row_mirror
    v = dpp_umax<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_umax<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

struct RayK {   // wave-uniform ray constants (SGPRs)
    float ox, oy, oz, dx, dy, dz, den, rcp;
};

// |D|^2 of point p for ray c, in the reference's operation order
__device__ __forceinline__ float ray_dist2(const RayK& c, float px, float py, float pz) {
    float vx = px - c.ox, vy = py - c.oy, vz = pz - c.oz;
    float vd = (vx * c.dx + vy * c.dy) + vz * c.dz;
    // correctly rounded vd / den from the exact reciprocal (one Newton step on the quotient)
    float q0 = vd * c.rcp;
    float rem = __builtin_fmaf(-q0, c.den, vd);
    float tt = __builtin_fmaf(rem, c.rcp, q0);
    float ex = vx - c.dx * tt, ey = vy - c.dy * tt, ez = vz - c.dz * tt;
    // torch.norm accumulates its squares with an fma chain: fma(z,z, fma(y,y, x*x))
    return __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
}

// The same quantity with contracted arithmetic: 13 VALU operations instead of 22.  NOT used for the selection itself -- only as a
// conservative filter in front of it: a point whose cheap distance lies more than 1/64 above the current k-th distance cannot beat
// it in exact arithmetic either (the two evaluations differ by rounding only: relative 1e-7 x |v| / |D|, i.e. below 1e-4 for any
// point near the threshold in scenes of this scale), so the exact, reference-ordered evaluation runs only for the few batches that hold
// such a point (a few percent once the threshold has settled).  Round 2's profile had the kernel bound by exactly these VALU operations.
__device__ __forceinline__ float ray_dist2_fast(const RayK& c, float px, float py, float pz) {
    const float vx = px - c.ox, vy = py - c.oy, vz = pz - c.oz;
    const float tt = __builtin_fmaf(vz, c.dz, __builtin_fmaf(vy, c.dy, vx * c.dx)) * c.rcp;
    const float ex = __builtin_fmaf(-c.dx, tt, vx), ey = __builtin_fmaf(-c.dy, tt, vy), ez = __builtin_fmaf(-c.dz, tt, vz);
    return __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
}

constexpr unsigned INF_BITS = 0x7f800000u;

// One ray's k-set: lanes < k hold (distance bits, point index), unordered; lanes >= k hold 0 bits.
struct KSet {
    unsigned bits;   // per lane
    int idx;         // per lane
    unsigned thr;    // uniform: largest member = current k-th smallest
    unsigned thr_hi; // uniform: thr x (1 + 1/64): what the cheap distance is tested against
    int tl;          // uniform: a lane holding it
    __device__ __forceinline__ void refresh() {
        thr = wave_umax(bits);
        thr_hi = __float_as_uint(__uint_as_float(thr) * 1.015625f);
        tl = __builtin_ctzll(__ballot(bits == thr));
    }
    // does any lane hold a point that may enter the set? (cheap distances)
    __device__ __forceinline__ bool may_enter(float d2_fast) const { return __ballot(__float_as_uint(d2_fast) < thr_hi) != 0; }
    // offer the candidates flagged in d2 < thr; DEDUP skips points that are already members (seeded sets)
    template <bool DEDUP>
    __device__ __forceinline__ void offer(float d2, int pidx, int lane) {
        unsigned long long m = __ballot(__float_as_uint(d2) < thr);
        while (m) {
            int src = __builtin_ctzll(m);
            m &= m - 1;
            unsigned cb = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(d2), src);
            if (cb < thr) {
                int ci = __builtin_amdgcn_readlane(pidx, src);
                if (DEDUP && __ballot(idx == ci) != 0) continue;
                bool here = lane == tl;
                bits = here ? cb : bits;
                idx = here ? ci : idx;
                refresh();
            }
        }
    }
};

// Phase A finds ray 0's neighbours by streaming the whole cloud.  The other T-1 rays of the tile are
// adjacent pixels: their k-sets are SEEDED with ray 0's neighbours (exact distances recomputed per
// ray), which puts their thresholds within a few percent of final before the stream starts, so phase
// B performs a handful of insertions per ray instead of ~k(1 + ln(P/k)).  Exactness does not depend
// on the seed: any closer point met in the stream still replaces the current maximum.
template <int T, int PPL>
__global__ __launch_bounds__(256) void ray_knn_kernel(const float4* __restrict__ stream, const float* __restrict__ points,
                                                      int P, const float* rec, long R, int k, int* __restrict__ out_idx,
                                                      float* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long tile = (long)blockIdx.x * 4 + wave;
    const long r0 = tile * T;
    if (r0 >= R) return;

    cfloat* crec = (cfloat*)rec;
    RayK rk[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        long r = r0 + t < R ? r0 + t : R - 1;
        cfloat* p = crec + r * 8;
        rk[t].ox = p[0]; rk[t].oy = p[1]; rk[t].oz = p[2]; rk[t].dx = p[3];
        rk[t].dy = p[4]; rk[t].dz = p[5]; rk[t].den = p[6]; rk[t].rcp = p[7];
    }
    KSet ks[T];
    ks[0].bits = lane < k ? INF_BITS : 0u; ks[0].idx = -1; ks[0].thr = INF_BITS; ks[0].thr_hi = INF_BITS; ks[0].tl = 0;

    // ---- phase A: ray 0 against every point
    for (int base = 0; base < P; base += 64 * PPL) {
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            int pi = base + q * 64 + lane;
            bool ok = pi < P;
            float4 v = stream[ok ? pi : P - 1];
            if (ks[0].may_enter(ok ? ray_dist2_fast(rk[0], v.x, v.y, v.z) : INFINITY)) {
                float d2 = ray_dist2(rk[0], v.x, v.y, v.z);
                ks[0].template offer<false>(ok ? d2 : INFINITY, __float_as_int(v.w), lane);
            }
        }
    }
    // ---- seed rays 1..T-1 with ray 0's neighbours
    {
        int pi = lane < k ? ks[0].idx : 0;
        float sx = points[pi * 3 + 0], sy = points[pi * 3 + 1], sz = points[pi * 3 + 2];
#pragma unroll
        for (int t = 1; t < T; ++t) {
            float d2 = ray_dist2(rk[t], sx, sy, sz);
            ks[t].bits = lane < k ? __float_as_uint(d2) : 0u;
            ks[t].idx = lane < k ? ks[0].idx : -1;
            ks[t].refresh();
        }
    }
    // ---- phase B: rays 1..T-1 against every point
    for (int base = 0; base < P; base += 64 * PPL) {
        float px[PPL], py[PPL], pz[PPL];
        int pidx[PPL];
        bool ok[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            int pi = base + q * 64 + lane;
            ok[q] = pi < P;
            float4 v = stream[ok[q] ? pi : P - 1];
            px[q] = v.x; py[q] = v.y; pz[q] = v.z; pidx[q] = __float_as_int(v.w);
        }
#pragma unroll
        for (int t = 1; t < T; ++t) {
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                if (ks[t].may_enter(ok[q] ? ray_dist2_fast(rk[t], px[q], py[q], pz[q]) : INFINITY)) {
                    float d2 = ray_dist2(rk[t], px[q], py[q], pz[q]);
                    ks[t].template offer<true>(ok[q] ? d2 : INFINITY, pidx[q], lane);
                }
            }
        }
    }
    // rank the k members of each set by (distance, index) and write them in ascending order
#pragma unroll
    for (int t = 0; t < T; ++t) {
        long r = r0 + t;
        if (r >= R) continue;
        int rank = 0;
        for (int j = 0; j < k; ++j) {
            unsigned dj = (unsigned)__builtin_amdgcn_readlane((int)ks[t].bits, j);
            int ij = __builtin_amdgcn_readlane(ks[t].idx, j);
            rank += (dj < ks[t].bits || (dj == ks[t].bits && ij < ks[t].idx)) ? 1 : 0;
        }
        if (lane < k) {
            out_idx[r * k + rank] = ks[t].idx;
            if (out_dist) out_dist[r * k + rank] = sqrtf(__uint_as_float(ks[t].bits));
        }
    }
}

int coprime_stride(long n) {
    long mul = (long)(n * 0.6180339887) | 1;
    auto gcd = [](long a, long b) { while (b) { long t = a % b; a = b; b = t; } return a; };
    while (gcd(mul, n) != 1) mul += 2;
    return (int)(mul % n);
}

}  // namespace

extern "C" size_t papr_ray_knn_workspace_bytes(int64_t R, int64_t P) {
    return (size_t)R * 8 * sizeof(float) + (size_t)P * 4 * sizeof(float);
}

extern "C" int papr_ray_knn(const float* points, int64_t P, const float* rays_o, const float* rays_d,
                            int64_t R, int64_t rays_per_image, int k, float eps, int32_t* out_idx,
                            float* out_dist, void* workspace, papr_stream_t stream) {
    PAPR_REQUIRE(points && rays_o && rays_d && out_idx && workspace, "papr_ray_knn: null pointer");
    PAPR_REQUIRE(k >= 1 && k <= 64, "papr_ray_knn: k=%d outside [1,64]", k);
    PAPR_REQUIRE(P >= k && P * 3 < (int64_t)1 << 31, "papr_ray_knn: P=%lld must be >= k and < 2^31/3", (long long)P);
    PAPR_REQUIRE(rays_per_image >= 1, "papr_ray_knn: rays_per_image must be positive");
    if (R <= 0) return 0;
    hipStream_t s = as_stream(stream);
    float* rec = static_cast<float*>(workspace);
    float4* pstream = reinterpret_cast<float4*>(rec + (size_t)R * 8);
    pack_rays_kernel<<<dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s>>>(rays_o, rays_d, R, rays_per_image, eps, rec);
    PAPR_CHECK_LAUNCH("pack_rays");
    scatter_points_kernel<<<dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s>>>(points, (int)P, P > 1 ? coprime_stride(P) : 0, pstream);
    PAPR_CHECK_LAUNCH("scatter_points");
    // Rays per wave T: a tile costs ~(1.8 + T - 1) ray passes over the cloud (its first ray starts cold, the others are seeded), and the
    // launch takes as long as the SIMD with the most tiles: ceil(tiles / SIMDs) of them.  With T = 8 a 160 x 160 patch is 3,200 tiles on
    // 1,024 SIMDs -- 4 on some, 3.125 on average, a quarter of the machine idle at the end (round 2); T = 5 makes it exactly 5 each.
    constexpr int PPL = 4;
    static int n_simd = 0;
    if (!n_simd) { int dev = 0, cu = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev); n_simd = 4 * (cu > 0 ? cu : 256); }
    static const int t_env = getenv("PAPR_KNN_T") ? atoi(getenv("PAPR_KNN_T")) : 0;      // (A/B switch)
    int T = 8;
    double best = 1e300;
    for (int t : {4, 5, 6, 8}) {
        const long tl = (R + t - 1) / t;
        const double cost = (double)((tl + n_simd - 1) / n_simd) * (0.8 + t);
        if (cost < best - 1e-9) { best = cost; T = t; }
    }
    if (t_env == 4 || t_env == 5 || t_env == 6 || t_env == 8) T = t_env;
    const long tiles = (R + T - 1) / T;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin(5, R, (int)P, k, s);
    if (T == 4) ray_knn_kernel<4, PPL><<<grid, block, 0, s>>>(pstream, points, (int)P, rec, R, k, out_idx, out_dist);
    else if (T == 5) ray_knn_kernel<5, PPL><<<grid, block, 0, s>>>(pstream, points, (int)P, rec, R, k, out_idx, out_dist);
    else if (T == 6) ray_knn_kernel<6, PPL><<<grid, block, 0, s>>>(pstream, points, (int)P, rec, R, k, out_idx, out_dist);
    else ray_knn_kernel<8, PPL><<<grid, block, 0, s>>>(pstream, points, (int)P, rec, R, k, out_idx, out_dist);
    if (prof) papr_prof_end(s);
    PAPR_CHECK_LAUNCH("ray_knn");
    return 0;
}
