// K1: for every ray, the k points of the cloud nearest to the ray (perpendicular distance).
//
// Replaces PAPR._calculate_global_distances (reference models/model.py:258-283), which materialises
// five R x P fp32 tensors and runs torch.topk over them.  Here nothing of size R x P ever exists.
//
// Two forms.  Clouds of 2,048 points and more: the SPATIAL form further down (ray_knn_blocks_kernel: the cloud binned and cut into blocks
// of 64 points with bounding spheres, a ray walks only the blocks that can hold a nearer point; 197 -> 158 us at P = 10,000, 448 -> 235 us
// at 30,000, per 25,600 rays, binning included).  Smaller clouds, and PAPR_KNN_BLOCKS=0: every point against every ray, described here.
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
// the k-th distance are unordered in the reference (topk sorted=False); both forms keep the smallest
// indices (the set is the k smallest by (distance, index)), so they agree bit for bit -- a training run is the
// same with either (scripts/probes/knn_forms_train.sh).
#include "papr_common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

typedef __attribute__((address_space(4))) const float cfloat;

// ray record: ox oy oz dx dy dz den rcp
__global__ __launch_bounds__(256) void pack_rays_kernel(const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d, long R,
                                                        long rays_per_image, float eps,
                                                        float* __restrict__ rec, unsigned* __restrict__ zero, int n_zero,
                                                        float4* __restrict__ pad, long n_pad) {
    long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long i = r; i < n_zero; i += (long)gridDim.x * blockDim.x) zero[i] = 0u;      // (the binning's cell counters, cursors, block count)
    for (long i = r; i < n_pad; i += (long)gridDim.x * blockDim.x) pad[i] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));      // (stream places no point takes)
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

// A LOWER BOUND of the same quantity with contracted arithmetic: 17 VALU operations instead of 22.  NOT used for the selection itself -- only as a
// conservative filter in front of it.  The two evaluations differ by rounding only: |d2_fast - d2| <= c eps |v| |D| + (c eps |v|)^2 with c ~ 4
// (the error of e = v - d t is ~ eps |v| per component).  Relative to d2 = |D|^2 that is c eps |v| / |D|: below 1/64 unless |D| < 1.5e-5 |v|, and
// there the absolute difference is below 4e-12 |v|^2.  So a point with exact d2 <= thr always has
//     max(d2_fast - 2^-30 |v|^2, 0) <= thr (1 + 1/64):
// the relative slack covers every point further than 1.5e-5 |v| from the ray, the absolute term (250x the bound) the ones closer -- coincident
// and on-ray points (duplicates after add_points), thr == 0 included (the test is <=).  The exact, reference-ordered evaluation runs only for the
// few batches that hold such a point (a few percent once the threshold has settled).
__device__ __forceinline__ float ray_dist2_fast(const RayK& c, float px, float py, float pz) {
    const float vx = px - c.ox, vy = py - c.oy, vz = pz - c.oz;
    const float tt = __builtin_fmaf(vz, c.dz, __builtin_fmaf(vy, c.dy, vx * c.dx)) * c.rcp;
    const float ex = __builtin_fmaf(-c.dx, tt, vx), ey = __builtin_fmaf(-c.dy, tt, vy), ez = __builtin_fmaf(-c.dz, tt, vz);
    const float d2 = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
    const float vv = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
    return fmaxf(__builtin_fmaf(-0x1p-30f, vv, d2), 0.f);
}


// One ray's k-set: lanes < k hold (distance bits, point index), unordered; lanes >= k hold 0 bits.
constexpr unsigned INF_BITS = 0x7f800000u;

struct KSet {
    unsigned bits;   // per lane
    int idx;         // per lane (-1: not filled yet -- counts as the largest index)
    unsigned thr;    // uniform: largest member = current k-th smallest
    unsigned thr_idx;// uniform: its index
    unsigned thr_hi; // uniform: thr x (1 + 1/64): what the cheap lower bound of the distance is tested against (<=)
    int tl;          // uniform: the lane holding it
    unsigned long long kmask;       // uniform: lanes 0 .. k-1
    // the set is the k smallest by (distance bits, index) -- the same total order as the spatial form's, so that both forms give the same answer at
    // exact ties of the k-th distance as well (and a training run the same bits with either)
    __device__ __forceinline__ void refresh() {
        thr = wave_umax(bits);
        unsigned long long mm = __ballot(bits == thr) & kmask;
        if (mm & (mm - 1)) {                        // several members at the largest distance: the one with the largest index goes first
            const unsigned mi = wave_umax(bits == thr ? (unsigned)idx : 0u);
            mm = __ballot(bits == thr && (unsigned)idx == mi) & kmask;
        }
        tl = __builtin_ctzll(mm);
        thr_idx = (unsigned)__builtin_amdgcn_readlane(idx, tl);
        thr_hi = __float_as_uint(__uint_as_float(thr) * 1.015625f);
    }
    // does any lane hold a point that may enter the set? (cheap distances)
    __device__ __forceinline__ bool may_enter(float d2_fast) const { return __ballot(__float_as_uint(d2_fast) <= thr_hi) != 0; }
    // offer the candidates; DEDUP skips points that are already members (seeded sets).  (An infinite d2 marks a lane without a point.)
    template <bool DEDUP>
    __device__ __forceinline__ void offer(float d2, int pidx, int lane) {
        unsigned long long m = __ballot(__float_as_uint(d2) <= thr && __float_as_uint(d2) < INF_BITS);
        while (m) {
            int src = __builtin_ctzll(m);
            m &= m - 1;
            unsigned cb = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(d2), src);
            int ci = __builtin_amdgcn_readlane(pidx, src);
            if (cb < thr || (cb == thr && (unsigned)ci < thr_idx)) {
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
    const unsigned long long kmask = k >= 64 ? ~0ull : (1ull << k) - 1ull;
#pragma unroll
    for (int t = 0; t < T; ++t) ks[t].kmask = kmask;
    ks[0].bits = lane < k ? INF_BITS : 0u; ks[0].idx = -1; ks[0].thr = INF_BITS; ks[0].thr_idx = 0xffffffffu; ks[0].thr_hi = INF_BITS; ks[0].tl = 0;

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
        int pi = (lane < k && ks[0].idx >= 0) ? ks[0].idx : 0;         // (a slot nothing entered -- NaN coordinates -- holds -1)
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
            rank += (dj < ks[t].bits || (dj == ks[t].bits && (ij < ks[t].idx || (ij == ks[t].idx && j < lane)))) ? 1 : 0;     // (equal entries: only slots nothing entered, idx -1 -- by lane, so that every output slot is written)
        }
        if (lane < k) {
            out_idx[r * k + rank] = ks[t].idx < 0 ? 0 : ks[t].idx;      // (a slot no candidate entered -- NaN coordinates: a valid index, as torch.topk of NaN distances gives; -1 sent the gathers out of bounds)
            if (out_dist) out_dist[r * k + rank] = sqrtf(__uint_as_float(ks[t].bits));
        }
    }
}

// ================================================================================================
// k > 64 (round 5; the reference's topk takes any k < P, models/model.py:281): the set no longer fits one register across the wave.  NS registers
// per lane hold member (slot s, lane l) = entry 64 s + l; everything else is the every-point form above -- the same reference-ordered distance, the
// same total order (distance bits, index), the same conservative filter -- one ray per wave, no seeding (no shipped configuration asks for such
// a k: the form exists for generality, not for speed: ~k (1 + ln(P / k)) insertions of ~40 instructions per ray).
template <int NS>
struct KSetW {
    unsigned bits[NS];
    int idx[NS];
    unsigned thr, thr_idx, thr_hi;
    int tl, ts;                     // the lane and the slot that hold the largest member
    int k;
    __device__ __forceinline__ bool member(int s, int lane) const { return 64 * s + lane < k; }
    __device__ __forceinline__ void refresh(int lane) {
        unsigned m = 0u;
#pragma unroll
        for (int s = 0; s < NS; ++s) m = bits[s] > m ? bits[s] : m;
        thr = wave_umax(m);
        // among the members at the largest distance, the one with the largest index
        unsigned mi = 0u;
#pragma unroll
        for (int s = 0; s < NS; ++s) if (member(s, lane) && bits[s] == thr) mi = (unsigned)idx[s] > mi ? (unsigned)idx[s] : mi;
        mi = wave_umax(mi);
        tl = 0; ts = 0;
#pragma unroll
        for (int s = NS - 1; s >= 0; --s) {
            const unsigned long long mm = __ballot(member(s, lane) && bits[s] == thr && (unsigned)idx[s] == mi);
            if (mm) { ts = s; tl = __builtin_ctzll(mm); }
        }
        thr_idx = mi;
        thr_hi = __float_as_uint(__uint_as_float(thr) * 1.015625f);
    }
    __device__ __forceinline__ bool may_enter(float d2_fast) const { return __ballot(__float_as_uint(d2_fast) <= thr_hi) != 0; }
    __device__ __forceinline__ void offer(float d2, int pidx, int lane) {
        unsigned long long m = __ballot(__float_as_uint(d2) <= thr && __float_as_uint(d2) < INF_BITS);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const unsigned cb = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(d2), src);
            const int ci = __builtin_amdgcn_readlane(pidx, src);
            if (cb < thr || (cb == thr && (unsigned)ci < thr_idx)) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const bool here = lane == tl && s == ts;
                    bits[s] = here ? cb : bits[s];
                    idx[s] = here ? ci : idx[s];
                }
                refresh(lane);
            }
        }
    }
};

template <int NS>
__global__ __launch_bounds__(256) void ray_knn_wide_kernel(const float4* __restrict__ stream, int P, const float* rec, long R, int k,
                                                           int* __restrict__ out_idx, float* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    cfloat* p = (cfloat*)rec + r * 8;
    RayK rk;
    rk.ox = p[0]; rk.oy = p[1]; rk.oz = p[2]; rk.dx = p[3]; rk.dy = p[4]; rk.dz = p[5]; rk.den = p[6]; rk.rcp = p[7];
    KSetW<NS> ks;
    ks.k = k;
#pragma unroll
    for (int s = 0; s < NS; ++s) { ks.bits[s] = ks.member(s, lane) ? INF_BITS : 0u; ks.idx[s] = -1; }
    ks.thr = INF_BITS; ks.thr_idx = 0xffffffffu; ks.thr_hi = INF_BITS; ks.tl = 0; ks.ts = 0;
    for (int base = 0; base < P; base += 64) {
        const int pi = base + lane;
        const bool ok = pi < P;
        const float4 v = stream[ok ? pi : P - 1];
        if (ks.may_enter(ok ? ray_dist2_fast(rk, v.x, v.y, v.z) : INFINITY)) {
            const float d2 = ray_dist2(rk, v.x, v.y, v.z);
            ks.offer(ok ? d2 : INFINITY, __float_as_int(v.w), lane);
        }
    }
    // rank every member by (distance, index) and write the set in ascending order
    int rank[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) rank[s] = 0;
#pragma unroll
    for (int sj = 0; sj < NS; ++sj) {
        const int nj = k - 64 * sj < 64 ? k - 64 * sj : 64;
        for (int j = 0; j < nj; ++j) {
            const unsigned dj = (unsigned)__builtin_amdgcn_readlane((int)ks.bits[sj], j);
            const int ij = __builtin_amdgcn_readlane(ks.idx[sj], j);
#pragma unroll
            for (int s = 0; s < NS; ++s) rank[s] += (dj < ks.bits[s] || (dj == ks.bits[s] && (ij < ks.idx[s] || (ij == ks.idx[s] && 64 * sj + j < 64 * s + lane)))) ? 1 : 0;      // (equal entries: slots nothing entered -- by position)
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (ks.member(s, lane)) {
            out_idx[r * k + rank[s]] = ks.idx[s] < 0 ? 0 : ks.idx[s];
            if (out_dist) out_dist[r * k + rank[s]] = sqrtf(__uint_as_float(ks.bits[s]));
        }
}

// ================================================================================================
// The spatial form (P >= KNN_BLOCKS_MIN_P): the cloud is binned on a 16^3 grid over its bounding box, laid out in Morton order of
// the cells and cut into blocks of 64 consecutive points with a bounding sphere each (three small kernels per launch:
// the points move every training step; knn_count / knn_place / knn_bounds).  A ray then tests the SPHERES first -- 64 blocks per instruction, one per lane -- and only
// walks the blocks that can still hold a point nearer than its current k-th distance: ~17 of 156 blocks at P = 10,000, ~23 of 469
// at 30,000, instead of every point.  Exact all the same: a block is skipped only if  dist(centre, line) - radius - slack  is not
// below the k-th distance, where the slack covers the rounding of both evaluations; the points of a visited block get the
// reference-ordered distance (ray_dist2) as before.
//
// A wave owns a run of T consecutive rays (neighbouring pixels) and works through them one after the other; ray i + 1 starts from
// ray i's neighbour set (exact distances recomputed), so only the first ray of a run starts cold -- and that one walks the blocks
// its line pierces first, which brings the k-th distance down before the rest is looked at.
//
// Ties: the set is the k smallest by (distance, index) -- a total order, so the result does not depend on the order in which the
// points are met (the binning places the points of a cell with LDS atomics, in no particular order) nor on the seed.

constexpr int KNN_CELLS = 16;                   // per axis

__device__ __forceinline__ unsigned spread4(unsigned v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6); }

__device__ __forceinline__ float shfl_xor_f(float v, int m) { return __shfl_xor(v, m, 64); }

constexpr int KNN_NCELL = KNN_CELLS * KNN_CELLS * KNN_CELLS;
constexpr int KNN_SAMPLE = 1024;

// The grid's box: the bounding box of a fixed sample of 1,024 points, found by every workgroup for itself (12 KB out of L2, the same
// reduction everywhere, so every workgroup of every kernel lands on the same bits).  Points outside it are clamped into the border
// cells -- the grid only has to group neighbours, the spheres are taken from the points themselves.
struct CellGrid {
    float org[3], inv[3];
    __device__ __forceinline__ unsigned cell_of(float x, float y, float z) const {
        const unsigned cx = (unsigned)fminf(fmaxf((x - org[0]) * inv[0], 0.f), (float)(KNN_CELLS - 1));      // (NaN -> 0)
        const unsigned cy = (unsigned)fminf(fmaxf((y - org[1]) * inv[1], 0.f), (float)(KNN_CELLS - 1));
        const unsigned cz = (unsigned)fminf(fmaxf((z - org[2]) * inv[2], 0.f), (float)(KNN_CELLS - 1));
        return spread4(cx) | (spread4(cy) << 1) | (spread4(cz) << 2);
    }
};

// (1,024 threads)
__device__ __forceinline__ CellGrid sample_grid(const float* __restrict__ points, int P, float (*red)[16], float* bb) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long step = P > KNN_SAMPLE ? P / KNN_SAMPLE : 1;
    const long i = (long)tid * step;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < P) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = points[i * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }      // (a NaN drops out)
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { lo[a] = fminf(lo[a], shfl_xor_f(lo[a], m)); hi[a] = fmaxf(hi[a], shfl_xor_f(hi[a], m)); }
        if (lane == 0) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    if (tid < 6) {
        float v = red[tid][0];
        for (int w = 1; w < 16; ++w) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
        bb[tid] = v;
    }
    __syncthreads();
    CellGrid g;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g.org[a] = bb[a];
        const float ext = bb[3 + a] - bb[a];
        g.inv[a] = ext > 0.f && ext < INFINITY ? (float)KNN_CELLS / ext : 0.f;
        if (!(g.org[a] > -INFINITY && g.org[a] < INFINITY)) { g.org[a] = 0.f; g.inv[a] = 0.f; }
    }
    return g;
}

// points per cell (counts[] zeroed by pack_rays_kernel)
__global__ __launch_bounds__(1024) void knn_count_kernel(const float* __restrict__ points, int P, unsigned* __restrict__ counts) {
    __shared__ float red[6][16];
    __shared__ float bb[6];
    const CellGrid g = sample_grid(points, P, red, bb);
    const long i = (long)blockIdx.x * 1024 + threadIdx.x;
    if (i < P) atomicAdd(&counts[g.cell_of(points[i * 3], points[i * 3 + 1], points[i * 3 + 2])], 1u);
}

// every point to its place in the Morton-ordered stream: offset of its cell + a ticket from the cell's cursor (cursors[] zeroed by
// pack_rays_kernel; the order inside a cell is whatever the atomics give).  Blocks are ALIGNED: the cells sharing the first `bits` bits
// of the Morton key form a group, every group starts at a multiple of 64 (the gap is padding, pre-filled by pack_rays_kernel), so no
// block straddles two groups -- cut every 64 points regardless, a tenth of the blocks spanned a jump of the curve with spheres
// half the scene wide, and every ray visited them (visited blocks per ray 32 -> 19 at P = 10,000, 55 -> 21 at 30,000; bits is chosen
// so that a group holds ~64 points).  The offsets (two prefix sums over 4,096 values) are redone by every workgroup.
__global__ __launch_bounds__(1024) void knn_place_kernel(const float* __restrict__ points, int P, const unsigned* __restrict__ counts,
                                                         unsigned* __restrict__ cursors, float4* __restrict__ stream, int bits,
                                                         unsigned* __restrict__ nblk_out) {
    __shared__ float red[6][16];
    __shared__ float bb[6];
    __shared__ unsigned ex[KNN_NCELL];      // points in front of a cell
    __shared__ unsigned gb[KNN_NCELL];      // padded start of a group
    __shared__ unsigned wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const CellGrid g = sample_grid(points, P, red, bb);
    // exclusive prefix over 4,096 values, four per thread: a wave scan, the sixteen wave totals
    auto scan4 = [&](unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned* out) {
        const unsigned mine = c0 + c1 + c2 + c3;
        unsigned incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        __syncthreads();                            // (wtot free again)
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; ++w) base += wtot[w];
        const unsigned e = base + incl - mine;
        out[4 * tid] = e; out[4 * tid + 1] = e + c0; out[4 * tid + 2] = e + c0 + c1; out[4 * tid + 3] = e + c0 + c1 + c2;
        __syncthreads();
    };
    scan4(counts[4 * tid], counts[4 * tid + 1], counts[4 * tid + 2], counts[4 * tid + 3], ex);
    const int shift = 12 - bits, G = 1 << bits;
    unsigned sz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int gi = 4 * tid + u;
        unsigned n = 0;
        if (gi < G) {
            const int c_end = (gi + 1) << shift;
            n = (c_end < KNN_NCELL ? ex[c_end] : (unsigned)P) - ex[gi << shift];
        }
        sz[u] = (n + 63u) & ~63u;
    }
    scan4(sz[0], sz[1], sz[2], sz[3], gb);
    if (blockIdx.x == 0 && tid == 1023) *nblk_out = (gb[KNN_NCELL - 1] + sz[3]) / 64u;      // (groups beyond G are empty)
    const long i = (long)blockIdx.x * 1024 + tid;
    if (i < P) {
        const float x = points[i * 3], y = points[i * 3 + 1], z = points[i * 3 + 2];
        const unsigned c = g.cell_of(x, y, z);
        const unsigned gi = c >> shift;
        const unsigned pos = gb[gi] + (ex[c] - ex[gi << shift]) + atomicAdd(&cursors[c], 1u);
        stream[pos] = make_float4(x, y, z, __int_as_float((int)i));
    }
}

// one wave per block of 64 stream positions: bounding sphere (centre of the block's box, largest distance to it, rounded outwards)
__global__ __launch_bounds__(256) void knn_bounds_kernel(const float4* __restrict__ stream, const unsigned* __restrict__ nblk_p, float4* __restrict__ blk) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= (int)*nblk_p) return;
    float4 v = stream[b * 64 + lane];
    if (__float_as_int(v.w) < 0) v = stream[b * 64];       // (padding; a block's first place always holds a point)
    float l[3] = {v.x, v.y, v.z}, h[3] = {v.x, v.y, v.z};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { l[a] = fminf(l[a], shfl_xor_f(l[a], m)); h[a] = fmaxf(h[a], shfl_xor_f(h[a], m)); }
    const float cx = 0.5f * (l[0] + h[0]), cy = 0.5f * (l[1] + h[1]), cz = 0.5f * (l[2] + h[2]);
    const float dx = v.x - cx, dy = v.y - cy, dz = v.z - cz;
    float d2 = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) d2 = fmaxf(d2, shfl_xor_f(d2, m));
    // (NaN / infinite coordinates: an infinite radius -- the block is always visited and the exact evaluation decides; fmaxf drops a NaN, so test the sum too)
    float r = sqrtf(d2) * 1.000002f + 1e-30f;
    const float chk = (l[0] + h[0]) + (l[1] + h[1]) + (l[2] + h[2]);
    float any_nan = (v.x == v.x && v.y == v.y && v.z == v.z) ? 0.f : 1.f;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) any_nan = fmaxf(any_nan, shfl_xor_f(any_nan, m));
    const bool bad = !(r < INFINITY) || !(chk == chk) || !(chk > -INFINITY && chk < INFINITY) || any_nan != 0.f;
    if (lane == 0) blk[b] = bad ? make_float4(0.f, 0.f, 0.f, INFINITY) : make_float4(cx, cy, cz, r);
}

// One ray's k-set under the total order (distance bits, index), kept SORTED across lanes 0 .. k-1 (ascending; entries that are not
// filled yet are (inf, -1) and count as the largest): the k-th distance is lane k-1's, an insertion is one whole-wave shift by a lane
// (DPP wave_shr:1) behind the candidate's place -- no maximum to look for again, no ranking at the end.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }

struct KSetT {
    unsigned bits;   // per lane
    int idx;         // per lane; -1 = not filled yet (counts as the largest index)
    unsigned thr;    // uniform: distance bits of the largest member
    unsigned thr_idx;// uniform: its index
    int k;
#ifdef KNN_DEBUG_VISITS
    int n_ins = 0, n_off = 0;
#endif
    __device__ __forceinline__ void refresh() {
        thr = (unsigned)__builtin_amdgcn_readlane((int)bits, k - 1);
        thr_idx = (unsigned)__builtin_amdgcn_readlane(idx, k - 1);
    }
    // lanes < k hold k entries in any order -> ascending (k rounds of rank counting, one forward permute)
    __device__ __forceinline__ void sort(int lane) {
        int rank = 0;
        for (int j = 0; j < k; ++j) {
            const unsigned bj = (unsigned)__builtin_amdgcn_readlane((int)bits, j);
            const unsigned ij = (unsigned)__builtin_amdgcn_readlane(idx, j);
            rank += (bj < bits || (bj == bits && ij < (unsigned)idx)) ? 1 : 0;
        }
        const int to = (lane < k ? rank : lane) * 4;
        bits = (unsigned)__builtin_amdgcn_ds_permute(to, (int)bits);
        idx = __builtin_amdgcn_ds_permute(to, idx);
        refresh();
    }
    // DEDUP: the set was started from a neighbour ray's members, which the stream brings up again
    // (tried: the members' stream positions kept in the set and masked out of a block's candidates up front -- a ballot per visited block
    // and a third register to shift per insertion cost what the twenty short trips through the loop cost: 158 -> 166 us)
    template <bool DEDUP>
    __device__ __forceinline__ void offer(bool ok, float d2, int pidx, int lane) {
        unsigned long long m = __ballot(ok && __float_as_uint(d2) <= thr);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const unsigned cb = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(d2), src);
            const int ci = __builtin_amdgcn_readlane(pidx, src);
#ifdef KNN_DEBUG_VISITS
            ++n_off;
#endif
            if (cb < thr || (cb == thr && (unsigned)ci < thr_idx)) {
                if (DEDUP && __ballot(idx == ci) != 0) continue;
#ifdef KNN_DEBUG_VISITS
                ++n_ins;
#endif
                // the members below the candidate stay, the others move up a lane, the last one leaves
                const int at = __builtin_popcountll(__ballot(lane < k && (bits < cb || (bits == cb && (unsigned)idx < (unsigned)ci))));
                const unsigned sb = (unsigned)dpp_mov<0x138>((int)bits, (int)bits);
                const int si = dpp_mov<0x138>(idx, idx);
                const bool up = lane > at && lane < k;
                bits = lane == at ? cb : (up ? sb : bits);
                idx = lane == at ? ci : (up ? si : idx);
                refresh();
            }
        }
    }
};

// lower bound (squared) of the reference distance of any point inside the sphere B to the ray, 0 if the line may touch the sphere.
// The reference's D = v - d (v.d)/(d.d + eps) is never shorter than the perpendicular, so the perpendicular with the true 1/|d|^2 is
// a valid bound; slack: the rounding of this evaluation and of ray_dist2 (a few eps |v| each) and of the radius.
__device__ __forceinline__ float block_need2(const RayK& c, float rcp_true, float4 B) {
    const float vx = B.x - c.ox, vy = B.y - c.oy, vz = B.z - c.oz;
    const float tt = __builtin_fmaf(vz, c.dz, __builtin_fmaf(vy, c.dy, vx * c.dx)) * rcp_true;
    const float ex = __builtin_fmaf(-c.dx, tt, vx), ey = __builtin_fmaf(-c.dy, tt, vy), ez = __builtin_fmaf(-c.dz, tt, vz);
    const float dl = __builtin_sqrtf(__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)));
    const float slack = 4e-6f * ((fabsf(vx) + fabsf(vy)) + (fabsf(vz) + B.w));
    const float need = (dl - B.w) - slack;
    return need > 0.f ? need * need * 0.999999f : 0.f;       // (NaN -> 0: visit)
}

__global__ __launch_bounds__(256) void ray_knn_blocks_kernel(const float4* __restrict__ stream, const float4* __restrict__ blk, const unsigned* nblk_p,
                                                             const float* __restrict__ points, int P, const float* rec, long R, int k, int T,
                                                             int* __restrict__ out_idx, float* __restrict__ out_dist) {
    const int nblk = __builtin_amdgcn_readfirstlane((int)*nblk_p);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long tile = (long)blockIdx.x * 4 + wave;
    const long r0 = tile * T;
    if (r0 >= R) return;
    cfloat* crec = (cfloat*)rec;
    KSetT ks;
    ks.k = k;
#ifdef KNN_DEBUG_VISITS
    int dbg_visits = 0;
#endif
    ks.bits = lane < k ? INF_BITS : 0u; ks.idx = -1; ks.thr = INF_BITS; ks.thr_idx = 0xffffffffu;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const long r = r0 + t;
        if (r >= R) break;
        RayK rk;
        {
            cfloat* q = crec + r * 8;
            rk.ox = q[0]; rk.oy = q[1]; rk.oz = q[2]; rk.dx = q[3]; rk.dy = q[4]; rk.dz = q[5]; rk.den = q[6]; rk.rcp = q[7];
        }
        const float dd = __builtin_fmaf(rk.dz, rk.dz, __builtin_fmaf(rk.dy, rk.dy, rk.dx * rk.dx));
        const float rcp_true = dd > 0.f ? 1.0f / dd : 0.f;
        const bool cold = t == 0;
        if (!cold) {                                // the previous ray's neighbours, at this ray's distances
            const int pi = (lane < k && ks.idx >= 0) ? ks.idx : 0;       // (a slot nothing entered -- NaN coordinates -- holds -1)
            const float d2 = ray_dist2(rk, points[pi * 3 + 0], points[pi * 3 + 1], points[pi * 3 + 2]);
            ks.bits = lane < k ? __float_as_uint(d2) : 0u;
            ks.sort(lane);
        }
        // pass 0 (cold start only): the blocks the line pierces; pass 1: every block that can still hold a nearer point.  The next
        // block's points are requested before the current one's are looked at.
        for (int pass = cold ? 0 : 1; pass < 2; ++pass) {
            float4 B = blk[lane < nblk ? lane : nblk - 1];
            for (int c0 = 0; c0 < nblk; c0 += 64) {
                const float need2 = c0 + lane < nblk ? block_need2(rk, rcp_true, B) : INFINITY;
                if (c0 + 64 < nblk) B = blk[c0 + 64 + lane < nblk ? c0 + 64 + lane : nblk - 1];
                unsigned long long m;
                if (pass == 0) m = __ballot(need2 == 0.f);
                else m = __ballot(need2 < __uint_as_float(ks.thr) && !(cold && need2 == 0.f));
                // the walk is bound by the latency of the blocks' loads (five waves per SIMD, each waiting for its one block): take the chunk's
                // blocks eight at a time -- eight loads in flight, then the eight evaluations; a block whose bound the k-th distance has passed
                // in the meantime costs its load only
                constexpr int D = 8;
                while (m) {
                    int ss[D];
                    float4 vv[D];
#pragma unroll
                    for (int u = 0; u < D; ++u) {
                        ss[u] = m ? __builtin_ctzll(m) : -1;
                        m &= m - 1;                 // (0 stays 0)
                        if (ss[u] >= 0) vv[u] = stream[(c0 + ss[u]) * 64 + lane];       // (the stream is padded to whole blocks)
                    }
#pragma unroll
                    for (int u = 0; u < D; ++u) {
                        if (ss[u] < 0) break;
                        if (pass == 1 && !(read_lane(need2, ss[u]) < __uint_as_float(ks.thr))) continue;
#ifdef KNN_DEBUG_VISITS
                        ++dbg_visits;
#endif
                        const float d2 = ray_dist2(rk, vv[u].x, vv[u].y, vv[u].z);
                        const int pidx = __float_as_int(vv[u].w);       // (< 0: padding)
                        if (cold) ks.template offer<false>(pidx >= 0, d2, pidx, lane);
                        else ks.template offer<true>(pidx >= 0, d2, pidx, lane);
                    }
                }
            }
        }
        if (lane < k) {                             // (the set is ascending in (distance, index))
            out_idx[r * k + lane] = ks.idx < 0 ? 0 : ks.idx;
            if (out_dist) out_dist[r * k + lane] = sqrtf(__uint_as_float(ks.bits));
        }
#ifdef KNN_DEBUG_VISITS
        if (out_dist && lane == 0) { out_dist[r * k + 0] = (float)dbg_visits; out_dist[r * k + 1] = (float)ks.n_ins; out_dist[r * k + 2] = (float)ks.n_off; }
        dbg_visits = 0; ks.n_ins = 0; ks.n_off = 0;
#endif
    }
}

int coprime_stride(long n) {
    long mul = (long)(n * 0.6180339887) | 1;
    auto gcd = [](long a, long b) { while (b) { long t = a % b; a = b; b = t; } return a; };
    while (gcd(mul, n) != 1) mul += 2;
    return (int)(mul % n);
}

}  // namespace

constexpr int64_t KNN_BLOCKS_MIN_P = 2048;      // below: every point against every ray (nothing to skip in a dozen blocks)

// Morton prefix bits of a group: ~64 points per group
static int knn_group_bits(int64_t P) {
    int bits = 3;
    while (bits < 12 && (double)P / (double)(1 << bits) > 90.5) ++bits;       // (64 sqrt 2)
    return bits;
}
// places of the aligned stream: every group may end with up to 63 places of padding
static int64_t knn_stream_places(int64_t P) { return (P + 63 * ((int64_t)1 << knn_group_bits(P)) + 127) / 64 * 64; }

extern "C" size_t papr_ray_knn_workspace_bytes(int64_t R, int64_t P) {
    // ray records | point stream, padded to whole 64-point blocks | one bounding sphere per block | cell counters and cursors
    return (size_t)R * 8 * sizeof(float) + (size_t)knn_stream_places(P) * 4 * sizeof(float) + (size_t)(knn_stream_places(P) / 64) * 4 * sizeof(float) +
           (size_t)(2 * KNN_NCELL + 4) * sizeof(unsigned);
}

extern "C" int papr_ray_knn(const float* points, int64_t P, const float* rays_o, const float* rays_d,
                            int64_t R, int64_t rays_per_image, int k, float eps, int32_t* out_idx,
                            float* out_dist, void* workspace, papr_stream_t stream) {
    PAPR_REQUIRE(points && rays_o && rays_d && out_idx && workspace, "papr_ray_knn: null pointer");
    PAPR_REQUIRE(k >= 1 && k <= 256, "papr_ray_knn: k=%d outside [1,256]", k);
    PAPR_REQUIRE(P >= k && P * 3 < (int64_t)1 << 31, "papr_ray_knn: P=%lld must be >= k and < 2^31/3", (long long)P);
    PAPR_REQUIRE(rays_per_image >= 1, "papr_ray_knn: rays_per_image must be positive");
    if (R <= 0) return 0;
    hipStream_t s = as_stream(stream);
    float* rec = static_cast<float*>(workspace);
    float4* pstream = reinterpret_cast<float4*>(rec + (size_t)R * 8);
    const int blocks_env = papr_switch(PAPR_SW_KNN_BLOCKS);      // (A/B switch: 0 = every point against every ray)
    const bool spatial = blocks_env && P >= KNN_BLOCKS_MIN_P && k < 64;      // (k = 64: the every-point form; k > 64: the wide form)
    const int64_t places = knn_stream_places(P);
    float4* blk = pstream + (size_t)places;
    unsigned* counts = reinterpret_cast<unsigned*>(blk + (size_t)(places / 64));
    pack_rays_kernel<<<dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s>>>(rays_o, rays_d, R, rays_per_image, eps, rec, counts, spatial ? 2 * KNN_NCELL + 4 : 0,
                                                                             pstream, spatial ? places : 0);
    PAPR_CHECK_LAUNCH("pack_rays");
    const int n_simd = 4 * papr_cu_count();
    const int t_env = papr_switch(PAPR_SW_KNN_T);      // (A/B switch)
    if (spatial) {
        const unsigned pg = (unsigned)((P + 1023) / 1024);
        unsigned* nblk_p = counts + 2 * KNN_NCELL;
        knn_count_kernel<<<dim3(pg), dim3(1024), 0, s>>>(points, (int)P, counts);
        knn_place_kernel<<<dim3(pg), dim3(1024), 0, s>>>(points, (int)P, counts, counts + KNN_NCELL, pstream, knn_group_bits(P), nblk_p);
        knn_bounds_kernel<<<dim3((unsigned)((places / 64 + 3) / 4)), dim3(256), 0, s>>>(pstream, nblk_p, blk);
        PAPR_CHECK_LAUNCH("knn_bin_points");
        // rays per wave.  The kernel is bound by instruction issue (3.5k instructions per ray, SQ_INSTS_*): a run costs its cold first ray
        // (~2.8 warm rays' worth) + T - 1 warm ones, and the launch takes as long as the SIMD with the most runs; but with fewer than four
        // waves per SIMD the loads' latency shows (T = 25, one run per SIMD: 396 us against 158), so T stays <= R / (4 SIMDs).
        int T = 2;
        double best = 1e300;
        const int t_max = (int)std::max<long>(2, std::min<long>(16, R / (4L * n_simd)));
        for (int t = 2; t <= t_max; ++t) {
            const long tl = (R + t - 1) / t;
            const double cost = (double)((tl + n_simd - 1) / n_simd) * (1.8 + t);
            if (cost < best - 1e-9) { best = cost; T = t; }
        }
        if (t_env >= 1 && t_env <= 64) T = t_env;
        const long tiles = (R + T - 1) / T;
        const bool prof = papr_prof_on();
        if (prof) papr_prof_begin(5, R, (int)P, k, s);
        ray_knn_blocks_kernel<<<dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s>>>(pstream, blk, nblk_p, points, (int)P, rec, R, k, T, out_idx, out_dist);
        if (prof) papr_prof_end(s);
        PAPR_CHECK_LAUNCH("ray_knn_blocks");
        return 0;
    }
    scatter_points_kernel<<<dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s>>>(points, (int)P, P > 1 ? coprime_stride(P) : 0, pstream);
    PAPR_CHECK_LAUNCH("scatter_points");
    if (k > 64) {                                   // the set across several registers per lane (ray_knn_wide_kernel)
        const dim3 grid((unsigned)((R + 3) / 4)), block(256);
        const bool prof = papr_prof_on();
        if (prof) papr_prof_begin(5, R, (int)P, k, s);
        if (k <= 128) ray_knn_wide_kernel<2><<<grid, block, 0, s>>>(pstream, (int)P, rec, R, k, out_idx, out_dist);
        else ray_knn_wide_kernel<4><<<grid, block, 0, s>>>(pstream, (int)P, rec, R, k, out_idx, out_dist);
        if (prof) papr_prof_end(s);
        PAPR_CHECK_LAUNCH("ray_knn_wide");
        return 0;
    }
    // Rays per wave T: a tile costs ~(1.8 + T - 1) ray passes over the cloud (its first ray starts cold, the others are seeded), and the
    // launch takes as long as the SIMD with the most tiles: ceil(tiles / SIMDs) of them.  With T = 8 a 160 x 160 patch is 3,200 tiles on
    // 1,024 SIMDs -- 4 on some, 3.125 on average, a quarter of the machine idle at the end (round 2); T = 5 makes it exactly 5 each.
    constexpr int PPL = 4;
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
