// K1: for every ray, the k points of the cloud nearest to the ray (perpendicular distance).
//
// Replaces PAPR._calculate_global_distances (reference models/model.py:258-283), which materialises
// five R x P fp32 tensors and runs torch.topk over them.  Here nothing of size R x P ever exists.
//
// Work decomposition (wave64, one wave = one tile of T rays against ALL points):
//   * the 64 lanes of a wave each hold PPL points of the current batch in VGPRs (coalesced loads
//     of the xyz stream, 12 B/point); the batch is reused for all T rays of the tile, so the
//     point stream is read once per T rays;
//   * the ray constants (origin, direction, d.d+eps and its reciprocal) are wave-uniform and come
//     in through scalar loads (s_load -> SGPRs), so every VALU op has one VGPR and one SGPR operand;
//   * each ray's running top list lives ACROSS the lanes of one VGPR (lane j = j-th nearest so
//     far, +inf padded): a candidate test is one v_cmp + ballot against the k-th distance kept in
//     an SGPR, and an insertion is ballot/popcount + a one-lane shift.  There is no per-lane
//     divergence and no scratch: the list never leaves registers.
//
// Arithmetic follows the reference formulation operation by operation (no FMA contraction, IEEE
// division) so that near-tie neighbours resolve the same way: v = p - o; t = (v.d)/(d.d+eps);
// D = v - d t; select on |D|^2 (sqrt is monotone; exact ties are unordered in the reference too).
#include "papr_common.h"

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

__device__ __forceinline__ float lane_shift_up(float v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ int lane_shift_up(int v) { return __shfl_up(v, 1, 64); }

__device__ __forceinline__ float read_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

template <int T, int PPL>
__global__ __launch_bounds__(256) void ray_knn_kernel(const float* __restrict__ points, int P,
                                                      const float* rec, long R, int k,
                                                      int* __restrict__ out_idx,
                                                      float* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long tile = (long)blockIdx.x * 4 + wave;
    const long r0 = tile * T;
    if (r0 >= R) return;

    cfloat* crec = (cfloat*)rec;
    float ox[T], oy[T], oz[T], dx[T], dy[T], dz[T], den[T], rcp[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        long r = r0 + t < R ? r0 + t : R - 1;
        cfloat* p = crec + r * 8;
        ox[t] = p[0]; oy[t] = p[1]; oz[t] = p[2]; dx[t] = p[3];
        dy[t] = p[4]; dz[t] = p[5]; den[t] = p[6]; rcp[t] = p[7];
    }

    float bd[T];  // lane j: j-th smallest squared distance of ray t so far
    int bi[T];
    float thr[T]; // wave-uniform: current k-th smallest
#pragma unroll
    for (int t = 0; t < T; ++t) { bd[t] = INFINITY; bi[t] = -1; thr[t] = INFINITY; }

    const int km1 = k - 1;
    for (int base = 0; base < P; base += 64 * PPL) {
        float px[PPL], py[PPL], pz[PPL];
        bool ok[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            int pi = base + q * 64 + lane;
            ok[q] = pi < P;
            int pc = ok[q] ? pi : P - 1;
            px[q] = points[pc * 3 + 0]; py[q] = points[pc * 3 + 1]; pz[q] = points[pc * 3 + 2];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                float vx = px[q] - ox[t], vy = py[q] - oy[t], vz = pz[q] - oz[t];
                float vd = (vx * dx[t] + vy * dy[t]) + vz * dz[t];
                // correctly rounded vd / den from the exact reciprocal (one Newton step on the quotient)
                float q0 = vd * rcp[t];
                float rem = __builtin_fmaf(-q0, den[t], vd);
                float tt = __builtin_fmaf(rem, rcp[t], q0);
                float ex = vx - dx[t] * tt, ey = vy - dy[t] * tt, ez = vz - dz[t] * tt;
                // torch.norm accumulates its squares with an fma chain: fma(z,z, fma(y,y, x*x))
                float d2 = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                d2 = ok[q] ? d2 : INFINITY;
                unsigned long long m = __ballot(d2 < thr[t]);
                while (m) {
                    int src = __builtin_ctzll(m);
                    m &= m - 1;
                    float cd = read_lane(d2, src);
                    if (cd < thr[t]) {
                        int ci = base + q * 64 + src;
                        int pos = __popcll(__ballot(bd[t] <= cd));
                        float ud = lane_shift_up(bd[t]);
                        int ui = lane_shift_up(bi[t]);
                        bd[t] = lane < pos ? bd[t] : (lane == pos ? cd : ud);
                        bi[t] = lane < pos ? bi[t] : (lane == pos ? ci : ui);
                        thr[t] = read_lane(bd[t], km1);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        long r = r0 + t;
        if (r < R && lane < k) {
            out_idx[r * k + lane] = bi[t];
            if (out_dist) out_dist[r * k + lane] = sqrtf(bd[t]);
        }
    }
}

}  // namespace

extern "C" size_t papr_ray_knn_workspace_bytes(int64_t R) { return (size_t)R * 8 * sizeof(float); }

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
    pack_rays_kernel<<<dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s>>>(rays_o, rays_d, R, rays_per_image, eps, rec);
    PAPR_CHECK_LAUNCH("pack_rays");
    constexpr int T = 8, PPL = 4;
    long tiles = (R + T - 1) / T;
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin(5, R, (int)P, k, s);
    ray_knn_kernel<T, PPL><<<dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s>>>(points, (int)P, rec, R, k, out_idx, out_dist);
    if (prof) papr_prof_end(s);
    PAPR_CHECK_LAUNCH("ray_knn");
    return 0;
}
