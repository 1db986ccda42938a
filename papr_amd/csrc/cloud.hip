// K6: point-vs-points k nearest neighbours of the cloud, for the growth step (add_points_knn, models/utils.py:9-109): the
// reference builds a scipy KDTree of the cloud on the host and queries it twice -- all P points with k = add_sample_k for the
// sparsity ranking, the chosen sites with k = add_k + 1 for the convex combinations.  Here the cloud never leaves the device:
// exact brute force (P <= a few 10^4: 10^9 squared distances, milliseconds), distances in double like the KDTree's (float32
// coordinates widened: differences and squares are exact, the two additions round once each), neighbours ascending in
// (distance, point index).
#include "papr_common.h"

namespace {

constexpr int CK_MAX = 16;          // neighbours per query
constexpr int CK_TILE = 256;        // cloud points staged per pass = threads per workgroup

__global__ __launch_bounds__(CK_TILE) void points_knn_kernel(const float* __restrict__ points, long P, const int* __restrict__ query_idx, long Q,
                                                             int k, int* __restrict__ nn_idx, double* __restrict__ nn_dist) {
    __shared__ float tile[CK_TILE * 3];
    const long q = (long)blockIdx.x * CK_TILE + threadIdx.x;
    const bool live = q < Q;
    const long qp = live ? (query_idx ? (long)query_idx[q] : q) : 0;
    const double qx = points[qp * 3], qy = points[qp * 3 + 1], qz = points[qp * 3 + 2];
    double bd[CK_MAX];
    int bi[CK_MAX];
#pragma unroll
    for (int i = 0; i < CK_MAX; ++i) { bd[i] = 1e300; bi[i] = -1; }
    for (long t0 = 0; t0 < P; t0 += CK_TILE) {
        const long n = P - t0 < CK_TILE ? P - t0 : CK_TILE;
        __syncthreads();
        for (int e = threadIdx.x; e < n * 3; e += CK_TILE) tile[e] = points[t0 * 3 + e];
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < n; ++j) {
            const double dx = qx - (double)tile[3 * j], dy = qy - (double)tile[3 * j + 1], dz = qz - (double)tile[3 * j + 2];
            double cd = dx * dx + dy * dy + dz * dz;
            if (cd < bd[CK_MAX - 1] || k < CK_MAX) {            // (lists shorter than CK_MAX keep their unused tail at 1e300)
                int ci = (int)(t0 + j);
#pragma unroll
                for (int i = 0; i < CK_MAX; ++i) {              // one pass through the sorted list, carrying the displaced entry
                    const bool sw = i < k && cd < bd[i];        // (strict: among equal distances the lower index stays in front)
                    const double td = sw ? bd[i] : cd;
                    const int ti = sw ? bi[i] : ci;
                    bd[i] = sw ? cd : bd[i];
                    bi[i] = sw ? ci : bi[i];
                    cd = td; ci = ti;
                }
            }
        }
    }
    if (live)
        for (int i = 0; i < k; ++i) {
            nn_idx[q * k + i] = bi[i];
            if (nn_dist) nn_dist[q * k + i] = sqrt(bd[i]);
        }
}

}  // namespace

extern "C" int papr_points_knn(const float* points, int64_t P, const int32_t* query_idx, int64_t Q, int32_t k, int32_t* nn_idx,
                               double* nn_dist, papr_stream_t stream) {
    PAPR_REQUIRE(points && nn_idx && P >= 1 && Q >= 0, "papr_points_knn: P = %ld points, Q = %ld queries", (long)P, (long)Q);
    PAPR_REQUIRE(k >= 1 && k <= CK_MAX && k <= P, "papr_points_knn: 1 <= k = %d <= min(%d, P = %ld)", (int)k, CK_MAX, (long)P);
    if (Q == 0) return 0;
    points_knn_kernel<<<dim3((unsigned)((Q + CK_TILE - 1) / CK_TILE)), dim3(CK_TILE), 0, as_stream(stream)>>>(points, P, query_idx, Q, k, nn_idx, nn_dist);
    PAPR_CHECK_LAUNCH("points_knn");
    return 0;
}
