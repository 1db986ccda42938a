// Grouping of the (ray, neighbour) pairs by selected point, for papr_segment_reduce (features.hip): the atomic-free
// backward of the three gathers of the reference (models/model.py:330,435,509: index_put_(accumulate=True) in autograd).
//
// Round 1 did this on the host side with torch.sort(stable) + torch.bincount + torch.cumsum (five launches and a
// 64-bit key sort).  Here: one stable LSD radix sort over only the bits a point index can have (rocPRIM's device
// radix sort: the platform primitive, tuned per architecture -- nothing in PAPR's arithmetic depends on how the
// permutation is found, only on its being THE stable one), an iota kernel in front and one kernel behind that turns
// the sorted keys into the P + 1 group bounds.
#include "papr_common.h"
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

__global__ __launch_bounds__(256) void iota_kernel(long* __restrict__ v, long M) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < M) v[i] = i;
}

// seg[p] = first position whose key is >= p (p = 0 .. P): one thread per bound, a binary search in the sorted keys (the first
// version -- one thread per entry, each closing the groups between its neighbour's key and its own -- took 100 us at M = 512,000)
__global__ __launch_bounds__(256) void group_bounds_kernel(const int* __restrict__ sorted_pts, long M, long P, long* __restrict__ seg) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p > P) return;
    long lo = 0, hi = M;                            // first i in [0, M] with sorted_pts[i] >= p
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (sorted_pts[mid] < p) lo = mid + 1; else hi = mid;
    }
    seg[p] = lo;
}

int key_bits(long P) {
    int b = 1;
    while (b < 31 && (1L << b) < P) ++b;
    return b;
}

}  // namespace

extern "C" size_t papr_group_pairs_workspace_bytes(int64_t M, int64_t P) {
    if (M <= 0) return 256;
    size_t temp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, temp, static_cast<const int*>(nullptr), static_cast<int*>(nullptr), static_cast<const long*>(nullptr),
                                    static_cast<long*>(nullptr), (size_t)M, 0u, (unsigned)key_bits(P), (hipStream_t)0);
    return (temp + 255) / 256 * 256 + (size_t)M * sizeof(long);          // sort scratch | the iota values
}

extern "C" int papr_group_pairs(const int32_t* idx, int64_t M, int64_t P, int64_t* order, int32_t* sorted_pts, int64_t* seg,
                                void* workspace, size_t workspace_bytes, papr_stream_t stream) {
    PAPR_REQUIRE(P >= 1 && M >= 0, "papr_group_pairs: M = %ld pairs, P = %ld points", (long)M, (long)P);
    PAPR_REQUIRE(seg && (M == 0 || (idx && order && sorted_pts)), "papr_group_pairs: null pointer");
    hipStream_t s = as_stream(stream);
    if (M == 0) {
        PAPR_REQUIRE(hipMemsetAsync(seg, 0, (size_t)(P + 1) * sizeof(int64_t), s) == hipSuccess, "papr_group_pairs: memset failed");
        return 0;
    }
    const size_t need = papr_group_pairs_workspace_bytes(M, P);
    PAPR_REQUIRE(workspace && workspace_bytes >= need, "papr_group_pairs: workspace of %zu bytes, %zu needed", workspace_bytes, need);
    size_t temp = need - (size_t)M * sizeof(long);
    long* iota = reinterpret_cast<long*>(static_cast<char*>(workspace) + temp);
    iota_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s>>>(iota, M);
    PAPR_REQUIRE(rocprim::radix_sort_pairs(workspace, temp, idx, sorted_pts, iota, reinterpret_cast<long*>(order), (size_t)M, 0u,
                                           (unsigned)key_bits(P), s) == hipSuccess, "papr_group_pairs: radix sort failed");
    group_bounds_kernel<<<dim3((unsigned)((P + 256) / 256)), dim3(256), 0, s>>>(sorted_pts, M, P, reinterpret_cast<long*>(seg));
    PAPR_CHECK_LAUNCH("group_pairs");
    return 0;
}
