// Grouping of the (ray, neighbour) pairs by selected point, for papr_segment_reduce (features.hip): the atomic-free
// backward of the three gathers of the reference (models/model.py:330,435,509: index_put_(accumulate=True) in autograd).
//
// Round 1 did this on the host side with torch.sort(stable) + torch.bincount + torch.cumsum (five launches and a 64-bit key sort),
// rounds 2-3 with rocPRIM's LSD radix sort (18 launches, 0.12 ms per step).  Round 4: a stable COUNTING sort written for the shape of
// the problem, four launches and ~25 us.  What makes it cheap is a property of the input that the caller states (`run`): every aligned
// run of `run` consecutive entries holds DISTINCT points -- a ray's k neighbours.  The pairs are dealt to 256 one-wave workgroups in
// contiguous chunks of whole runs:
//   count     each workgroup's histogram of its chunk over the points (LDS, integer atomics: order-free)        -> hist[b][p]
//   offsets   per point: the exclusive prefix of the workgroups' counts (one thread per point), the point's total
//   scan      exclusive scan of the totals                                                                      -> seg[0 .. P]
//   place     each workgroup walks its chunk run by run: entry e of point p goes to position cnt[p]++ with cnt[p] starting at
//             seg[p] + (this workgroup's prefix).  One returning LDS add per run -- the entries of a run are distinct points, so no two
//             lanes of the instruction meet on one counter, and the LDS executes a wave's instructions in order: the permutation is
//             THE stable one (pair ids ascending inside a group), the same on every run, with no sort and no float anywhere.
#include "papr_common.h"
#include <string.h>

namespace {

constexpr int GP_BLOCKS = 256;                  // workgroups = chunks of the pair list
constexpr int GP_THREADS = 256;                 // threads that fill and drain a workgroup's LDS tables (the placing itself is ONE wave's job: it is a sequence)
constexpr int GP_BINS = 36864;                  // points per pass: 144 KB of LDS counters (+ 16 KB of staged entries = 160 KB; beyond that: several passes over the chunk)

struct GPArgs {
    const int* idx; long M, P; int run; long chunk;     // chunk: entries per workgroup, a multiple of run
    unsigned* hist;                                 // [GP_BLOCKS][P]
    unsigned* total;                                // [P]
    long* order; int* sorted_pts; long* seg;
};

__global__ __launch_bounds__(GP_THREADS) void pairs_count_kernel(GPArgs a) {
    extern __shared__ unsigned cnt[];
    const long beg = (long)blockIdx.x * a.chunk, end = beg + a.chunk < a.M ? beg + a.chunk : a.M;
    for (long lo = 0; lo < a.P; lo += GP_BINS) {
        const long nb = a.P - lo < GP_BINS ? a.P - lo : GP_BINS;
        for (long i = threadIdx.x; i < nb; i += GP_THREADS) cnt[i] = 0u;
        __syncthreads();
        for (long i = beg + threadIdx.x; i < end; i += GP_THREADS) {
            const long key = a.idx[i] - lo;
            if (key >= 0 && key < nb) atomicAdd(&cnt[key], 1u);
        }
        __syncthreads();
        unsigned* row = a.hist + (size_t)blockIdx.x * a.P + lo;
        for (long i = threadIdx.x; i < nb; i += GP_THREADS) row[i] = cnt[i];
        __syncthreads();
    }
}

// per point: hist[b][p] <- sum of hist[b'][p] over b' < b; total[p] = the sum over all workgroups (threads of a wave read consecutive points)
__global__ __launch_bounds__(256) void pairs_offsets_kernel(GPArgs a) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.P) return;
    unsigned s = 0u;
    for (int b0 = 0; b0 < GP_BLOCKS; b0 += 32) {   // 32 loads in flight, then their 32 stores (a load behind a store to the same array waits for it)
        unsigned v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = a.hist[(size_t)(b0 + j) * a.P + p];
#pragma unroll
        for (int j = 0; j < 32; ++j) { a.hist[(size_t)(b0 + j) * a.P + p] = s; s += v[j]; }
    }
    a.total[p] = s;
}

// seg[p] = number of entries whose point is < p, seg[P] = M.  One workgroup: every thread sums a contiguous span, the spans' sums are
// scanned through LDS, every thread writes its span.
__global__ __launch_bounds__(1024) void pairs_scan_kernel(GPArgs a) {
    __shared__ long part[1024];
    const long span = (a.P + 1023) / 1024;
    const long p0 = (long)threadIdx.x * span, p1 = p0 + span < a.P ? p0 + span : a.P;
    long s = 0;
    for (long p = p0; p < p1; ++p) s += a.total[p];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {            // (Hillis-Steele: 10 rounds over 1,024 values)
        const long add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    long run = part[threadIdx.x] - s;               // exclusive
    for (long p = p0; p < p1; ++p) { a.seg[p] = run; run += a.total[p]; }
    if (threadIdx.x == 1023) a.seg[a.P] = part[1023];
}

constexpr int GP_STAGE = 4096;                  // entries of the chunk staged in LDS at a time (a step's global load would otherwise sit in the serial chain)

__global__ __launch_bounds__(GP_THREADS) void pairs_place_kernel(GPArgs a) {
    extern __shared__ unsigned cnt[];
    const long beg = (long)blockIdx.x * a.chunk, end = beg + a.chunk < a.M ? beg + a.chunk : a.M;
    const int lanes = a.run < 64 ? a.run : 64;      // entries per step: one run, or 64 consecutive entries of a longer one
    for (long lo = 0; lo < a.P; lo += GP_BINS) {
        const long nb = a.P - lo < GP_BINS ? a.P - lo : GP_BINS;
        int* const keys = reinterpret_cast<int*>(cnt + nb);
        const unsigned* row = a.hist + (size_t)blockIdx.x * a.P + lo;
        for (long i = threadIdx.x; i < nb; i += GP_THREADS) cnt[i] = (unsigned)a.seg[lo + i] + row[i];
        // whole runs per stage (a run longer than the stage: 64-entry pieces of it, still in order)
        const long per_stage = a.run <= GP_STAGE ? (GP_STAGE / a.run) * (long)a.run : GP_STAGE;
        for (long g0 = beg; g0 < end; g0 += per_stage) {
            const long g1 = g0 + per_stage < end ? g0 + per_stage : end;
            __syncthreads();
            for (long i = g0 + threadIdx.x; i < g1; i += GP_THREADS) keys[i - g0] = a.idx[i];
            __syncthreads();
            if (threadIdx.x >= 64) continue;                                // (the placing is a sequence: one wave)
            // run by run, in order; a step = one run (or 64 consecutive entries of a longer one).  The next step's points are read from LDS
            // before this step's counters answer, so that a step costs one LDS round trip, not two
            long s0 = g0;
            auto step_end = [&](long b) { const long r1 = (b / a.run + 1) * (long)a.run; const long e = b + lanes < r1 ? b + lanes : r1; return e < g1 ? e : g1; };
            long e0 = step_end(s0);
            int pt = (s0 + threadIdx.x < e0) ? keys[s0 + threadIdx.x - g0] : -1;
            while (s0 < g1) {
                const long s1 = e0, e1 = s1 < g1 ? step_end(s1) : s1;
                const int pt_next = (s1 + threadIdx.x < e1) ? keys[s1 + threadIdx.x - g0] : -1;
                const long key = (long)pt - lo;
                if (pt >= 0 && key >= 0 && key < nb) {
                    const unsigned pos = atomicAdd(&cnt[key], 1u);           // (distinct keys inside the instruction: no lane meets another)
                    a.order[pos] = s0 + threadIdx.x;
                    a.sorted_pts[pos] = pt;
                }
                s0 = s1; e0 = e1; pt = pt_next;
            }
        }
        __syncthreads();
    }
}

size_t lds_bytes(long P, bool stage) { return ((size_t)(P < GP_BINS ? P : GP_BINS) + (stage ? GP_STAGE : 0)) * sizeof(unsigned); }

}  // namespace

extern "C" size_t papr_group_pairs_workspace_bytes(int64_t M, int64_t P) {
    if (M <= 0 || P <= 0) return 256;
    return ((size_t)GP_BLOCKS * (size_t)P + (size_t)P) * sizeof(unsigned) + 256;         // the workgroups' histograms | the points' totals
}

extern "C" int papr_group_pairs(const int32_t* idx, int64_t M, int64_t P, int32_t run, int64_t* order, int32_t* sorted_pts, int64_t* seg,
                                void* workspace, size_t workspace_bytes, papr_stream_t stream) {
    PAPR_REQUIRE(P >= 1 && M >= 0 && M < ((int64_t)1 << 32), "papr_group_pairs: M = %ld pairs, P = %ld points", (long)M, (long)P);
    PAPR_REQUIRE(seg && (M == 0 || (idx && order && sorted_pts)), "papr_group_pairs: null pointer");
    PAPR_REQUIRE(run >= 1, "papr_group_pairs: run = %d (entries per run of distinct points) must be positive", run);
    hipStream_t s = as_stream(stream);
    if (M == 0) {
        PAPR_REQUIRE(hipMemsetAsync(seg, 0, (size_t)(P + 1) * sizeof(int64_t), s) == hipSuccess, "papr_group_pairs: memset failed");
        return 0;
    }
    const size_t need = papr_group_pairs_workspace_bytes(M, P);
    PAPR_REQUIRE(workspace && workspace_bytes >= need, "papr_group_pairs: workspace of %zu bytes, %zu needed", workspace_bytes, need);
    GPArgs a;
    a.idx = idx; a.M = M; a.P = P; a.run = run;
    const long runs = (M + run - 1) / run;
    a.chunk = ((runs + GP_BLOCKS - 1) / GP_BLOCKS) * run;
    a.hist = static_cast<unsigned*>(workspace);
    a.total = a.hist + (size_t)GP_BLOCKS * (size_t)P;
    a.order = reinterpret_cast<long*>(order); a.sorted_pts = sorted_pts; a.seg = reinterpret_cast<long*>(seg);
    if (papr_first_on_device(PAPR_ONCE_PAIRS)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pairs_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GP_BINS * sizeof(unsigned)));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pairs_place_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((GP_BINS + GP_STAGE) * sizeof(unsigned)));
    }
    pairs_count_kernel<<<dim3(GP_BLOCKS), dim3(GP_THREADS), lds_bytes(P, false), s>>>(a);
    pairs_offsets_kernel<<<dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s>>>(a);
    pairs_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>(a);
    pairs_place_kernel<<<dim3(GP_BLOCKS), dim3(GP_THREADS), lds_bytes(P, true), s>>>(a);
    PAPR_CHECK_LAUNCH("group_pairs");
    return 0;
}
