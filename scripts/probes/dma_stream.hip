// LDS-DMA streaming probe for gemm_tn_tr_kernel's request pattern: two operands of M x 256 f16 rows (512 bytes each), one row slice per workgroup
// (512 threads, one workgroup per CU), stages of ROWS rows per operand, DEPTH stage buffers in LDS, a barrier per stage as in the kernel.  No matrix work.
//   PAT 0: one instruction = rows q and q + ROWS / 2 of the stage (the kernel's pairs)      PAT 1: rows 2 q and 2 q + 1 (1 KB contiguous)
//   PLAIN: the same bytes with global_load_dwordx4 into registers (two stages in flight), no LDS
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/dma_stream scripts/probes/dma_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

struct Jobs { const char* g[4]; const char* x[4]; int n; };
__device__ __forceinline__ void dma4(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
template <int ROWS, int DEPTH, int PAT, bool BARRIER, int PITCH = 1024, int MAXIMA = 0>
__global__ __launch_bounds__(512) void dma_read_jobs(Jobs jobs, long M, int* out, const float* mx = nullptr) {      // the kernel's job-parallel dealing: workgroup b = slice b / n of job b % n
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)smem;
    constexpr int STAGE = ROWS * PITCH;
    constexpr int NI = 2 * ROWS / 2 / 8;
    const int jb = blockIdx.x % jobs.n, slice = blockIdx.x / jobs.n, S = gridDim.x / jobs.n;
    const char* G = jobs.g[jb]; const char* X = jobs.x[jb];
    const long stages = M / ROWS, per = (stages + S - 1) / S;
    const long s0 = slice * per, s1 = s0 + per < stages ? s0 + per : stages;
    auto issue = [&](long st) {
        long s = st < s1 ? st : s1 - 1;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int idx = NI * wave + q, op = idx / (ROWS / 2), pair = idx % (ROWS / 2);
            const int row = PAT == 0 ? pair + (ROWS / 2) * (lane >> 5) : 2 * pair + (lane >> 5);
            const char* src = (op ? X : G) + (s * ROWS + row) * 512 + 16 * (lane & 31);
            dma16(src, lds0 + (unsigned)((st % DEPTH) * STAGE + idx * PITCH));
        }
        if (MAXIMA && wave == 0) dma4(mx + s * ROWS + (lane & 31) + (lane >> 5) * M, lds0 + (unsigned)(DEPTH * STAGE + (st % DEPTH) * 256));
    };
    for (long st = s0; st < s0 + DEPTH - 1; ++st) issue(st);
    for (long st = s0; st < s1; ++st) {
        if (MAXIMA) { if (wave == 0) wait_vm<(NI + 1) * (DEPTH - 2)>(); else wait_vm<NI * (DEPTH - 2)>(); }
        else wait_vm<NI * (DEPTH - 2)>();
        if (BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue(st + DEPTH - 1);
    }
    wait_vm<0>();
    if (smem[tid] == 123 && smem[tid + 7777] == 45) out[0] = 1;
}

template <int ROWS, int DEPTH, int PAT, bool BARRIER>
__global__ __launch_bounds__(512) void dma_read(const char* __restrict__ G, const char* __restrict__ X, long M, int* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)smem;
    constexpr int STAGE = 2 * ROWS * 512;                  // bytes per stage (both operands)
    constexpr int NI = 2 * ROWS / 2 / 8;                   // instructions per wave and stage (one instruction = two rows)
    const long stages = M / ROWS, per = (stages + gridDim.x - 1) / gridDim.x;
    const long s0 = blockIdx.x * per, s1 = s0 + per < stages ? s0 + per : stages;
    auto issue = [&](long st) {
        long s = st < s1 ? st : s1 - 1;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int idx = NI * wave + q, op = idx / (ROWS / 2), pair = idx % (ROWS / 2);
            const int row = PAT == 0 ? pair + (ROWS / 2) * (lane >> 5) : 2 * pair + (lane >> 5);
            const char* src = (op ? X : G) + (s * ROWS + row) * 512 + 16 * (lane & 31);
            dma16(src, lds0 + (unsigned)((st % DEPTH) * STAGE + idx * 1024));
        }
    };
    for (long st = s0; st < s0 + DEPTH - 1; ++st) issue(st);
    for (long st = s0; st < s1; ++st) {
        wait_vm<NI * (DEPTH - 2)>();
        if (BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue(st + DEPTH - 1);
    }
    wait_vm<0>();
    if (smem[tid] == 123 && smem[tid + 7777] == 45) out[0] = 1;
}

template <int ROWS, int LOADS>
__global__ __launch_bounds__(512) void plain_read(const char* __restrict__ G, const char* __restrict__ X, long M, int* out) {
    const int tid = threadIdx.x;
    constexpr int N4 = 2 * ROWS * 512 / 16 / 512;          // float4 per thread and stage
    const long stages = M / ROWS, per = (stages + gridDim.x - 1) / gridDim.x;
    const long s0 = blockIdx.x * per, s1 = s0 + per < stages ? s0 + per : stages;
    float4 acc = make_float4(0, 0, 0, 0);
    for (long s = s0; s < s1; s += LOADS) {
        float4 v[LOADS][N4];
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const long ss = s + u < s1 ? s + u : s;
#pragma unroll
            for (int i = 0; i < N4; ++i) {
                const int e = tid + 512 * i, op = e / (ROWS * 32);
                v[u][i] = *reinterpret_cast<const float4*>((op ? X : G) + ss * ROWS * 512 + (long)(e % (ROWS * 32)) * 16);
            }
        }
#pragma unroll
        for (int u = 0; u < LOADS; ++u)
#pragma unroll
            for (int i = 0; i < N4; ++i) { acc.x += v[u][i].x; acc.y += v[u][i].y; acc.z += v[u][i].z; acc.w += v[u][i].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1;
}

template <typename F> static float time_ms(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
    const long M = 512000;
    const size_t bytes = (size_t)M * 512;
    const int NB = 6;                                     // buffer pairs: consecutive launches must not find their rows in the 256 MB infinity cache
    std::vector<char*> xs(NB), gs(NB);
    for (int i = 0; i < NB; ++i) { hipMalloc(&xs[i], bytes); hipMalloc(&gs[i], bytes); hipMemset(xs[i], 0, bytes); hipMemset(gs[i], 0, bytes); }
    int* out; hipMalloc(&out, 64);
    int it = 0; char *X, *G;
    auto next = [&]() { X = xs[it % NB]; G = gs[it % NB]; ++it; };
    const double gb = 2.0 * bytes / 1e9;
#define DMA(ROWS, DEPTH, PAT, BAR, NWG) { const int lds = DEPTH * 2 * ROWS * 512; \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_read<ROWS, DEPTH, PAT, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        float ms = time_ms([&]() { next(); dma_read<ROWS, DEPTH, PAT, BAR><<<NWG, 512, lds>>>(G, X, M, out); }, 20); \
        printf("dma rows %2d depth %d pat %d barrier %d nwg %4d lds %6d  %.1f us  %.2f TB/s\n", ROWS, DEPTH, PAT, (int)BAR, NWG, lds, ms * 1e3, gb / ms); }
#define PLAIN(ROWS, LOADS, NWG) { float ms = time_ms([&]() { next(); plain_read<ROWS, LOADS><<<NWG, 512>>>(G, X, M, out); }, 20); \
        printf("plain rows %2d loads %d nwg %4d  %.1f us  %.2f TB/s\n", ROWS, LOADS, NWG, ms * 1e3, gb / ms); }
    PLAIN(32, 1, 256) PLAIN(32, 2, 256) PLAIN(64, 1, 256) PLAIN(64, 2, 256) PLAIN(32, 2, 512)
    DMA(32, 4, 0, true, 256) DMA(32, 4, 1, true, 256) DMA(32, 4, 0, false, 256) DMA(32, 4, 1, false, 256)
    for (int nj : {1, 2, 4}) for (int nwg : {256, 252}) {
        Jobs jobs; jobs.n = nj;
        const int lds = 4 * 2 * 32 * 512;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_read_jobs<32, 4, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        float ms = time_ms([&]() { for (int j = 0; j < nj; ++j) { next(); jobs.g[j] = G; jobs.x[j] = X; } dma_read_jobs<32, 4, 0, true><<<nwg / nj * nj, 512, lds>>>(jobs, M, out); }, 12);
        printf("dma jobs %d nwg %d: %.1f us  %.2f TB/s\n", nj, nwg / nj * nj, ms * 1e3, nj * gb / ms);
    }
    float* mx; hipMalloc(&mx, 2 * M * 4); hipMemset(mx, 0, 2 * M * 4);
#define JOBS(PITCH, MAXIMA, NJ, NWG) { Jobs jobs; jobs.n = NJ; const int lds = 4 * 32 * PITCH + 1024; \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_read_jobs<32, 4, 0, true, PITCH, MAXIMA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        float ms = time_ms([&]() { for (int j = 0; j < NJ; ++j) { next(); jobs.g[j] = G; jobs.x[j] = X; } dma_read_jobs<32, 4, 0, true, PITCH, MAXIMA><<<NWG, 512, lds>>>(jobs, M, out, mx); }, 12); \
        printf("dma jobs %d nwg %d pitch %d maxima %d: %.1f us  %.2f TB/s\n", NJ, NWG, PITCH, MAXIMA, ms * 1e3, NJ * gb / ms); }
    JOBS(1024, 0, 4, 256) JOBS(1088, 0, 4, 256) JOBS(1024, 1, 4, 256) JOBS(1088, 1, 4, 256) JOBS(1152, 0, 4, 256) JOBS(1280, 0, 4, 256)
    DMA(32, 3, 0, true, 256) DMA(32, 2, 0, true, 256)
    DMA(16, 8, 0, true, 256) DMA(16, 8, 1, true, 256)
    DMA(64, 2, 0, true, 256) DMA(64, 2, 1, true, 256)
    DMA(32, 4, 1, true, 252) DMA(32, 2, 1, true, 512) DMA(16, 4, 1, true, 512) DMA(16, 4, 1, true, 504)
    return 0;
}
