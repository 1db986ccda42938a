// Read-bandwidth probe for the weight-gradient kernel's access pattern (two M x 256 fp32 operands).
//  A: one contiguous row slice per workgroup (what gemm_tn_h3 does), 32-row stages, 512 threads
//  B: the same stages dealt round-robin to the workgroups
//  C: plain grid-stride float4 read, many small workgroups
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/read_patterns scripts/probes/read_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE, int LOADS>
__global__ __launch_bounds__(512) void staged_read(const float4* __restrict__ X, const float4* __restrict__ G, long M, float* out) {
    // a stage = 32 rows x 256 floats of each operand = 2048 float4 per operand: 4 float4 per thread per operand
    const long stages = M / 32;
    const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    long s_begin, s_end, s_step;
    if (MODE == 0) { long per = (stages + nwg - 1) / nwg; s_begin = wg * per; s_end = s_begin + per < stages ? s_begin + per : stages; s_step = 1; }
    else { s_begin = wg; s_end = stages; s_step = nwg; }
    float4 acc = make_float4(0, 0, 0, 0);
    for (long s = s_begin; s < s_end; s += s_step * LOADS) {
        float4 x[LOADS][4], g[LOADS][4];
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            long ss = s + u * s_step; if (ss >= s_end) ss = s;
            const float4* px = X + ss * 2048; const float4* pg = G + ss * 2048;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[u][i] = px[tid + 512 * i]; g[u][i] = pg[tid + 512 * i]; }
        }
#pragma unroll
        for (int u = 0; u < LOADS; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc.x += x[u][i].x * g[u][i].x; acc.y += x[u][i].y * g[u][i].y; acc.z += x[u][i].z * g[u][i].z; acc.w += x[u][i].w * g[u][i].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

__global__ __launch_bounds__(256) void stride_read(const float4* __restrict__ X, long n4, float* out) {
    float4 acc = make_float4(0, 0, 0, 0);
    long i = (long)blockIdx.x * 256 + threadIdx.x, step = (long)gridDim.x * 256;
    for (; i + 3 * step < n4; i += 4 * step) {
        float4 a = X[i], b = X[i + step], c = X[i + 2 * step], d = X[i + 3 * step];
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y; acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
    }
    for (; i < n4; i += step) { float4 a = X[i]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <typename F> static float time_ms(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
    const long M = 512000;
    const size_t bytes = (size_t)M * 256 * 4;
    float *X, *G, *out;
    // several buffer pairs so that consecutive launches do not find their data in the 256 MB infinity cache
    const int NB = 4;
    std::vector<float*> xs(NB), gs(NB);
    for (int i = 0; i < NB; ++i) { hipMalloc(&xs[i], bytes); hipMalloc(&gs[i], bytes); hipMemset(xs[i], 0, bytes); hipMemset(gs[i], 0, bytes); }
    hipMalloc(&out, 64);
    int it = 0;
    auto next = [&]() { X = xs[it % NB]; G = gs[it % NB]; ++it; };
    const double gb = 2.0 * bytes / 1e9;
#define RUN(name, MODE, LOADS, NWG) { float ms = time_ms([&]() { next(); staged_read<MODE, LOADS><<<NWG, 512>>>((const float4*)X, (const float4*)G, M, out); }, 20); \
        printf("%-34s nwg=%4d  %.1f us  %.2f TB/s\n", name, NWG, ms * 1e3, gb / ms); }
    RUN("A slice/WG, 1 stage in flight", 0, 1, 256)
    RUN("A slice/WG, 1 stage in flight", 0, 1, 512)
    RUN("A slice/WG, 2 stages in flight", 0, 2, 256)
    RUN("A slice/WG, 2 stages in flight", 0, 2, 512)
    RUN("A slice/WG, 1 stage in flight", 0, 1, 1024)
    RUN("B round-robin, 1 stage", 1, 1, 256)
    RUN("B round-robin, 1 stage", 1, 1, 512)
    RUN("B round-robin, 2 stages", 1, 2, 256)
    RUN("B round-robin, 2 stages", 1, 2, 512)
    RUN("B round-robin, 1 stage", 1, 1, 1024)
    RUN("B round-robin, 1 stage", 1, 1, 2048)
    for (int nwg : {2048, 8192, 32768}) {
        float ms = time_ms([&]() { next(); stride_read<<<nwg, 256>>>((const float4*)X, (long)(bytes / 16), out); }, 20);
        printf("%-34s nwg=%5d  %.1f us  %.2f TB/s\n", "C grid-stride one operand", nwg, ms * 1e3, bytes / 1e9 / ms);
    }
    return 0;
}
