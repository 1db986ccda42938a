// How fast can 256 workgroups pull one 64-row tile each (16 MB in one burst), by where the tiles lie?
//   mode 0: tile b = rows [128 b, 128 b + 64) x 1 KB          (chain3 / chain4 staging: regions 128 KB apart, every CU at the same row offset)
//   mode 1: tile b row i = global row i * 512 + 2 b            (rows of one step adjacent in memory)
//   mode 2: like 0 but each workgroup starts at a different row (rotation by 5 b mod 64)
// hipcc --offload-arch=gfx950 -O3 scripts/probes/stage_pattern.hip -o scripts/probes/bin/stage_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void pull(const float4* __restrict__ x, float* out, int mode, int reps) {
    const int b = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int i = 8 * w + q;
            long row;
            if (mode == 0) row = 128L * b + i + 32768L * rep;
            else if (mode == 1) row = (long)i * 512 + 2 * b + 32768L * rep;
            else row = 128L * b + ((i + 5 * b) & 63) + 32768L * rep;
            v[q] = x[row * 64 + lane];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q].x + v[q].w;
        __syncthreads();
    }
    out[b * 512 + threadIdx.x] = acc;
}
int main() {
    float4* x; float* out;
    const size_t rows = 32768L * 16;
    hipMalloc(&x, rows * 1024); hipMalloc(&out, 256 * 512 * 4);
    hipMemset(x, 0, rows * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int reps : {1, 16}) {
            float best = 1e9;
            for (int t = 0; t < 5; ++t) {
                hipMemset(out, 0, 4);        // (something in between so that the caches turn over a little)
                hipEventRecord(e0); pull<<<256, 512>>>(x, out, mode, reps); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            printf("mode %d reps %2d: %.1f us  = %.2f TB/s\n", mode, reps, best * 1e3, 16.777216e6 * reps / (best * 1e-3) / 1e12);
        }
    return 0;
}
