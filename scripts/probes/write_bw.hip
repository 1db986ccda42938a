// HBM write bandwidth of the access pattern the fused runs leave behind: 256 workgroups x 512 threads, each wave stores whole 1 KB rows
// (16 bytes per lane), row after row, tile after tile; plus the same with a read of the rows in front (copy).
// hipcc --offload-arch=gfx950 -O3 scripts/probes/write_bw.hip -o scripts/probes/bin/write_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void wr(float4* __restrict__ y, long rows, int mode, const float4* __restrict__ x) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long tiles = rows / 64;
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const long row = t * 64 + 8 * w + q;
            float4 v = make_float4((float)row, 1.f, 2.f, (float)lane);
            if (mode == 1) v = x[row * 64 + lane];
            y[row * 64 + lane] = v;
        }
    }
}
int main() {
    const long rows = 2048000;      // 2 GB
    float4 *x, *y;
    hipMalloc(&x, rows * 1024); hipMalloc(&y, rows * 1024);
    hipMemset(x, 0, rows * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int grid : {256, 512, 1024}) {
            float best = 1e9;
            for (int t = 0; t < 4; ++t) {
                hipEventRecord(e0); wr<<<grid, 512>>>(y, rows, mode, x); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            printf("%s grid %4d: %.0f us  = %.2f TB/s %s\n", mode ? "copy " : "write", grid, best * 1e3, rows * 1024.0 / (best * 1e-3) / 1e12, mode ? "(each way)" : "");
        }
    return 0;
}
