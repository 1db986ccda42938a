// Sustained v_mfma_f32_32x32x2_f32 rate and effective clock on this chip (no memory traffic).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* cyc) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu *= 2) {
        int grid = 256 * blocks_per_cu, iters = 20000;
        mfma_loop<4><<<grid, 256>>>(out, 100, cyc); hipDeviceSynchronize();
        hipEventRecord(a); mfma_loop<4><<<grid, 256>>>(out, iters, cyc); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double flops = (double)grid * 4 /*waves*/ * iters * 4 /*acc*/ * 2.0 * 32 * 32 * 2;
        printf("waves/SIMD=%d  %.1f TFLOP/s  (%.2f ms)  cycles/iter(wave0)=%.1f  eff.clock=%.2f GHz\n", blocks_per_cu,
               flops / ms / 1e9, ms, (double)c / iters, (double)c / (ms * 1e6));
    }
    return 0;
}
