// v_mfma_f32_32x32x16_f16 issue rate of ONE wave per SIMD with the fused-run kernel's accumulator pattern:
// NACC accumulators, each updated REP times per round, consecutive updates of one accumulator SPACING MFMAs apart.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int NACC, int REP, bool INTERLEAVE>
__global__ __launch_bounds__(256) void loop(float* out, int iters, long long* cyc) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 1e-3f + e); b[e] = (_Float16)(1.0f + e * 0.1f); }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (INTERLEAVE) {
#pragma unroll
            for (int r = 0; r < REP; ++r)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int r = 0; r < REP; ++r) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, int REP, bool IL>
void run(const char* tag, int wps, float* out, long long* cyc) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int grid = 256 * wps, iters = 4000;
    loop<NACC, REP, IL><<<grid, 256>>>(out, 50, cyc); hipDeviceSynchronize();
    hipEventRecord(a); loop<NACC, REP, IL><<<grid, 256>>>(out, iters, cyc); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double flops = (double)grid * 4 * iters * NACC * REP * 2.0 * 32 * 32 * 16;
    printf("%-34s waves/SIMD=%d  %7.1f TFLOP/s  cycles/MFMA(wave0)=%.1f  eff.clock=%.2f GHz\n", tag, wps, flops / ms / 1e9,
           (double)c / iters / (NACC * REP), (double)c / (ms * 1e6));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    for (int w = 1; w <= 2; ++w) {
        run<4, 3, false>("4 acc x 3 back-to-back", w, out, cyc);
        run<4, 3, true>("4 acc x 3, 4 apart", w, out, cyc);
        run<2, 3, true>("2 acc x 3, 2 apart", w, out, cyc);
        run<12, 1, true>("12 independent", w, out, cyc);
    }
    return 0;
}
