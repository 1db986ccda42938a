// How many independent vector instructions fit under a v_mfma_f32_32x32x16_f16 (32 cycles of matrix pipe)?
// cycles per MFMA with NF fillers after every MFMA; 1 or 2 waves per SIMD; accumulators in VGPRs or AGPRs.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/mfma_valu_overlap scripts/probes/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int KIND, int NF, bool AGPR, int THREADS>
__global__ __launch_bounds__(THREADS) void loop(float* out, int iters, long long* cyc) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 1e-3f + e); b[e] = (_Float16)(1.0f + e * 0.1f); }
    unsigned x32 = threadIdx.x;
    float f0 = threadIdx.x, f1 = 1.0f, f2 = 2.0f, f3 = 3.f;
    unsigned h = 0;
    long long w0 = wall_clock64();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x32) : "s"(it));
                    if (KIND == 6) { if (f & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(f1), "v"(f2)); }
                    if (KIND == 7) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(f1), "v"(f2));
                }
            }
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    float s = (float)x32 + f0 + f3 + h;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <int KIND, int NF, bool AGPR, int THREADS>
void run(const char* tag, float* out, long long* cyc) {
    int grid = 256, iters = 2000;
    loop<KIND, NF, AGPR, THREADS><<<grid, THREADS>>>(out, 20, cyc); (void)hipDeviceSynchronize();
    hipEvent_t ea, eb; (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    (void)hipEventRecord(ea);
    loop<KIND, NF, AGPR, THREADS><<<grid, THREADS>>>(out, iters, cyc);
    (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
    float ms; (void)hipEventElapsedTime(&ms, ea, eb);
    printf("[%.3f ms, %.0f TFLOP/s] ", ms, (double)grid * (THREADS / 64) * iters * 12 * 32768.0 / ms / 1e9);
    long long c[2]; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-14s fillers=%d acc=%s waves/SIMD=%d: %6.1f ticks per MFMA per wave, %6.2f ns (100 MHz wall counter: %lld)\n", tag, NF, AGPR ? "agpr" : "vgpr", THREADS / 256, (double)c[0] / iters / 12, (double)c[1] * 10.0 / iters / 12, c[1]);
}
template <int KIND, bool AGPR, int THREADS> void sweep(const char* tag, float* out, long long* cyc) {
    run<KIND, 0, AGPR, THREADS>(tag, out, cyc); run<KIND, 2, AGPR, THREADS>(tag, out, cyc); run<KIND, 4, AGPR, THREADS>(tag, out, cyc);
    run<KIND, 6, AGPR, THREADS>(tag, out, cyc); run<KIND, 8, AGPR, THREADS>(tag, out, cyc); run<KIND, 12, AGPR, THREADS>(tag, out, cyc);
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4096 * 512 * 4); (void)hipMalloc(&cyc, 16);
    sweep<6, false, 256>("v_fma_f32", out, cyc);
    sweep<6, false, 512>("v_fma_f32", out, cyc);
    sweep<2, false, 512>("v_add_u32", out, cyc);
    sweep<7, false, 512>("v_fma_mixlo", out, cyc);
    return 0;
}
