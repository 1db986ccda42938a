// One wave per SIMD runs a pure MFMA stream, a second wave on the same SIMD a pure vector stream: how fast does each go?
// (fused-run kernel: one workgroup multiplies while the other one of the CU is in its row phases)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/mfma_vs_valu_waves scripts/probes/mfma_vs_valu_waves.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
// mode bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run vector instructions (kind: 0 = v_fma_f32, 1 = v_fma_mixlo, 2 = ds_read)
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 1e-3f + e); b[e] = (_Float16)(1.0f + e * 0.1f); }
    float f0 = threadIdx.x, f1 = 1.0001f, f2 = 0.5f, f3 = 3.f, f4 = 4.f, f5 = 5.f;
    unsigned h = 0;
    __shared__ float4 lds[512];
    lds[threadIdx.x] = make_float4(1, 2, 3, 4);
    __syncthreads();
    float4 dl = make_float4(0, 0, 0, 0);
    long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int r = 0; r < 12; ++r) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r & 3]) : "v"(a), "v"(b));
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int r = 0; r < 24; ++r) {
                    if (KIND == 0) {
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(f1), "v"(f2));
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f4) : "v"(f1), "v"(f2));
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f5) : "v"(f1), "v"(f2));
                    } else if (KIND == 1) {
                        asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(f1), "v"(f2));
                        asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(f1), "v"(f2));
                        asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(f1), "v"(f2));
                        asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(f1), "v"(f2));
                    } else {
                        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dl) : "v"((threadIdx.x & 255) * 16) : "memory");
                        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dl) : "v"((threadIdx.x & 255) * 16) : "memory");
                        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dl) : "v"((threadIdx.x & 255) * 16) : "memory");
                        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dl) : "v"((threadIdx.x & 255) * 16) : "memory");
                    }
                }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = f0 + f3 + f4 + f5 + h + dl.x;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) cyc[wave / 4] = t1 - t0;
}
template <int KIND> void run(const char* tag, float* out, long long* cyc) {
    const int iters = 2000;
    for (int mode = 1; mode <= 3; ++mode) {
        k<KIND><<<256, 512>>>(out, 20, mode, cyc); (void)hipDeviceSynchronize();
        k<KIND><<<256, 512>>>(out, iters, mode, cyc); (void)hipDeviceSynchronize();
        long long c[2]; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
        printf("%-12s %-22s: %6.1f cycles per MFMA (wave 0), %6.2f cycles per vector instruction (wave 4)\n", tag,
               mode == 1 ? "MFMA wave alone" : mode == 2 ? "vector wave alone" : "both on the SIMD",
               (mode & 1) ? (double)c[0] / iters / 12 : 0.0, (mode & 2) ? (double)c[1] / iters / 96 : 0.0);
    }
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 16);
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_fma_mixlo", out, cyc);
    run<2>("ds_read_b128", out, cyc);
    return 0;
}
