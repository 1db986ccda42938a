// What one wave per SIMD loses when single instructions sit between its v_mfma_f32_32x32x16_f16:
// cycles per MFMA with 0/1/2 fillers of a given kind after every MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int KIND, int NF>
__global__ __launch_bounds__(256) void loop(float* out, const float4* src, int iters, long long* cyc) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 1e-3f + e); b[e] = (_Float16)(1.0f + e * 0.1f); }
    unsigned long long p64 = (unsigned long long)src + threadIdx.x * 16;
    unsigned x32 = threadIdx.x;
    unsigned sc = blockIdx.x;
    float4 ld = make_float4(0, 0, 0, 0);
    extern __shared__ float4 lds[];
    lds[threadIdx.x] = make_float4(1, 2, 3, 4);
    __syncthreads();
    float4 dl = make_float4(0, 0, 0, 0);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    if (KIND == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(p64) : "s"((unsigned long long)it));
                    if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x32) : "s"(it));
                    if (KIND == 3) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld) : "v"(p64) : "memory");
                    if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(dl) : "v"(x32 * 16 & 4095) : "memory");
                    if (KIND == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        if (KIND == 3 || KIND == 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    float s = (float)p64 + x32 + ld.x + dl.x + sc;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND, int NF>
void run(const char* tag, float* out, const float4* src, long long* cyc) {
    int grid = 256, iters = 2000;
    loop<KIND, NF><<<grid, 256, 4096>>>(out, src, 20, cyc); (void)hipDeviceSynchronize();
    loop<KIND, NF><<<grid, 256, 4096>>>(out, src, iters, cyc); (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s x%d per MFMA: %.1f cycles/MFMA\n", tag, NF, (double)c / iters / 12);
}
int main() {
    float* out; long long* cyc; float4* src;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&src, 1 << 20);
    run<0, 0>("no filler", out, src, cyc);
    run<1, 1>("v_lshl_add_u64", out, src, cyc); run<1, 2>("v_lshl_add_u64", out, src, cyc);
    run<2, 1>("v_add_u32", out, src, cyc); run<2, 2>("v_add_u32", out, src, cyc); run<2, 4>("v_add_u32", out, src, cyc);
    run<3, 1>("global_load_dwordx4 (vaddr64)", out, src, cyc);
    run<4, 1>("ds_read_b128", out, src, cyc);
    run<5, 2>("s_add_u32", out, src, cyc);
    return 0;
}
