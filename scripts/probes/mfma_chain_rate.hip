// How fast does ONE wave issue the fused-run k-step (six 32x32x16 f16 MFMAs on two accumulators, A operand in AGPRs or VGPRs,
// four ds_read_b128 per k-step)?   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_chain_rate.hip -o /tmp/mcr && /tmp/mcr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(long long* out, float* sink, int waves_active) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
    __syncthreads();
    if (wave >= waves_active) return;
    f32x16 a0, a1;
    for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
    half8 x0, x1, y0, y1, w;
    for (int e = 0; e < 8; ++e) { x0[e] = (_Float16)0.5f; x1[e] = (_Float16)0.25f; y0[e] = (_Float16)0.125f; y1[e] = (_Float16)1.f; w[e] = (_Float16)0.75f; }
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0\n\t"
                 "v_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %0" :: "v"(0x3c003c00) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    const unsigned addr = (unsigned)(size_t)lds + lane * 16;
    long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        if (MODE == 0)          // A operand in AGPRs, no LDS
            asm volatile("v_mfma_f32_32x32x16_f16 %0, a[0:3], %4, %0\n\tv_mfma_f32_32x32x16_f16 %1, a[0:3], %5, %1\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, a[4:7], %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, a[4:7], %3, %1\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, a[0:3], %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, a[0:3], %3, %1"
                         : "+v"(a0), "+v"(a1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
        else if (MODE == 1)     // A operand in VGPRs, no LDS
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %6, %4, %0\n\tv_mfma_f32_32x32x16_f16 %1, %6, %5, %1\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, %6, %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, %6, %3, %1\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, %6, %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, %6, %3, %1"
                         : "+v"(a0), "+v"(a1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(w));
        else if (MODE == 2) {   // AGPRs + four ds_read_b128 for the next step in front of the block
            half8 n0, n1, n2, n3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:32768\n\tds_read_b128 %3, %4 offset:36864"
                         : "=v"(n0), "=v"(n1), "=v"(n2), "=v"(n3) : "v"(addr + ((it & 15) * 32)));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, a[0:3], %4, %0\n\tv_mfma_f32_32x32x16_f16 %1, a[0:3], %5, %1\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, a[4:7], %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, a[4:7], %3, %1\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, a[0:3], %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, a[0:3], %3, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
            x0 = n0; x1 = n2; y0 = n1; y1 = n3;
        } else if (MODE == 3) { // four accumulators (no dependency closer than four instructions), AGPRs
            asm volatile("v_mfma_f32_32x32x16_f16 %0, a[0:3], %4, %0\n\tv_mfma_f32_32x32x16_f16 %1, a[0:3], %5, %1\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, a[4:7], %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, a[4:7], %3, %1\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, a[0:3], %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, a[0:3], %3, %1"
                         : "+v"(a0), "+v"(a1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");       // thirty-two idle issue cycles between the blocks
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a0), "+v"(a1));
    long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[3];
}

int main() {
    long long* d; float* s;
    hipMalloc(&d, 64); hipMalloc(&s, 256 * 512 * 4);
    const char* names[4] = {"A in AGPRs", "A in VGPRs", "AGPRs + 4 ds_read_b128 per block", "AGPRs + 32 idle cycles per block"};
    for (int active = 4; active <= 8; active += 4)
        for (int m = 0; m < 4; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) k<0><<<256, 512>>>(d, s, active); else if (m == 1) k<1><<<256, 512>>>(d, s, active);
                else if (m == 2) k<2><<<256, 512>>>(d, s, active); else k<3><<<256, 512>>>(d, s, active);
            }
            long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            printf("%d waves/CU  %-36s  clock64 ticks per MFMA: wave0 %.2f  last wave %.2f\n", active, names[m], h[0] / (256.0 * 6), h[active - 1] / (256.0 * 6));
        }
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    int wall = 0; hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
    printf("shader clock %d kHz, wall clock (clock64?) %d kHz\n", khz, wall);
    return 0;
}
