// Does a wave's MFMA stream slow down when its SIMD partner issues plain VALU / LDS / DPP work, and by how much?
// Waves 0-3 of a 512-thread workgroup (one per SIMD) run the fused-run k-step pattern (six dependent-pair 32x32x16 f16 MFMAs
// per block); waves 4-7 (their partners) run `partner` = nothing / independent v_fma_f32 / v_fma_mix / ds_read_b128 / DPP max.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_valu_partner.hip -o scripts/probes/bin/mvp && scripts/probes/bin/mvp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PARTNER>
__global__ __launch_bounds__(512, 2) void k(long long* out, float* sink, int prio) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
    __syncthreads();
    long long t0 = 0, t1 = 0;
    float res = 0.f;
    if (wave < 4) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        f32x16 a0, a1;
        for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
        half8 x0, x1, y0, y1, w;
        for (int e = 0; e < 8; ++e) { x0[e] = (_Float16)0.5f; x1[e] = (_Float16)0.25f; y0[e] = (_Float16)0.125f; y1[e] = (_Float16)1.f; w[e] = (_Float16)0.75f; }
        t0 = clock64();
#pragma unroll 1
        for (int it = 0; it < 256; ++it)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %6, %4, %0\n\tv_mfma_f32_32x32x16_f16 %1, %6, %5, %1\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, %6, %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, %6, %3, %1\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, %6, %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, %6, %3, %1"
                         : "+v"(a0), "+v"(a1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(w));
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a0), "+v"(a1));
        t1 = clock64();
        res = a0[0] + a1[3];
    } else {
        float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = 1.f, v5 = 2.f, v6 = 3.f, v7 = 4.f;
        const unsigned addr = (unsigned)(size_t)lds + lane * 16;
        t0 = clock64();
#pragma unroll 1
        for (int it = 0; it < 1024; ++it) {
            if (PARTNER == 1)       // 16 independent-ish plain VALU
                asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
                             "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
                             "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
                             "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5));
            else if (PARTNER == 2)  // 16 v_fma_mixlo_f16
                asm volatile("v_fma_mixlo_f16 %0, %1, %4, 0\n\tv_fma_mixlo_f16 %0, %2, %4, 0\n\tv_fma_mixlo_f16 %0, %3, %4, 0\n\tv_fma_mixlo_f16 %0, %1, %5, 0\n\t"
                             "v_fma_mixlo_f16 %0, %1, %4, 0\n\tv_fma_mixlo_f16 %0, %2, %4, 0\n\tv_fma_mixlo_f16 %0, %3, %4, 0\n\tv_fma_mixlo_f16 %0, %1, %5, 0\n\t"
                             "v_fma_mixlo_f16 %0, %1, %4, 0\n\tv_fma_mixlo_f16 %0, %2, %4, 0\n\tv_fma_mixlo_f16 %0, %3, %4, 0\n\tv_fma_mixlo_f16 %0, %1, %5, 0\n\t"
                             "v_fma_mixlo_f16 %0, %1, %4, 0\n\tv_fma_mixlo_f16 %0, %2, %4, 0\n\tv_fma_mixlo_f16 %0, %3, %4, 0\n\tv_fma_mixlo_f16 %0, %1, %5, 0"
                             : "+v"(v0) : "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5));
            else if (PARTNER == 3) { // 4 ds_read_b128 + wait
                float4 r0, r1, r2, r3;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %4 offset:12288\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr));
                v0 += r0.x + r1.y + r2.z + r3.w;
            } else if (PARTNER == 4) // 16 DPP max
                asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_max_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                             "v_max_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                             "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                             "v_max_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_max_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                             "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                             "v_max_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_max_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            else break;
        }
        t1 = clock64();
        res = v0 + v1 + v2 + v3 + v6 + v7;
    }
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = res;
}

int main() {
    long long* d; float* s;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 256 * 512 * 4);
    const char* names[5] = {"idle", "v_fma_f32 x16", "v_fma_mixlo_f16 x16", "4 ds_read_b128 + wait", "v_max_f32_dpp x16"};
    for (int prio = 0; prio <= 3; prio += 3)
        for (int m = 0; m < 5; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) k<0><<<256, 512>>>(d, s, prio); else if (m == 1) k<1><<<256, 512>>>(d, s, prio); else if (m == 2) k<2><<<256, 512>>>(d, s, prio);
                else if (m == 3) k<3><<<256, 512>>>(d, s, prio); else k<4><<<256, 512>>>(d, s, prio);
            }
            long long h[8]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            printf("MFMA wave prio %d, partner %-24s cycles per MFMA %.1f   partner: cycles per block of 16 %.1f\n", prio, names[m], h[0] / (256.0 * 6), m ? h[4] / 1024.0 : 0.0);
        }
    return 0;
}
