// MODE.FP16_OVFL (bit 23): does an f16 result that overflows clamp to +-65504 instead of becoming inf -- for v_cvt_pk_f16_f32, v_fma_mixlo_f16, v_pk_mul_f16 -- and does a true inf stay inf?
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/fp16_ovfl scripts/probes/fp16_ovfl.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int set) {
    if (set) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const float a = in[0], b = in[1], one = in[2];
    unsigned r0, r1, r2;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r0) : "v"(a), "v"(b));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r1) : "v"(a), "v"(one));
    unsigned h = 0x7bff7bffu, four = 0x44004400u;      // 65504 | 65504, 4.0 | 4.0
    asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r2) : "v"(h), "v"(four));
    out[0] = r0; out[1] = r1; out[2] = r2;
    if (set) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
}
int main() {
    float hin[3] = {1.0e6f, __builtin_inff(), 1.0f}; float* din; unsigned* dout; unsigned hout[3];
    hipMalloc(&din, 12); hipMalloc(&dout, 12); hipMemcpy(din, hin, 12, hipMemcpyHostToDevice);
    for (int set = 0; set < 2; ++set) {
        k<<<1, 64>>>(din, dout, set); hipMemcpy(hout, dout, 12, hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d: cvt_pk(1e6, inf) = %04x %04x   fma_mixlo(1e6) = %04x   pk_mul(65504 x 4) = %04x   (7bff = 65504, 7c00 = inf)\n", set, hout[0] & 0xffff, hout[0] >> 16, hout[1] & 0xffff, hout[2] & 0xffff);
    }
    return 0;
}
