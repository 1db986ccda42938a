// Device helpers shared by the split-f16 GEMM kernels (gemm.hip, chain.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a full workgroup fence, and
// on gfx950 loads and stores share vmcnt: the fence therefore drains every global load in flight, i.e.
// the slab prefetch, at each barrier.  Harmless when a slab's matrix work outlasts an HBM round trip
// (fp32 MFMA), fatal when it does not (split-f16: ~1.5k cycles of MFMA per slab).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pow2_from_biased(int e) {       // 2^(e-127), e clamped to the normal range
    e = e < 1 ? 1 : (e > 254 ? 254 : e);
    return __uint_as_float((unsigned)e << 23);
}

// row maximum -> the power of two that brings it into [2^13, 2^14) (returned) and its reciprocal (inv); a zero row: scale 1 ... (exponent 140)
__device__ __forceinline__ float h3_scale_from_max(unsigned bits, float& inv) {
    const int ea = bits ? (int)((bits >> 23) & 0xff) : 127 + 13;
    inv = pow2_from_biased(127 - 13 + (ea - 127));
    return pow2_from_biased(127 + 13 - (ea - 127));
}

// One-product mode (round 6): ONE power-of-two scale per row and RUN of layers, chosen from the maximum of the run's input row (forward) or of its
// top gradient row (data-gradient) and carried through every layer of the run in the "scaled domain" (accumulators, biases times the scale, f16 rows
// all hold value x scale); only what leaves the run is multiplied by 1 / scale.  The scale brings the row maximum into [2^3, 2^4): 2^12 of headroom
// above it for the layers' activations / gradients before f16 overflows, full f16 precision down to 2^-17 of it.  The maximum's exponent is clamped
// from below (biased exponent `emin`): forward rows whose maximum is below 1 are scaled as if it were 1 -- the biases ride in the same scaled
// domain --, gradient rows down to 2^-40.  The weight-gradient kernel turns the same maximum into the same power of two (gemm.hip).
#ifndef ONE_TARGET_E_V
#define ONE_TARGET_E_V 3                       // (6 until the end of round 6: 2^9 of headroom.  A trained network's inner gradients were 8x its top gradients by step
                                               //  5,000 of a chair.yml run; the golden and h1 error tables are the same to the third digit with 3, 6 and 9: the
                                               //  headroom is the scarcer side.  Probe builds: -DONE_TARGET_E_V=..., scripts/probes/lib_variant3.sh)
#endif
constexpr int ONE_TARGET_E = ONE_TARGET_E_V, ONE_EMIN_FWD = 127, ONE_EMIN_DGRAD = 87;
__device__ __forceinline__ float one_scale_from_max(unsigned bits, int emin, float& inv) {
    int ea = (int)((bits >> 23) & 0xff);
    ea = ea < emin ? emin : ea;
    inv = pow2_from_biased(ea - ONE_TARGET_E);
    return pow2_from_biased(254 + ONE_TARGET_E - ea);
}

// hi = f16(v * s), lo = f16(v * s - hi) for four values; s is a power of two, so v * s is exact and the fused forms
// below round exactly like the multiply / convert / convert back / subtract / convert sequence they replace:
// v_fma_mixlo/hi_f16 write one half of a register from an fp32 fma, and take the f16 hi as an operand: 8 instructions
// instead of ~20 per four values (they issue at half rate, `scripts/probes/mfma_valu_overlap.hip`).  Worth 14 % in the
// weight-gradient kernel, whose split sits between matrix instructions; 1-2 % in the fused-run kernel (in-box A/B).
__device__ __forceinline__ void split4(const float4& v, float s, half4& hi, half4& lo) {
    unsigned h01, h23, l01, l23;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h01) : "v"(v.x), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h23) : "v"(v.z), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(v.y), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(v.w), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l01) : "v"(v.x), "v"(s), "v"(h01));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l23) : "v"(v.z), "v"(s), "v"(h23));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(v.y), "v"(s), "v"(h01));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(v.w), "v"(s), "v"(h23));
    const uint2 h = make_uint2(h01, h23), l = make_uint2(l01, l23);
    hi = *reinterpret_cast<const half4*>(&h);
    lo = *reinterpret_cast<const half4*>(&l);
}
