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

__device__ __forceinline__ void split4(const float4& v, float s, half4& hi, half4& lo) {
    float x0 = v.x * s, x1 = v.y * s, x2 = v.z * s, x3 = v.w * s;
    hi = half4{(_Float16)x0, (_Float16)x1, (_Float16)x2, (_Float16)x3};
    lo = half4{(_Float16)(x0 - (float)hi[0]), (_Float16)(x1 - (float)hi[1]), (_Float16)(x2 - (float)hi[2]), (_Float16)(x3 - (float)hi[3])};
}

