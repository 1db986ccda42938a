// Shared host/device helpers for libpapr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/papr_hip.h"

void papr_set_error(const char* fmt, ...);
bool papr_prof_on();
void papr_prof_begin(int kernel, long M, int N, int K, hipStream_t s);
void papr_prof_begin2(int kernel, long M, int N, int K, long long bytes, long long flops, hipStream_t s);
void papr_prof_end(hipStream_t s);

#define PAPR_REQUIRE(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            papr_set_error(__VA_ARGS__);       \
            return 1;                          \
        }                                      \
    } while (0)

#define PAPR_CHECK_LAUNCH(name)                                                  \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            papr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return 2;                                                            \
        }                                                                        \
    } while (0)

static inline hipStream_t as_stream(papr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float papr_act(float x, int act) {
    if (act == PAPR_ACT_RELU) return x > 0.f ? x : 0.f;
    if (act == PAPR_ACT_LEAKY_RELU) return x > 0.f ? x : 0.2f * x;
    return x;
}
// derivative expressed on the activation's OUTPUT y (sign(y) == sign(pre-activation))
__device__ __forceinline__ float papr_act_grad(float y, int act) {
    if (act == PAPR_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == PAPR_ACT_LEAKY_RELU) return y > 0.f ? 1.f : 0.2f;
    return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
