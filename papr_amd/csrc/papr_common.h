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
// process-wide A/B switches (papr_set_switch in papr_hip.h; the library reads no environment) and per-device facts
int papr_switch(int which);
int papr_rownorm_stats(const float* x, int64_t rows, int width, int ld, float eps, float* stats, float* mean, papr_stream_t stream);      // rowops.hip
int papr_rownorm_apply(float* x, int64_t rows, int width, int ld, const float* stats, const float* mean, papr_stream_t stream);      // rowops.hip
int papr_cu_count();                       // compute units of the CURRENT device (cached per device id)
bool papr_first_on_device(int slot);       // true once per (current device, slot): hipFuncSetAttribute calls of a launcher
enum { PAPR_ONCE_CHAIN4 = 0, PAPR_ONCE_NT_H3, PAPR_ONCE_TN_H3, PAPR_ONCE_CONV, PAPR_ONCE_CONV_WGRAD, PAPR_ONCE_PAIRS, PAPR_ONCE_TN_TR, PAPR_ONCE_SLOTS };

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

// Wave-wide reductions, result in every lane.  16 lanes meet on the DPP network (four steps, no LDS traffic), the four
// 16-lane rows on the scalar unit; a __shfl_xor butterfly is six ds_bpermute round trips through the LDS pipe.
#define PAPR_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xf, 0xf, false))
__device__ __forceinline__ float wave_sum(float v) {
    v += PAPR_DPP_F(v, 0xB1);      // quad_perm [1,0,3,2]
    v += PAPR_DPP_F(v, 0x4E);      // quad_perm [2,3,0,1]
    v += PAPR_DPP_F(v, 0x141);     // row_half_mirror
    v += PAPR_DPP_F(v, 0x140);     // row_mirror
    const int r = __float_as_int(v);
    const float a = __int_as_float(__builtin_amdgcn_readlane(r, 0)), b = __int_as_float(__builtin_amdgcn_readlane(r, 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(r, 32)), d = __int_as_float(__builtin_amdgcn_readlane(r, 48));
    return (a + b) + (c + d);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, PAPR_DPP_F(v, 0xB1));
    v = fmaxf(v, PAPR_DPP_F(v, 0x4E));
    v = fmaxf(v, PAPR_DPP_F(v, 0x141));
    v = fmaxf(v, PAPR_DPP_F(v, 0x140));
    const int r = __float_as_int(v);
    const float a = __int_as_float(__builtin_amdgcn_readlane(r, 0)), b = __int_as_float(__builtin_amdgcn_readlane(r, 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(r, 32)), d = __int_as_float(__builtin_amdgcn_readlane(r, 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// Tensor maxima left by the kernel that PRODUCES a tensor for the kernels that scale by it (small_unet.hip).  A "slot" is PAPR_SLOT_W = 256 words
// (zeroed beforehand): a workgroup adds ONE atomicMax of its max |.| (bit pattern) to word (its linear block index mod 256), a consumer wave takes
// the largest of the 256 (one 16-byte load per lane).  Not one word: same-address atomics are served one at a time at the memory side, ~10 ns each
// -- 12,800 of them (a wave each of a 3,200-workgroup elementwise launch) made a 7-us kernel a 150-us one, and even one per workgroup of a
// 1,024-workgroup launch cost 10 us; spread over 256 words the queue per word is four deep.  Every thread of the workgroup must call the writer
// (it holds two barriers: an early return of some threads deadlocks); it may be called more than once per kernel (the trailing barrier keeps a second
// call's writes off the table thread 0 is still reading).
constexpr int PAPR_SLOT_W = 256;
__device__ __forceinline__ void papr_wg_max_to_slot(unsigned* slot, float v) {
    __shared__ float papr_wmax[16];
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) papr_wmax[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = papr_wmax[0];
        for (unsigned i = 1; i < (blockDim.x + 63) >> 6; ++i) m = fmaxf(m, papr_wmax[i]);
        const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (m > 0.f) atomicMax(slot + (bid & (PAPR_SLOT_W - 1)), __float_as_uint(m));
    }
    __syncthreads();
}
// the maximum a slot holds (n == PAPR_SLOT_W), or the single word a caller of the single-layer entry points supplies (n == 1); in every lane
__device__ __forceinline__ unsigned papr_slot_max(const unsigned* slot, int n) {
    if (n == 1) return *slot;
    const uint4 v = reinterpret_cast<const uint4*>(slot)[threadIdx.x & 63];
    const unsigned a = v.x > v.y ? v.x : v.y, b = v.z > v.w ? v.z : v.w;
    return __float_as_uint(wave_max(__uint_as_float(a > b ? a : b)));       // (bit patterns of non-negative floats order like the floats)
}
