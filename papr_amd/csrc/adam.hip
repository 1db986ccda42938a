// K8: the optimizer step of PAPR.step (reference models/model.py:439-460: `scaler.step(opt)` for each of its five to seven
// torch.optim.Adam instances) as ONE launch over all parameters of all optimizers.
//
// torch's fused Adam is one multi-tensor launch per optimizer plus one `_foreach_add_` launch for the step counters: twelve
// launches of 25-40 us for 150 MB of traffic (profiles/r02_kernel_trace_v5_final.txt: 0.25 ms per step).  Here a launch
// carries up to 64 tensors in its kernel arguments -- parameter, gradient, first and second moment, step counter, size and
// the index of its hyper-parameter group -- and the groups' scalars (learning rate, betas, eps, weight decay); blockIdx.y is
// the tensor, blockIdx.x a 4,096-element chunk of it (blocks past the tensor's end leave at once).  The step counters stay
// on the device like torch's: a first small launch adds one to each, the main launch reads its tensor's counter and forms the
// bias corrections from it in double precision, once per workgroup.  The arithmetic is torch.optim.Adam's (amsgrad off,
// maximize off; `torch/optim/adam.py: _single_tensor_adam`, the order of the fused kernel's `adam_math`):
//     g    = grad + weight_decay * p
//     m    = m + (1 - beta1) (g - m)                     (lerp)
//     v    = beta2 v + (1 - beta2) g g
//     p    = p - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),      bc_i = 1 - beta_i^step
// on the optimizers' own state tensors (`exp_avg`, `exp_avg_sq`, `step`), so checkpoints stay torch's.
#include "papr_common.h"

namespace {

constexpr int ADAM_CHUNK = 4096;            // elements per workgroup (256 threads x 4 float4)

struct AdamBatch {
    papr_adam_tensor t[PAPR_ADAM_MAX_TENSORS];
    papr_adam_group g[PAPR_ADAM_MAX_GROUPS];
};

// found_inf (GradScaler, `use_amp: true`): 1.0f once any gradient element of the call is not finite, else what it held (the caller zeroes it).
// torch's `_amp_foreach_non_finite_check_and_unscale_` + the `found_inf` argument of its fused Adam as one pass of this grid: no step, no
// counter, no moment moves when it is set (torch/optim/adam.py: _fused_adam -- the step counters are bumped and un-bumped there)
__global__ __launch_bounds__(256) void adam_check_kernel(AdamBatch b, float* __restrict__ found_inf) {
    const papr_adam_tensor& t = b.t[blockIdx.y];
    const long base = (long)blockIdx.x * ADAM_CHUNK;
    if (base >= t.n) return;
    bool bad = false;
    for (int u = 0; u < 16; ++u) {
        const long e = base + (long)u * 256 + threadIdx.x;
        if (e < t.n) { const float g = t.g[e]; bad |= !(fabsf(g) <= 3.402823466e38f); }      // (NaN and +-inf fail the comparison)
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) {                 // (racing stores of the same value)
        found_inf[b.g[t.group].found_slot] = 1.0f;
        found_inf[PAPR_ADAM_MAX_GROUPS] = 1.0f;                  // "any slot": what GradScaler.update() makes of the slots (their sum > 0)
    }
}

__global__ __launch_bounds__(64) void adam_bump_steps_kernel(AdamBatch b, int n, const float* __restrict__ found_inf) {
    if ((int)threadIdx.x >= n) return;
    if (found_inf && found_inf[b.g[b.t[threadIdx.x].group].found_slot] != 0.f) return;
    *b.t[threadIdx.x].step += 1.0f;
}

__global__ __launch_bounds__(256) void adam_step_kernel(AdamBatch b, const float* __restrict__ grad_scale, const float* __restrict__ found_inf) {
    __shared__ float bc[2];
    const papr_adam_tensor& t = b.t[blockIdx.y];
    const long base = (long)blockIdx.x * ADAM_CHUNK;
    if (base >= t.n) return;
    if (found_inf && found_inf[b.g[t.group].found_slot] != 0.f) return;      // an overflowed optimizer is skipped whole
    const float inv_scale = grad_scale ? 1.0f / *grad_scale : 1.0f;    // (the GradScaler's scale is a power of two: the product below is exact)
    const papr_adam_group& h = b.g[t.group];
    if (threadIdx.x == 0) {
        const double step = (double)*t.step;                       // (already this step's number)
        bc[0] = (float)(h.lr / (1.0 - pow(h.beta1, step)));        // step size
        bc[1] = (float)(1.0 / sqrt(1.0 - pow(h.beta2, step)));
    }
    __syncthreads();
    // (the hyper-parameters are doubles like torch's Python floats: 1 - beta is formed in double and rounded once)
    const float lr_c = bc[0], inv_bc2s = bc[1];
    const float one_m_b1 = (float)(1.0 - h.beta1), one_m_b2 = (float)(1.0 - h.beta2), beta2 = (float)h.beta2, eps = (float)h.eps, wd = (float)h.weight_decay;
    auto upd = [&](float& p, float g, float& m, float& v) {
        if (grad_scale) g = g * inv_scale;
        if (wd != 0.f) g = g + wd * p;
        m = m + one_m_b1 * (g - m);
        v = beta2 * v + one_m_b2 * g * g;
        const float denom = sqrtf(v) * inv_bc2s + eps;
        p = p - lr_c * (m / denom);
    };
    // 16-byte accesses only where all four arrays allow them: a gradient that is a view into the data-parallel flat bucket starts
    // wherever the tensors before it end (4-byte aligned only once 3 P is no multiple of 4 after a prune)
    const bool vec = (t.n & 3) == 0 && ((((uintptr_t)t.p) | ((uintptr_t)t.g) | ((uintptr_t)t.m) | ((uintptr_t)t.v)) & 15) == 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long e = base + 4L * (u * 256 + threadIdx.x);
        if (e >= t.n) break;
        if (vec) {
            float4 p = *reinterpret_cast<float4*>(t.p + e), m = *reinterpret_cast<float4*>(t.m + e), v = *reinterpret_cast<float4*>(t.v + e);
            const float4 g = *reinterpret_cast<const float4*>(t.g + e);
            upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
            *reinterpret_cast<float4*>(t.p + e) = p; *reinterpret_cast<float4*>(t.m + e) = m; *reinterpret_cast<float4*>(t.v + e) = v;
        } else {
            for (long i = e; i < e + 4 && i < t.n; ++i) upd(t.p[i], t.g[i], t.m[i], t.v[i]);
        }
    }
}

}  // namespace

static int adam_step_impl(const papr_adam_tensor* tensors, int32_t n_tensors, const papr_adam_group* groups, int32_t n_groups,
                          const float* grad_scale, float* found_inf, papr_stream_t stream);

extern "C" int papr_adam_step(const papr_adam_tensor* tensors, int32_t n_tensors, const papr_adam_group* groups, int32_t n_groups, papr_stream_t stream) {
    return adam_step_impl(tensors, n_tensors, groups, n_groups, nullptr, nullptr, stream);
}

extern "C" int papr_adam_step_scaled(const papr_adam_tensor* tensors, int32_t n_tensors, const papr_adam_group* groups, int32_t n_groups,
                                     const float* grad_scale, float* found_inf, papr_stream_t stream) {
    PAPR_REQUIRE(grad_scale && found_inf, "papr_adam_step_scaled: grad_scale and found_inf are required");
    return adam_step_impl(tensors, n_tensors, groups, n_groups, grad_scale, found_inf, stream);
}

static int adam_step_impl(const papr_adam_tensor* tensors, int32_t n_tensors, const papr_adam_group* groups, int32_t n_groups,
                          const float* grad_scale, float* found_inf, papr_stream_t stream) {
    PAPR_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || tensors) && groups && n_groups >= 1 && n_groups <= PAPR_ADAM_MAX_GROUPS,
                 "papr_adam_step: %d tensors, %d groups (at most %d)", n_tensors, n_groups, PAPR_ADAM_MAX_GROUPS);
    for (int i = 0; i < n_tensors; ++i) {
        const papr_adam_tensor& t = tensors[i];
        PAPR_REQUIRE(t.p && t.g && t.m && t.v && t.step && t.n >= 0 && t.group >= 0 && t.group < n_groups, "papr_adam_step: tensor %d: null pointer or bad group", i);
    }
    hipStream_t s = as_stream(stream);
    if (found_inf) {                                 // every gradient of the call is looked at before any tensor moves
        for (int i = 0; i < n_groups; ++i)
            PAPR_REQUIRE(groups[i].found_slot >= 0 && groups[i].found_slot < PAPR_ADAM_MAX_GROUPS, "papr_adam_step_scaled: group %d: found_slot %d", i, groups[i].found_slot);
        PAPR_REQUIRE(hipMemsetAsync(found_inf, 0, (PAPR_ADAM_MAX_GROUPS + 1) * sizeof(float), s) == hipSuccess, "papr_adam_step_scaled: memset failed");
        for (int first = 0; first < n_tensors; first += PAPR_ADAM_MAX_TENSORS) {
            AdamBatch b;
            const int n = n_tensors - first < PAPR_ADAM_MAX_TENSORS ? n_tensors - first : PAPR_ADAM_MAX_TENSORS;
            long most = 0;
            for (int i = 0; i < n; ++i) { b.t[i] = tensors[first + i]; most = b.t[i].n > most ? b.t[i].n : most; }
            for (int i = 0; i < n_groups; ++i) b.g[i] = groups[i];
            if (most == 0) continue;
            adam_check_kernel<<<dim3((unsigned)((most + ADAM_CHUNK - 1) / ADAM_CHUNK), (unsigned)n), dim3(256), 0, s>>>(b, found_inf);
            PAPR_CHECK_LAUNCH("adam_check");
        }
    }
    for (int first = 0; first < n_tensors; first += PAPR_ADAM_MAX_TENSORS) {
        AdamBatch b;
        const int n = n_tensors - first < PAPR_ADAM_MAX_TENSORS ? n_tensors - first : PAPR_ADAM_MAX_TENSORS;
        long most = 0;
        for (int i = 0; i < n; ++i) { b.t[i] = tensors[first + i]; most = b.t[i].n > most ? b.t[i].n : most; }
        for (int i = 0; i < n_groups; ++i) b.g[i] = groups[i];
        adam_bump_steps_kernel<<<dim3(1), dim3(64), 0, s>>>(b, n, found_inf);
        PAPR_CHECK_LAUNCH("adam_bump_steps");
        if (most == 0) continue;
        adam_step_kernel<<<dim3((unsigned)((most + ADAM_CHUNK - 1) / ADAM_CHUNK), (unsigned)n), dim3(256), 0, s>>>(b, grad_scale, found_inf);
        PAPR_CHECK_LAUNCH("adam_step");
    }
    return 0;
}
