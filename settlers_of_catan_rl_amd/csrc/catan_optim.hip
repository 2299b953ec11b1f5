// catan_optim.hip - the optimiser step of PPO.update (RL/ppo/ppo.py:23 `optim.Adam(lr, eps)`, :67-68 `clip_grad_norm_(max_grad_norm)`
// then `optimizer.step()`) over ALL parameters of the net in two launches.
//
// torch's own path is clip_grad_norm_ (a _foreach_norm, a stack + norm, a _foreach_mul_) + the nine _foreach_* launches of Adam over
// ~300 tensors: ~0.55 ms of a 22.8 ms minibatch step for 1.93 M parameters - 54 MB of traffic, ~15 us at the HBM rate.  Here the
// parameters are cut into chunks of OPT_CHUNK elements (a chunk never spans two tensors; table on the device, built once):
//   k_grad_sumsq   one workgroup per chunk: the chunk's sum of squared gradients (fp32 products, fp64 sum) -> partial[chunk]
//   k_adam_step    every workgroup adds the partials in index order (fp64: the same total in every workgroup, whatever the launch
//                  order), forms clip_grad_norm_'s coefficient min(max_norm / (norm + 1e-6), 1) and applies torch.optim.Adam's
//                  update in the operand order of its _foreach_ form (lerp, mul + addcmul, sqrt / div + eps, addcdiv) to its chunk.
// A tensor whose gradient pointer is null takes no step and adds nothing to the norm (torch skips parameters without a gradient).
#pragma once
#include <hip/hip_runtime.h>

namespace catan {

constexpr int OPT_CHUNK = 2048, OPT_BLOCK = 256;
struct AdamTensor { float* p; float* m; float* v; };
struct AdamChunk { int tensor; int count; long offset; };

__global__ __launch_bounds__(OPT_BLOCK) void k_grad_sumsq(const AdamChunk* __restrict__ chunks, const float* const* __restrict__ grads, double* __restrict__ partial) {
    __shared__ double sh[OPT_BLOCK / 64];
    const AdamChunk c = chunks[blockIdx.x];
    const float* __restrict__ g = grads[c.tensor];
    double s = 0.0;
    if (g != nullptr) {
        g += c.offset;
        const int n4 = c.count >> 2;
        for (int i = threadIdx.x; i < n4; i += OPT_BLOCK) {
            const float4 x = reinterpret_cast<const float4*>(g)[i];
            s += (double)(x.x * x.x) + (double)(x.y * x.y) + (double)(x.z * x.z) + (double)(x.w * x.w);
        }
        for (int i = (n4 << 2) + threadIdx.x; i < c.count; i += OPT_BLOCK) s += (double)(g[i] * g[i]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < OPT_BLOCK / 64; w++) t += sh[w]; partial[blockIdx.x] = t; }
}

struct AdamHyper { float max_norm, lr, beta1, beta2, eps, bias1, bias2_sqrt; int clip; };

DEVI void adam_one(float& p, float& m, float& v, float g, float coef, const AdamHyper& h, float step_size) {
    g = g * coef;                                                   // clip_grad_norm_: grads.mul_(clip_coef_clamped)
    m = m + (g - m) * (1.0f - h.beta1);                             // exp_avg.lerp_(grad, 1 - beta1)
    v = v * h.beta2 + (1.0f - h.beta2) * g * g;                     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / h.bias2_sqrt + h.eps;            // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - step_size * (m / denom);                                // param.addcdiv_(exp_avg, denom, value = -step_size)
}

__global__ __launch_bounds__(OPT_BLOCK) void k_adam_step(const AdamTensor* __restrict__ tensors, const AdamChunk* __restrict__ chunks, int n_chunks,
                                                         const float* const* __restrict__ grads, const double* __restrict__ partial, AdamHyper h,
                                                         float* __restrict__ norm_out) {
    __shared__ double sh[OPT_BLOCK / 64];
    __shared__ float s_coef;
    double s = 0.0;
    // (index order inside a lane, a fixed tree across lanes and waves: the same value in every workgroup and every run)
    for (int i = threadIdx.x; i < n_chunks; i += OPT_BLOCK) s += partial[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < OPT_BLOCK / 64; w++) t += sh[w];
        const float norm = (float)sqrt(t);
        float coef = 1.0f;
        if (h.clip) { coef = h.max_norm / (norm + 1e-6f); coef = coef < 1.0f ? coef : 1.0f; }      // torch.nn.utils.clip_grad_norm_
        s_coef = coef;
        if (blockIdx.x == 0 && norm_out != nullptr) *norm_out = norm;
    }
    __syncthreads();
    const float coef = s_coef;
    const AdamChunk c = chunks[blockIdx.x];
    const float* __restrict__ g = grads[c.tensor];
    if (g == nullptr) return;
    const AdamTensor t = tensors[c.tensor];
    g += c.offset;
    float* __restrict__ p = t.p + c.offset; float* __restrict__ m = t.m + c.offset; float* __restrict__ v = t.v + c.offset;
    const float step_size = h.lr / h.bias1;
    const int n4 = c.count >> 2;
    for (int i = threadIdx.x; i < n4; i += OPT_BLOCK) {
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pv.x, mv.x, vv.x, gv.x, coef, h, step_size); adam_one(pv.y, mv.y, vv.y, gv.y, coef, h, step_size);
        adam_one(pv.z, mv.z, vv.z, gv.z, coef, h, step_size); adam_one(pv.w, mv.w, vv.w, gv.w, coef, h, step_size);
        reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < c.count; i += OPT_BLOCK) adam_one(p[i], m[i], v[i], g[i], coef, h, step_size);
}

}  // namespace catan
