// optim.hip -- fused Adam step (SURVEY.md "next" N3) for gfx950.
//
// One pass per parameter tensor: grad *= grad_scale (the reference's `grad /= bsz`,
// train_internal.py:319-324), first/second moment update, bias-corrected step -- the dense semantics of
// stock torch.optim.Adam as the reference configures it (scene/gaussian_model.py:292: lr per group,
// eps 1e-15, no weight decay, no amsgrad).  Moments of Gaussians that were invisible this iteration
// (zero gradient) still decay and still move the parameter, exactly like the stock optimizer.
// Pure HBM streaming: 16 B read + 12 B written per element (float4 vectorised).
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
adam_kernel(long long n, float *__restrict__ param, const float *__restrict__ grad, float *__restrict__ exp_avg,
            float *__restrict__ exp_avg_sq, float lr_c, float b1, float b2, float omb1, float omb2,
            float inv_sqrt_bc2, float eps, float grad_scale) {
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        float4 p = *reinterpret_cast<float4 *>(param + i4);
        float4 g = *reinterpret_cast<const float4 *>(grad + i4);
        float4 m = *reinterpret_cast<float4 *>(exp_avg + i4);
        float4 v = *reinterpret_cast<float4 *>(exp_avg_sq + i4);
        gsr_adam1(p.x, g.x * grad_scale, m.x, v.x, lr_c, b1, b2, omb1, omb2, inv_sqrt_bc2, eps);
        gsr_adam1(p.y, g.y * grad_scale, m.y, v.y, lr_c, b1, b2, omb1, omb2, inv_sqrt_bc2, eps);
        gsr_adam1(p.z, g.z * grad_scale, m.z, v.z, lr_c, b1, b2, omb1, omb2, inv_sqrt_bc2, eps);
        gsr_adam1(p.w, g.w * grad_scale, m.w, v.w, lr_c, b1, b2, omb1, omb2, inv_sqrt_bc2, eps);
        *reinterpret_cast<float4 *>(param + i4) = p;
        *reinterpret_cast<float4 *>(exp_avg + i4) = m;
        *reinterpret_cast<float4 *>(exp_avg_sq + i4) = v;
    } else {
        for (long long i = i4; i < n; i++) {
            float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
            gsr_adam1(p, grad[i] * grad_scale, m, v, lr_c, b1, b2, omb1, omb2, inv_sqrt_bc2, eps);
            param[i] = p;
            exp_avg[i] = m;
            exp_avg_sq[i] = v;
        }
    }
}

// All parameter tensors of the model in ONE launch (six launches of 50 us each otherwise: the per-launch tails and
// the ~2 us boundaries between them are ~8 % of the optimizer).  Blocks are assigned to tensors by a prefix table.
constexpr int ADAM_MAX_TENSORS = 16;
struct AdamBatch {
    float *param[ADAM_MAX_TENSORS];
    const float *grad[ADAM_MAX_TENSORS];
    float *m[ADAM_MAX_TENSORS];
    float *v[ADAM_MAX_TENSORS];
    long long n[ADAM_MAX_TENSORS];
    int block0[ADAM_MAX_TENSORS + 1];  // first block of each tensor
    float lr_c[ADAM_MAX_TENSORS], b1[ADAM_MAX_TENSORS], b2[ADAM_MAX_TENSORS], omb1[ADAM_MAX_TENSORS],
        omb2[ADAM_MAX_TENSORS], inv_sqrt_bc2[ADAM_MAX_TENSORS], eps[ADAM_MAX_TENSORS];
    int count;
};
constexpr int ADAM_ELEMS_PER_BLOCK = 256 * 4 * 4;  // 256 threads x 4 float4 each

__global__ void __launch_bounds__(256) adam_multi_kernel(AdamBatch a, float grad_scale) {
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.block0[t + 1]) t++;  // <= 16 uniform compares
    const long long n = a.n[t];
    float *__restrict__ param = a.param[t];
    const float *__restrict__ grad = a.grad[t];
    float *__restrict__ em = a.m[t];
    float *__restrict__ ev = a.v[t];
    const float lr_c = a.lr_c[t], b1 = a.b1[t], b2 = a.b2[t], omb1 = a.omb1[t], omb2 = a.omb2[t],
                inv_sqrt_bc2 = a.inv_sqrt_bc2[t], eps = a.eps[t];
    const long long base = (long long)((int)blockIdx.x - a.block0[t]) * ADAM_ELEMS_PER_BLOCK;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const long long i4 = base + ((long long)r * 256 + threadIdx.x) * 4;
        if (i4 + 3 < n) {
            // streaming hints: the gradient and both moments are touched exactly once per step and nothing reads them
            // before the next step (measured: 0.303 -> 0.252 ms for the 59 M elements of a 1 M-Gaussian model,
            // 5.5 -> 6.6 TB/s); the parameters stay cacheable -- 236 MB of them fit the 256 MB memory-side cache and
            // the next iteration's K1 reads them from there (0.075 -> 0.056 ms; streaming them too was slower)
            typedef float vf4 __attribute__((ext_vector_type(4)));
            vf4 pn = *reinterpret_cast<vf4 *>(param + i4);
            vf4 gn = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(grad + i4));
            vf4 mn = __builtin_nontemporal_load(reinterpret_cast<vf4 *>(em + i4));
            vf4 vn = __builtin_nontemporal_load(reinterpret_cast<vf4 *>(ev + i4));
            float4 p = make_float4(pn.x, pn.y, pn.z, pn.w), g = make_float4(gn.x, gn.y, gn.z, gn.w);
            float4 m = make_float4(mn.x, mn.y, mn.z, mn.w), v = make_float4(vn.x, vn.y, vn.z, vn.w);
            gsr_adam1(p.x, g.x * grad_scale, m.x, v.x, lr_c, b1, b2, omb1, omb2, inv_sqrt_bc2, eps);
            gsr_adam1(p.y, g.y * grad_scale, m.y, v.y, lr_c, b1, b2, omb1, omb2, inv_sqrt_bc2, eps);
            gsr_adam1(p.z, g.z * grad_scale, m.z, v.z, lr_c, b1, b2, omb1, omb2, inv_sqrt_bc2, eps);
            gsr_adam1(p.w, g.w * grad_scale, m.w, v.w, lr_c, b1, b2, omb1, omb2, inv_sqrt_bc2, eps);
            *reinterpret_cast<float4 *>(param + i4) = p;
            __builtin_nontemporal_store(vf4{m.x, m.y, m.z, m.w}, reinterpret_cast<vf4 *>(em + i4));
            __builtin_nontemporal_store(vf4{v.x, v.y, v.z, v.w}, reinterpret_cast<vf4 *>(ev + i4));
        } else {
            for (long long i = i4; i < n && i < i4 + 4; i++) {
                float p = param[i], m = em[i], v = ev[i];
                gsr_adam1(p, grad[i] * grad_scale, m, v, lr_c, b1, b2, omb1, omb2, inv_sqrt_bc2, eps);
                param[i] = p;
                em[i] = m;
                ev[i] = v;
            }
        }
    }
}

}  // namespace

extern "C" int gsr_adam_step_multi(int num_tensors, const int64_t *numels, float *const *params,
                                   const float *const *grads, float *const *exp_avgs, float *const *exp_avg_sqs,
                                   const double *lrs, const double *beta1s, const double *beta2s, const double *epss,
                                   const int64_t *steps, float grad_scale, gsr_stream_t stream) {
    if (num_tensors < 0 || num_tensors > ADAM_MAX_TENSORS) return GSR_EINVAL;
    if (num_tensors == 0) return 0;
    if (!numels || !params || !grads || !exp_avgs || !exp_avg_sqs || !lrs || !beta1s || !beta2s || !epss || !steps)
        return GSR_EINVAL;
    AdamBatch a{};
    int blocks = 0, k = 0;
    for (int t = 0; t < num_tensors; t++) {
        if (numels[t] < 0 || steps[t] < 1) return GSR_EINVAL;
        if (numels[t] == 0) continue;
        if (!params[t] || !grads[t] || !exp_avgs[t] || !exp_avg_sqs[t]) return GSR_EINVAL;
        if (((uintptr_t)params[t] | (uintptr_t)grads[t] | (uintptr_t)exp_avgs[t] | (uintptr_t)exp_avg_sqs[t]) & 15)
            return GSR_EINVAL;
        const double bc1 = 1.0 - pow(beta1s[t], (double)steps[t]);
        const double bc2 = 1.0 - pow(beta2s[t], (double)steps[t]);
        a.param[k] = params[t];
        a.grad[k] = grads[t];
        a.m[k] = exp_avgs[t];
        a.v[k] = exp_avg_sqs[t];
        a.n[k] = numels[t];
        a.block0[k] = blocks;
        a.lr_c[k] = (float)(lrs[t] / bc1);
        a.b1[k] = (float)beta1s[t];
        a.b2[k] = (float)beta2s[t];
        a.omb1[k] = (float)(1.0 - beta1s[t]);
        a.omb2[k] = (float)(1.0 - beta2s[t]);
        a.inv_sqrt_bc2[k] = (float)(1.0 / sqrt(bc2));
        a.eps[k] = (float)epss[t];
        const long long nb = (numels[t] + ADAM_ELEMS_PER_BLOCK - 1) / ADAM_ELEMS_PER_BLOCK;
        if (blocks + nb > 0x7fffffffLL) return GSR_EINVAL;
        blocks += (int)nb;
        k++;
    }
    if (k == 0) return 0;
    a.block0[k] = blocks;
    a.count = k;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a,
                       grad_scale);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_adam_step(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, double lr,
                             double beta1, double beta2, double eps, int64_t step, float grad_scale,
                             gsr_stream_t stream) {
    if (n < 0 || step < 1) return GSR_EINVAL;
    if (n == 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return GSR_EINVAL;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return GSR_EINVAL;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float lr_c = (float)(lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const long long groups = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3(gsr_div_up(groups, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (long long)n, param, grad, exp_avg, exp_avg_sq, lr_c, (float)beta1, (float)beta2,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), inv_sqrt_bc2, (float)eps, grad_scale);
    GSR_LAUNCH_CHECK();
    return 0;
}
