// adam.hip -- the optimizer step of the reference's training loop (main.py:104 `optimizer.step()` on the
// torch.optim.Adam of utils.py:163-172) for every parameter tensor of both fields in ONE launch.
//
// torch.optim.Adam walks the 48 small tensors through multi-tensor kernels in chunks (two launches of ~45 us for
// 1.2 M parameters, plus ~0.2 ms of host-side bookkeeping per step); here the pointer table travels in the kernel
// arguments and one grid covers all tensors.  HBM-bound: 28 B per parameter (p, g, m, v read; p, m, v written) =
// 33 MB per step of both fields.
//
// Arithmetic (fp32, the order of torch/optim/adam.py _single_tensor_adam):
//   m = m + (1 - beta1) (g - m)                      (Tensor.lerp_)
//   v = v beta2 + (1 - beta2) g g                    (mul_, addcmul_)
//   p = p - step_size * m / (sqrt(v) / sqrt(bias_correction2) + eps),   step_size = lr / bias_correction1
// with g = g + weight_decay p first when weight_decay != 0.  bias corrections are computed by the caller in double.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace na {

constexpr int ADAM_MAX_TENSORS = 64;      // per launch (kernel arguments are limited to 4 KiB)
constexpr int ADAM_CHUNK = 2048;          // elements per block: 256 threads x 8

struct AdamTable {
    float *p[ADAM_MAX_TENSORS];
    const float *g[ADAM_MAX_TENSORS];
    float *m[ADAM_MAX_TENSORS];
    float *v[ADAM_MAX_TENSORS];
    int first_block[ADAM_MAX_TENSORS + 1];     // blocks [first_block[t], first_block[t+1]) cover tensor t
    int numel[ADAM_MAX_TENSORS];
    int n;
};

// The step-dependent scalars of a CAPTURED step (a HIP graph replays the kernel arguments it was captured with): the step
// count and the learning rate live in device memory; one thread advances the count and leaves step_size = lr /
// bias_correction1 and sqrt(bias_correction2) for the update kernel, in double like the host path.
__global__ void adam_scalars_kernel(int64_t *step, const double *lr, double beta1, double beta2, float *scalars) {
    const int64_t s = *step + 1;
    *step = s;
    const double bc1 = 1.0 - pow(beta1, (double)s), bc2 = 1.0 - pow(beta2, (double)s);
    scalars[0] = (float)(*lr / bc1);
    scalars[1] = (float)sqrt(bc2);
}

__global__ __launch_bounds__(256) void adam_step_kernel(AdamTable T, float step_size, float w1, float beta2, float w2, float eps,
                                                        float weight_decay, float bc2_sqrt, const float *dev_scalars) {
    if (dev_scalars) { step_size = dev_scalars[0]; bc2_sqrt = dev_scalars[1]; }
    int t = 0;                                  // uniform per block: scalar search
    while (t + 1 < T.n && (int)blockIdx.x >= T.first_block[t + 1]) ++t;
    const int base = ((int)blockIdx.x - T.first_block[t]) * ADAM_CHUNK;
    const int n = T.numel[t];
    float *__restrict__ p = T.p[t];
    const float *__restrict__ g = T.g[t];
    float *__restrict__ m = T.m[t];
    float *__restrict__ v = T.v[t];
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const int i = base + k * 256 + (int)threadIdx.x;
        if (i >= n) break;
        float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
        if (weight_decay != 0.0f) gi = gi + weight_decay * pi;
        mi = mi + w1 * (gi - mi);
        vi = vi * beta2 + w2 * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

int launch_adam(int n, float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                const int64_t *numel, float step_size, double beta1, double beta2, float eps, float weight_decay,
                float bc2_sqrt, hipStream_t s, int64_t *step_dev, const double *lr_dev, float *scalars_dev) {
    if (scalars_dev) hipLaunchKernelGGL(adam_scalars_kernel, dim3(1), dim3(1), 0, s, step_dev, lr_dev, beta1, beta2, scalars_dev);
    // 1 - beta in double, as torch evaluates the Python scalar (1.0f - 0.999f is off by 5e-5 relative)
    const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2);
    for (int t0 = 0; t0 < n; t0 += ADAM_MAX_TENSORS) {
        AdamTable T;
        T.n = n - t0 < ADAM_MAX_TENSORS ? n - t0 : ADAM_MAX_TENSORS;
        int blocks = 0;
        for (int t = 0; t < T.n; ++t) {
            T.p[t] = params[t0 + t]; T.g[t] = grads[t0 + t]; T.m[t] = exp_avg[t0 + t]; T.v[t] = exp_avg_sq[t0 + t];
            T.numel[t] = (int)numel[t0 + t];
            T.first_block[t] = blocks;
            blocks += (int)((numel[t0 + t] + ADAM_CHUNK - 1) / ADAM_CHUNK);
        }
        T.first_block[T.n] = blocks;
        if (blocks > 0)
            hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, T, step_size, w1, (float)beta2, w2, eps,
                               weight_decay, bc2_sqrt, (const float *)scalars_dev);
    }
    return hipGetLastError() == hipSuccess ? NERF_AMD_OK : NERF_AMD_EHIP;
}

}  // namespace na
