// Optimizer step over a flat fp32 parameter arena: global grad-norm + clip + Adam in two launches.
// Replaces train.py:81-85 of the reference (clip_grad_norm_ over model.parameters(), then
// bitsandbytes Adam8bit.step()).  State is kept in fp32 (the 8-bit block-wise state of
// bitsandbytes is a memory optimisation the 288 GB part does not need); weight decay can be
// decoupled (p *= 1 - lr*wd, what the bitsandbytes kernels do) or L2 (g += wd*p).
// Pure HBM streaming: 16 B per lane, grid-stride, no host synchronisation -- the clip
// coefficient is read from device memory.
#include "common.h"

namespace sp {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n,
                                                     float* __restrict__ out) {
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = x4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = x[(n4 << 2) + threadIdx.x];
        acc += v * v;
    }
    acc = wave_sum(acc);
    __shared__ float s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    // one partial per workgroup, summed in a fixed order by sumsq_final_kernel: no float atomics, so
    // every data-parallel replica derives the same clip coefficient from the same reduced gradient
    if (threadIdx.x == 0) out[1 + blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(float* __restrict__ out, int nparts) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += out[1 + i];
    acc = wave_sum(acc);
    __shared__ float s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

struct AdamArgs {
    float lr, beta1, beta2, eps, wd, bc1, bc2, max_norm, grad_div;
    int decoupled;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a,
                                         float gscale) {
    g *= gscale;
    if (!a.decoupled) g += a.wd * p;
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float mh = m / a.bc1, vh = v / a.bc2;
    float pn = p - a.lr * mh / (sqrtf(vh) + a.eps);
    if (a.decoupled) pn *= (1.f - a.lr * a.wd);
    p = pn;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    int64_t n, AdamArgs a,
                                                    const float* __restrict__ sumsq) {
    // clip coefficient as torch.nn.utils.clip_grad_norm_: max_norm / (norm + 1e-6), capped at 1, where
    // the norm is that of the gradient the update USES, g / grad_div (data parallel: the buffer holds the
    // SUM over the ranks and grad_div = world, so the mean gradient is what gets clipped -- the same
    // update whatever the world size).
    float gscale = 1.f;
    if (sumsq && a.max_norm > 0.f) {
        const float norm = sqrtf(sumsq[0]) / fabsf(a.grad_div);
        gscale = fminf(1.f, a.max_norm / (norm + 1e-6f));
    }
    gscale /= a.grad_div;
    const int64_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, a, gscale);
        adam_one(pp.y, gg.y, mm.y, vv.y, a, gscale);
        adam_one(pp.z, gg.z, mm.z, vv.z, a, gscale);
        adam_one(pp.w, gg.w, mm.w, vv.w, a, gscale);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        adam_one(p[i], g[i], m[i], v[i], a, gscale);
    }
}

}  // namespace sp

extern "C" int sp_sumsq_f32(const float* x, int64_t n, float* out, void* stream) {
    using namespace sp;
    SP_CHECK_ARG(n >= 0 && out, "sp_sumsq_f32: bad arguments");
    SP_CHECK_ARG(((uintptr_t)x & 15) == 0, "sp_sumsq_f32: x must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > SP_SUMSQ_WORDS - 1) blocks = SP_SUMSQ_WORDS - 1;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3((int)blocks), dim3(256), 0, s, x, n, out);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, out, (int)blocks);
    return check_launch("sumsq_kernel");
}

extern "C" int sp_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int step,
                                int decoupled, const float* sumsq, float max_norm, float grad_div,
                                void* stream) {
    using namespace sp;
    SP_CHECK_ARG(n >= 0 && step >= 1, "sp_adam_step_f32: bad n/step");
    if (n == 0) return SP_OK;
    SP_CHECK_ARG(p && g && m && v, "sp_adam_step_f32: null pointer");
    SP_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                 "sp_adam_step_f32: buffers must be 16-byte aligned");
    SP_CHECK_ARG(grad_div != 0.f, "sp_adam_step_f32: grad_div must be non-zero");
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = 1.f - std::pow(beta1, (float)step);
    a.bc2 = 1.f - std::pow(beta2, (float)step);
    a.max_norm = max_norm; a.grad_div = grad_div; a.decoupled = decoupled;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       p, g, m, v, n, a, sumsq);
    return check_launch("adam_kernel");
}
