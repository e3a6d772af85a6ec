// Optimizer step over a flat fp32 parameter arena: global grad-norm + clip + Adam in two launches.
// Replaces train.py:81-85 of the reference (clip_grad_norm_ over model.parameters(), then
// bitsandbytes Adam8bit.step()).  Two state formats: fp32 moments (sp_adam_step_f32; the default -- 288 GB
// do not need the saving) and bitsandbytes-style block-wise 8-bit moments (sp_adam8_step_f32: blocks of 2048
// elements, one f32 absmax per block and moment, a 256-entry dynamic-exponent code table, signed for the
// first moment and unsigned for the second; parity unpinned -- bitsandbytes is not installed here, the table
// is built from its published construction).  Weight decay can be
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

// ---- block-wise 8-bit moments -------------------------------------------------------------------
// Code tables (bitsandbytes create_dynamic_map(signed, max_exponent_bits = 7, total_bits = 8)): for every
// exponent i = 0..6 the midpoints of linspace(0.1, 1, 2^i + 1) (signed; 2^(i+1) + 1 unsigned) times 10^(i-6),
// both signs when signed, plus 0 and 1.  Sorted, the signed table is 127 negatives, 0, 127 positives, 1; the
// unsigned one 0, 254 positives, 1.  Entry i in closed form:
constexpr int kQBlock = 2048;
__device__ __forceinline__ float q_pos(int e, int j, int cnt) {     // midpoint j of `cnt` intervals of [0.1, 1], exponent e
    const float p10[7] = {1e-6f, 1e-5f, 1e-4f, 1e-3f, 1e-2f, 1e-1f, 1.f};
    return p10[e] * (0.1f + 0.9f * ((float)j + 0.5f) / (float)cnt);
}
__device__ __forceinline__ float code_signed(int i) {
    if (i == 127) return 0.f;
    if (i == 255) return 1.f;
    const int q = i < 127 ? 126 - i : i - 128;           // 0 .. 126, ascending magnitude
    const int e = 31 - __clz(q + 1), j = q + 1 - (1 << e);
    const float v = q_pos(e, j, 1 << e);
    return i < 127 ? -v : v;
}
__device__ __forceinline__ float code_unsigned(int i) {
    if (i == 0) return 0.f;
    if (i == 255) return 1.f;
    const int q = i - 1;                                  // 0 .. 253
    const int e = 30 - __clz(q + 2), j = q + 2 - (2 << e);
    return q_pos(e, j, 2 << e);
}
// index of the table entry nearest to x (ties to the lower entry); t ascending, 256 entries, in LDS
__device__ __forceinline__ int q_nearest(const float* t, float x) {
    int lo = 0;
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) lo += (t[lo + s] <= x) ? s : 0;     // largest lo with t[lo] <= x (t[0] <= x assumed)
    const int hi = min(lo + 1, 255);
    return (x - t[lo] <= t[hi] - x) ? lo : hi;
}

__global__ __launch_bounds__(256) void adam8_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     uint8_t* __restrict__ q1, uint8_t* __restrict__ q2,
                                                     float* __restrict__ absmax1, float* __restrict__ absmax2,
                                                     int64_t n, AdamArgs a, const float* __restrict__ sumsq) {
    __shared__ float c1[256], c2[256], red[2][4];
    const int tid = threadIdx.x;
    c1[tid] = code_signed(tid);
    c2[tid] = code_unsigned(tid);
    float gscale = 1.f;
    if (sumsq && a.max_norm > 0.f) {
        const float norm = sqrtf(sumsq[0]) / fabsf(a.grad_div);
        gscale = fminf(1.f, a.max_norm / (norm + 1e-6f));
    }
    gscale /= a.grad_div;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kQBlock + tid * 8;
    const float am1 = absmax1[blockIdx.x], am2 = absmax2[blockIdx.x];
    float pp[8], gg[8], mm[8], vv[8];
    const bool full = base + 8 <= n;          // n is a multiple of 4 and the buffers are 16-byte aligned
    uint2 b1 = make_uint2(0x7f7f7f7fu, 0x7f7f7f7fu), b2 = make_uint2(0, 0);
    if (full) {
        const float4 pa = *reinterpret_cast<const float4*>(p + base), pb = *reinterpret_cast<const float4*>(p + base + 4);
        const float4 ga = *reinterpret_cast<const float4*>(g + base), gb = *reinterpret_cast<const float4*>(g + base + 4);
        pp[0] = pa.x; pp[1] = pa.y; pp[2] = pa.z; pp[3] = pa.w; pp[4] = pb.x; pp[5] = pb.y; pp[6] = pb.z; pp[7] = pb.w;
        gg[0] = ga.x; gg[1] = ga.y; gg[2] = ga.z; gg[3] = ga.w; gg[4] = gb.x; gg[5] = gb.y; gg[6] = gb.z; gg[7] = gb.w;
        b1 = *reinterpret_cast<const uint2*>(q1 + base);
        b2 = *reinterpret_cast<const uint2*>(q2 + base);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool ok = base + k < n;
            pp[k] = ok ? p[base + k] : 0.f;
            gg[k] = ok ? g[base + k] : 0.f;
            if (ok) {
                (k < 4 ? b1.x : b1.y) = ((k < 4 ? b1.x : b1.y) & ~(0xffu << (8 * (k & 3)))) | ((uint32_t)q1[base + k] << (8 * (k & 3)));
                (k < 4 ? b2.x : b2.y) |= (uint32_t)q2[base + k] << (8 * (k & 3));
            }
        }
    }
    float mx1 = 0.f, mx2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t w1 = k < 4 ? b1.x : b1.y, w2 = k < 4 ? b2.x : b2.y;
        mm[k] = c1[(w1 >> (8 * (k & 3))) & 255] * am1;
        vv[k] = c2[(w2 >> (8 * (k & 3))) & 255] * am2;
        if (base + k < n) adam_one(pp[k], gg[k], mm[k], vv[k], a, gscale);
        mx1 = fmaxf(mx1, fabsf(mm[k]));
        mx2 = fmaxf(mx2, vv[k]);
    }
    mx1 = wave_max(mx1);
    mx2 = wave_max(mx2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = mx1; red[1][tid >> 6] = mx2; }
    __syncthreads();
    const float n1 = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    const float n2 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    const float i1 = n1 > 0.f ? 1.f / n1 : 0.f, i2 = n2 > 0.f ? 1.f / n2 : 0.f;
    uint2 o1 = make_uint2(0, 0), o2 = make_uint2(0, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t e1 = (uint32_t)q_nearest(c1, fmaxf(mm[k] * i1, c1[0]));
        const uint32_t e2 = (uint32_t)q_nearest(c2, vv[k] * i2);
        (k < 4 ? o1.x : o1.y) |= e1 << (8 * (k & 3));
        (k < 4 ? o2.x : o2.y) |= e2 << (8 * (k & 3));
    }
    if (full) {
        *reinterpret_cast<float4*>(p + base) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        *reinterpret_cast<float4*>(p + base + 4) = make_float4(pp[4], pp[5], pp[6], pp[7]);
        *reinterpret_cast<uint2*>(q1 + base) = o1;
        *reinterpret_cast<uint2*>(q2 + base) = o2;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (base + k < n) {
                p[base + k] = pp[k];
                q1[base + k] = (uint8_t)(((k < 4 ? o1.x : o1.y) >> (8 * (k & 3))) & 255);
                q2[base + k] = (uint8_t)(((k < 4 ? o2.x : o2.y) >> (8 * (k & 3))) & 255);
            }
    }
    if (tid == 0) { absmax1[blockIdx.x] = n1; absmax2[blockIdx.x] = n2; }
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

extern "C" int sp_adam8_step_f32(float* p, const float* g, uint8_t* state1, uint8_t* state2, float* absmax1,
                                 float* absmax2, int64_t n, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int step, int decoupled, const float* sumsq, float max_norm,
                                 float grad_div, void* stream) {
    using namespace sp;
    SP_CHECK_ARG(n >= 0 && step >= 1 && n % 4 == 0, "sp_adam8_step_f32: bad n/step (n must be a multiple of 4)");
    if (n == 0) return SP_OK;
    SP_CHECK_ARG(p && g && state1 && state2 && absmax1 && absmax2, "sp_adam8_step_f32: null pointer");
    SP_CHECK_ARG((((uintptr_t)p | (uintptr_t)g) & 15) == 0 && (((uintptr_t)state1 | (uintptr_t)state2) & 7) == 0,
                 "sp_adam8_step_f32: p, g must be 16-byte aligned, the states 8-byte");
    SP_CHECK_ARG(grad_div != 0.f, "sp_adam8_step_f32: grad_div must be non-zero");
    const int64_t blocks = (n + kQBlock - 1) / kQBlock;
    SP_CHECK_ARG(blocks < ((int64_t)1 << 31), "sp_adam8_step_f32: too many blocks");
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = 1.f - std::pow(beta1, (float)step);
    a.bc2 = 1.f - std::pow(beta2, (float)step);
    a.max_norm = max_norm; a.grad_div = grad_div; a.decoupled = decoupled;
    hipLaunchKernelGGL(adam8_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       p, g, state1, state2, absmax1, absmax2, n, a, sumsq);
    return check_launch("adam8_kernel");
}
