// f32 convolutions of the encoder's conv stack for EVALUATION in fp32 (net.py:56-59 and :6-42 outside autocast, as
// train.py:105-135 and :162-168 run them): Conv2d(1,32,3,stride 2) and the six Conv2d(32,32,(3,5)) with the inference
// BatchNorm (running statistics) folded into a per-channel scale / shift, the residual add and the ReLU in the
// epilogue.  f32 operands, f32 accumulation on the vector ALUs in one fixed order per output -- this path exists so
// that an fp32 evaluation is an fp32 evaluation end to end (the training path's conv kernels keep bf16 activations);
// it is sized for correctness and a few milliseconds per batch, not tuned further.
//
// One thread = one output position (b, f, t) with all 32 output channels in registers; the whole weight tensor
// [KH][KW][Cin][32] sits in LDS (<= 60 KB) and is read as broadcast float4s; consecutive lanes take consecutive t.
#include "common.h"

namespace sp {

struct ConvF32Args {
    const float* x; const float* w; const float* scale; const float* shift; const float* res; float* out;
    int64_t xsb, xsf, xst, rsb, rsf, rst, osb, osf, ost;
    int B, Fin, Tin, Fout, Tout, KH, KW, stride, relu;
};

template <int CIN>
__global__ __launch_bounds__(256) void conv2d_f32_kernel(ConvF32Args g) {
    extern __shared__ __attribute__((aligned(16))) float sw[];      // [KH][KW][CIN][32]
    const int nw = g.KH * g.KW * CIN * 32;
    for (int i = threadIdx.x * 4; i < nw; i += 256 * 4)
        *reinterpret_cast<float4*>(sw + i) = *reinterpret_cast<const float4*>(g.w + i);
    __syncthreads();
    const int64_t npos = (int64_t)g.B * g.Fout * g.Tout;
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= npos) return;
    const int to = (int)(pos % g.Tout), fo = (int)((pos / g.Tout) % g.Fout), b = (int)(pos / ((int64_t)g.Tout * g.Fout));
    const int ph = g.KH / 2, pw = g.KW / 2;
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
    const float* xb = g.x + (int64_t)b * g.xsb;
    for (int kh = 0; kh < g.KH; ++kh) {
        const int fi = fo * g.stride + kh - ph;
        for (int kw = 0; kw < g.KW; ++kw) {
            const int ti = to * g.stride + kw - pw;
            const bool ok = fi >= 0 && fi < g.Fin && ti >= 0 && ti < g.Tin;
            const float* xp = xb + (int64_t)min(max(fi, 0), g.Fin - 1) * g.xsf + (int64_t)min(max(ti, 0), g.Tin - 1) * g.xst;
            const float4* wt = reinterpret_cast<const float4*>(sw + (kh * g.KW + kw) * CIN * 32);
            if constexpr (CIN == 1) {
                const float xv = ok ? xp[0] : 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 wv = wt[q];
                    acc[4 * q] = fmaf(xv, wv.x, acc[4 * q]);
                    acc[4 * q + 1] = fmaf(xv, wv.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(xv, wv.z, acc[4 * q + 2]);
                    acc[4 * q + 3] = fmaf(xv, wv.w, acc[4 * q + 3]);
                }
            } else {
#pragma unroll 2
                for (int ci = 0; ci < CIN; ci += 4) {
                    float4 xv = *reinterpret_cast<const float4*>(xp + ci);
                    if (!ok) xv = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float4 wv = wt[(ci + u) * 8 + q];
                            acc[4 * q] = fmaf(xs[u], wv.x, acc[4 * q]);
                            acc[4 * q + 1] = fmaf(xs[u], wv.y, acc[4 * q + 1]);
                            acc[4 * q + 2] = fmaf(xs[u], wv.z, acc[4 * q + 2]);
                            acc[4 * q + 3] = fmaf(xs[u], wv.w, acc[4 * q + 3]);
                        }
                    }
                }
            }
        }
    }
    float* op = g.out + (int64_t)b * g.osb + (int64_t)fo * g.osf + (int64_t)to * g.ost;
    const float* rp = g.res ? g.res + (int64_t)b * g.rsb + (int64_t)fo * g.rsf + (int64_t)to * g.rst : nullptr;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 sc = *reinterpret_cast<const float4*>(g.scale + 4 * q), sh = *reinterpret_cast<const float4*>(g.shift + 4 * q);
        float4 v = make_float4(fmaf(acc[4 * q], sc.x, sh.x), fmaf(acc[4 * q + 1], sc.y, sh.y), fmaf(acc[4 * q + 2], sc.z, sh.z),
                               fmaf(acc[4 * q + 3], sc.w, sh.w));
        if (rp) {
            const float4 r = *reinterpret_cast<const float4*>(rp + 4 * q);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (g.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        *reinterpret_cast<float4*>(op + 4 * q) = v;
    }
}

}  // namespace sp

extern "C" int sp_conv2d_f32(const float* x, int64_t xsb, int64_t xsf, int64_t xst, int Cin, const float* w, const float* scale,
                             const float* shift, const float* res, int64_t rsb, int64_t rsf, int64_t rst, int relu, int B,
                             int Fin, int Tin, int KH, int KW, int stride, float* out, int64_t osb, int64_t osf, int64_t ost,
                             void* stream) {
    using namespace sp;
    SP_CHECK_ARG(B >= 0 && Fin >= 1 && Tin >= 1, "sp_conv2d_f32: bad shape B=%d Fin=%d Tin=%d", B, Fin, Tin);
    SP_CHECK_ARG(Cin == 1 || Cin == 32, "sp_conv2d_f32: Cin %d (1 or 32)", Cin);
    SP_CHECK_ARG(KH >= 1 && KW >= 1 && (KH & 1) && (KW & 1) && KH * KW * Cin * 32 * 4 <= 61440, "sp_conv2d_f32: kernel %dx%d", KH, KW);
    SP_CHECK_ARG(stride == 1 || stride == 2, "sp_conv2d_f32: stride %d", stride);
    if (B == 0) return SP_OK;
    SP_CHECK_ARG(x && w && scale && shift && out, "sp_conv2d_f32: null pointer");
    SP_CHECK_ARG((((uintptr_t)w | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)out | (uintptr_t)res) & 15) == 0 &&
                     (osb | osf | ost) % 4 == 0 && (!res || (rsb | rsf | rst) % 4 == 0),
                 "sp_conv2d_f32: w / scale / shift / res / out need 16-byte alignment, their strides multiples of 4");
    SP_CHECK_ARG(Cin == 1 || ((((uintptr_t)x) & 15) == 0 && (xsb | xsf | xst) % 4 == 0), "sp_conv2d_f32: x alignment");
    ConvF32Args g;
    g.x = x; g.w = w; g.scale = scale; g.shift = shift; g.res = res; g.out = out;
    g.xsb = xsb; g.xsf = xsf; g.xst = xst; g.rsb = rsb; g.rsf = rsf; g.rst = rst; g.osb = osb; g.osf = osf; g.ost = ost;
    g.B = B; g.Fin = Fin; g.Tin = Tin; g.KH = KH; g.KW = KW; g.stride = stride; g.relu = relu;
    g.Fout = (Fin + 2 * (KH / 2) - KH) / stride + 1;
    g.Tout = (Tin + 2 * (KW / 2) - KW) / stride + 1;
    const int64_t npos = (int64_t)B * g.Fout * g.Tout;
    SP_CHECK_ARG((npos + 255) / 256 < ((int64_t)1 << 31), "sp_conv2d_f32: too many positions");
    const size_t lds = (size_t)KH * KW * Cin * 32 * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (Cin == 1) hipLaunchKernelGGL(conv2d_f32_kernel<1>, dim3((unsigned)((npos + 255) / 256)), dim3(256), lds, s, g);
    else hipLaunchKernelGGL(conv2d_f32_kernel<32>, dim3((unsigned)((npos + 255) / 256)), dim3(256), lds, s, g);
    return check_launch("conv2d_f32_kernel");
}
