// Row LayerNorm forward / backward for the SRU stack and classifier (net.py:61-76 of the reference:
// sru.SRU(layer_norm=True) normalises each layer's input, nn.LayerNorm precedes the classifier).
// One wave per row, the row lives in registers (<= 4 x 8 elements per lane, N <= 2048, N % 8 == 0),
// 16-byte loads/stores, two-pass statistics in f32.  The backward fuses the residual gradient that
// the SRU highway sends around the projection (gx = LN'(gy) + gres) and keeps per-lane partial
// sums of dgamma/dbeta over all rows a wave visits; one LDS reduction and one set of float atomics
// per block at the end.  HBM-bound: forward reads and writes the activation once.
#include "common.h"

namespace sp {

constexpr int kLnMaxChunks = 4;   // N <= 2048

template <typename T> struct Vec8;
template <> struct Vec8<__hip_bfloat16> {
    static __device__ __forceinline__ void ld(const __hip_bfloat16* p, float (&f)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ void st(__hip_bfloat16* p, const float (&f)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = __builtin_bit_cast(unsigned short, __float2bfloat16(f[2 * i]));
            const uint32_t hi = __builtin_bit_cast(unsigned short, __float2bfloat16(f[2 * i + 1]));
            w[i] = lo | (hi << 16);
        }
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct Vec8<__half> {
    static __device__ __forceinline__ void ld(const __half* p, float (&f)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __half2 h = __builtin_bit_cast(__half2, w[i]);
            f[2 * i] = __low2float(h);
            f[2 * i + 1] = __high2float(h);
        }
    }
    static __device__ __forceinline__ void st(__half* p, const float (&f)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = __builtin_bit_cast(uint32_t, __floats2half2_rn(f[2 * i], f[2 * i + 1]));
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct Vec8<float> {
    static __device__ __forceinline__ void ld(const float* p, float (&f)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&f)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
};

// Raw 8-element chunk as it sits in memory (prefetched a row ahead, converted when used).
template <typename T> struct Raw8;
template <> struct Raw8<__hip_bfloat16> {
    uint4 v;
    __device__ __forceinline__ void load(const __hip_bfloat16* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void zero() { v = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void get(float (&f)[8]) const {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
};
template <> struct Raw8<__half> {
    uint4 v;
    __device__ __forceinline__ void load(const __half* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void zero() { v = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void get(float (&f)[8]) const {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __half2 h = __builtin_bit_cast(__half2, w[i]);
            f[2 * i] = __low2float(h);
            f[2 * i + 1] = __high2float(h);
        }
    }
};
template <> struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void zero() { a = make_float4(0, 0, 0, 0); b = a; }
    __device__ __forceinline__ void get(float (&f)[8]) const {
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
};

// NCH = chunks of 8 per lane (N <= 512 * NCH).  Every wave walks its rows with the NEXT row's
// chunks already requested (raw, in registers): a row costs its arithmetic, not an HBM round trip.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int64_t M, int N, float eps,
                                                     T* __restrict__ y, int64_t ldy, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int nchunk = N >> 3;
    const float inv_n = 1.f / (float)N;
    const int64_t rstep = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    Raw8<T> nx[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        nx[c].zero();
        if (row < M && lane + 64 * c < nchunk) nx[c].load(x + row * N + (lane + 64 * c) * 8);
    }
    for (; row < M; row += rstep) {
        float v[NCH][8];
#pragma unroll
        for (int c = 0; c < NCH; ++c) nx[c].get(v[c]);
        const int64_t nrow = row + rstep;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (nrow < M && lane + 64 * c < nchunk) nx[c].load(x + nrow * N + (lane + 64 * c) * 8);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[c][j];   // chunks past the row hold zeros
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (lane + 64 * c < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float dlt = v[c][j] - mean; q += dlt * dlt; }
            }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunk) {
                float gm[8], bt[8], o[8];
                Raw8<float> rg, rb;
                rg.load(gamma + ch * 8); rb.load(beta + ch * 8);   // L1-resident
                rg.get(gm); rb.get(bt);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mean) * rstd * gm[j] + bt[j];
                Vec8<T>::st(y + row * ldy + ch * 8, o);
            }
        }
    }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                     const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const T* __restrict__ gres,
                                                     int64_t M, int N, T* __restrict__ gx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
    extern __shared__ float red[];  // [4 waves][2][N]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = N >> 3;
    const float inv_n = 1.f / (float)N;
    float dg[NCH][8], db[NCH][8], gm[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dg[c][j] = 0.f; db[c][j] = 0.f;
            const int ch = lane + 64 * c;
            gm[c][j] = ch < nchunk ? gamma[ch * 8 + j] : 0.f;
        }
    const int64_t rstep = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + wave;
    Raw8<T> ng[NCH], nxr[NCH], nr[NCH];
    float nmean = 0.f, nrstd = 0.f;
    auto fetch = [&](int64_t r) {
        if (r < M) {
            nmean = mean_in[r];
            nrstd = rstd_in[r];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = lane + 64 * c;
                if (ch < nchunk) {
                    ng[c].load(gy + r * N + ch * 8);
                    nxr[c].load(x + r * N + ch * 8);
                    if (gres) nr[c].load(gres + r * N + ch * 8);
                }
            }
        }
    };
#pragma unroll
    for (int c = 0; c < NCH; ++c) { ng[c].zero(); nxr[c].zero(); nr[c].zero(); }
    fetch(row);
    for (; row < M; row += rstep) {
        const float mean = nmean, rstd = nrstd;
        float g[NCH][8], xh[NCH][8], o[NCH][8];
#pragma unroll
        for (int c = 0; c < NCH; ++c) { ng[c].get(g[c]); nxr[c].get(xh[c]); nr[c].get(o[c]); }
        fetch(row + rstep);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (lane + 64 * c < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xh[c][j] = (xh[c][j] - mean) * rstd;
                    dg[c][j] += g[c][j] * xh[c][j];
                    db[c][j] += g[c][j];
                    g[c][j] *= gm[c][j];            // d/dxhat
                    s1 += g[c][j];
                    s2 += g[c][j] * xh[c][j];
                }
            }
        }
        const float c1 = wave_sum(s1) * inv_n, c2 = wave_sum(s2) * inv_n;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[c][j] += rstd * (g[c][j] - c1 - xh[c][j] * c2);   // o = gres or 0
                Vec8<T>::st(gx + row * N + ch * 8, o[c]);
            }
        }
    }
    // block reduction of the parameter gradients, then one atomic per column per block
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[(wave * 2 + 0) * N + ch * 8 + j] = dg[c][j];
                red[(wave * 2 + 1) * N + ch * 8 + j] = db[c][j];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) {
        atomicAdd(dgamma + i, red[0 * N + i] + red[2 * N + i] + red[4 * N + i] + red[6 * N + i]);
        atomicAdd(dbeta + i, red[1 * N + i] + red[3 * N + i] + red[5 * N + i] + red[7 * N + i]);
    }
}

static int ln_check(const char* who, int64_t M, int N, int dtype) {
    SP_CHECK_ARG(M >= 0 && N >= 8 && N % 8 == 0 && N <= 512 * kLnMaxChunks,
                 "%s: N=%d must be a multiple of 8 in [8, %d]", who, N, 512 * kLnMaxChunks);
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16 || dtype == SP_F16, "%s: dtype %d not supported (f32, bf16, f16)", who, dtype);
    return SP_OK;
}

}  // namespace sp

using namespace sp;

extern "C" int sp_layernorm_fwd(const void* x, const float* gamma, const float* beta, int64_t M, int N, float eps,
                                int dtype, void* y, int64_t ldy, float* mean, float* rstd, void* stream) {
    int rc = ln_check("sp_layernorm_fwd", M, N, dtype);
    if (rc) return rc;
    SP_CHECK_ARG(ldy >= N && ldy % 8 == 0, "sp_layernorm_fwd: ldy=%lld must be >= N and a multiple of 8", (long long)ldy);
    if (M == 0) return SP_OK;
    SP_CHECK_ARG(x && gamma && beta && y && mean && rstd, "sp_layernorm_fwd: null pointer");
    int64_t blocks = (M + 3) / 4;
    // a wave has one row in flight and one requested (2 x 2 KB at N = 1024): the grid, not the prefetch
    // depth, supplies the bytes in flight (measured at M = 24 k: 1024 blocks 20.9 us, 4096 blocks 18.6 us)
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    const int nch = (N + 511) / 512;
#define SP_LN_FWD(T, NCH) hipLaunchKernelGGL((ln_fwd_kernel<T, NCH>), dim3((int)blocks), dim3(256), 0, s, (const T*)x, \
                                             gamma, beta, M, N, eps, (T*)y, ldy, mean, rstd)
#define SP_LN_FWD_T(T) { if (nch == 1) SP_LN_FWD(T, 1); else if (nch == 2) SP_LN_FWD(T, 2); \
                         else if (nch == 3) SP_LN_FWD(T, 3); else SP_LN_FWD(T, 4); }
    if (dtype == SP_F32) SP_LN_FWD_T(float) else if (dtype == SP_F16) SP_LN_FWD_T(__half) else SP_LN_FWD_T(__hip_bfloat16)
#undef SP_LN_FWD_T
#undef SP_LN_FWD
    return check_launch("ln_fwd_kernel");
}

extern "C" int sp_layernorm_bwd(const void* gy, const void* x, const float* mean, const float* rstd,
                                const float* gamma, const void* gres, int64_t M, int N, int dtype, void* gx,
                                float* dgamma, float* dbeta, void* stream) {
    int rc = ln_check("sp_layernorm_bwd", M, N, dtype);
    if (rc) return rc;
    if (M == 0) return SP_OK;
    SP_CHECK_ARG(gy && x && mean && rstd && gamma && gx && dgamma && dbeta, "sp_layernorm_bwd: null pointer");
    int64_t blocks = (M + 3) / 4;
    // every block ends in 2 N float atomics onto the same 2 N words (~20 ns each, serialised per word):
    // one block per CU (measured at M = 24 k: 256 blocks 36 us, 512 blocks 42 us, 128 blocks 56 us)
    if (blocks > 256) blocks = 256;
    const size_t lds = (size_t)8 * N * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const int nch = (N + 511) / 512;
#define SP_LN_BWD(T, NCH) hipLaunchKernelGGL((ln_bwd_kernel<T, NCH>), dim3((int)blocks), dim3(256), lds, s, (const T*)gy, \
                                             (const T*)x, mean, rstd, gamma, (const T*)gres, M, N, (T*)gx, dgamma, dbeta)
#define SP_LN_BWD_T(T) { if (nch == 1) SP_LN_BWD(T, 1); else if (nch == 2) SP_LN_BWD(T, 2); \
                         else if (nch == 3) SP_LN_BWD(T, 3); else SP_LN_BWD(T, 4); }
    if (dtype == SP_F32) SP_LN_BWD_T(float) else if (dtype == SP_F16) SP_LN_BWD_T(__half) else SP_LN_BWD_T(__hip_bfloat16)
#undef SP_LN_BWD_T
#undef SP_LN_BWD
    return check_launch("ln_bwd_kernel");
}
