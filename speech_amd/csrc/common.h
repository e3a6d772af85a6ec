// Shared helpers for the gfx950 kernels of libspeech_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/speech_hip.h"

namespace sp {

constexpr int kWave = 64;  // CDNA wavefront

// thread-local error message, read through sp_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define SP_CHECK_ARG(cond, ...)                        \
    do {                                               \
        if (!(cond)) return sp::fail(SP_EINVAL, __VA_ARGS__); \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SP_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SP_OK;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- activation element types -------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float f(float v) { return v; }
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<__hip_bfloat16> {
    static __device__ __forceinline__ float f(__hip_bfloat16 v) { return __bfloat162float(v); }
    static __device__ __forceinline__ float ld(const __hip_bfloat16* p) {
        return __uint_as_float(((uint32_t)*reinterpret_cast<const uint16_t*>(p)) << 16);
    }
    static __device__ __forceinline__ void st(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
};
template <> struct Elem<__half> {
    static __device__ __forceinline__ float f(__half v) { return __half2float(v); }
    static __device__ __forceinline__ float ld(const __half* p) { return __half2float(*p); }
    static __device__ __forceinline__ void st(__half* p, float v) { *p = __float2half(v); }
};

// ---- wave-level primitives (64 lanes) -----------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

}  // namespace sp
