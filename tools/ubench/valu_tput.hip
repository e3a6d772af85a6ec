// VALU throughput on gfx950: W waves per SIMD, each issuing a long stream of INDEPENDENT instructions of one
// kind (8 accumulators).  Prints shader cycles per wave-instruction as seen by ONE SIMD (s_memtime of wave 0
// divided by the instructions all of that SIMD's waves issued), for plain and packed f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr int ITERS = 2048;

template <int MODE>
__global__ void stream(float* out, long long* cyc) {
    float a[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = out[threadIdx.x + i]; p[i].x = a[i]; p[i].y = a[i] + 1.f; }
    const float b = 1.0001f, c = 1e-4f;
    const f32x2 pb = {1.0001f, 0.9999f}, pc = {1e-4f, 2e-4f};
    const unsigned long long smask = __builtin_amdgcn_read_exec() >> 1;
    const float sc = __builtin_amdgcn_readfirstlane(out[0]) + 1e-4f;
    unsigned scnt[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    __shared__ float lds[16384];
    lds[threadIdx.x] = 0.f;
    const unsigned ldsaddr = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 2048;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4 q4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    __syncthreads();
    long long c0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            if (MODE == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));
            if (MODE == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
            if (MODE == 8) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(p[i]) : "v"(pc));
            if (MODE == 9) asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (MODE == 10) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(smask));
            if (MODE == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "s"(sc));
            if (MODE == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
            if (MODE == 13) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
            if (MODE == 14) asm volatile("s_add_u32 %0, %0, 1" : "+s"(scnt[i]));
            if (MODE == 15) asm volatile("ds_read_b64 %0, %1" : "=v"(p[i]) : "v"(ldsaddr + 8 * i));
            if (MODE == 16) asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(q4[i & 1]) : "v"(ldsaddr + 16 * i));
            if (MODE == 17) asm volatile("ds_write_b64 %0, %1" : : "v"(ldsaddr + 8 * i), "v"(p[i]));
            if (MODE == 18) asm volatile("ds_write_b32 %0, %1" : : "v"(ldsaddr / 2 + 4 * i), "v"(a[i]));
        }
        if (MODE >= 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + scnt[i] + q4[i & 1].x;
    out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}

int main() {
    float* out; long long* cyc;
    CK(hipMalloc(&out, 1 << 16)); CK(hipMalloc(&cyc, 8)); CK(hipMemset(out, 0, 1 << 16));
    const char* names[] = {"v_fma_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_mov_b32", "v_sqrt_f32",
                           "v_cndmask_b32", "v_lshl_add_u64", "v_pk_mov_b32", "v_cndmask e64 sgpr", "v_add_f32 sgpr", "v_cmp vcc", "v_cmp+v_cndmask",
                           "s_add_u32", "ds_read_b64", "ds_read2_b64", "ds_write_b64", "ds_write_b32"};
    for (int wps : {1, 3, 4}) {          // waves per SIMD: one workgroup of 4*wps waves on one CU
        for (int mode = 0; mode < 19; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                dim3 g(1), t(256 * wps);
                switch (mode) {
                    case 0: stream<0><<<g, t>>>(out, cyc); break; case 1: stream<1><<<g, t>>>(out, cyc); break;
                    case 2: stream<2><<<g, t>>>(out, cyc); break; case 3: stream<3><<<g, t>>>(out, cyc); break;
                    case 4: stream<4><<<g, t>>>(out, cyc); break; case 5: stream<5><<<g, t>>>(out, cyc); break;
                    case 6: stream<6><<<g, t>>>(out, cyc); break; case 7: stream<7><<<g, t>>>(out, cyc); break;
                    case 8: stream<8><<<g, t>>>(out, cyc); break; case 9: stream<9><<<g, t>>>(out, cyc); break;
                    case 10: stream<10><<<g, t>>>(out, cyc); break; case 11: stream<11><<<g, t>>>(out, cyc); break;
                    case 12: stream<12><<<g, t>>>(out, cyc); break; case 13: stream<13><<<g, t>>>(out, cyc); break;
                    case 14: stream<14><<<g, t>>>(out, cyc); break; case 15: stream<15><<<g, t>>>(out, cyc); break;
                    case 16: stream<16><<<g, t>>>(out, cyc); break; case 17: stream<17><<<g, t>>>(out, cyc); break;
                    case 18: stream<18><<<g, t>>>(out, cyc); break;
                }
                CK(hipDeviceSynchronize());
            }
            long long hc; CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
            printf("waves/SIMD %d  %-16s %6.2f cycles per wave-instruction on the SIMD (%.2f per instruction of one wave)\n", wps,
                   names[mode], (double)hc / (ITERS * 8.0 * wps), (double)hc / (ITERS * 8.0));
        }
    }
    return 0;
}
