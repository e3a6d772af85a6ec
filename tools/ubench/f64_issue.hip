// What one serial wave pays per instruction on gfx950 for the instruction kinds of the CTC lattice recursion
// (csrc/ctc.hip, ctc_lattice1_kernel): f64 add / mul / fma, the two-mov DPP shift of an f64, 16-byte LDS reads and
// 16-byte global stores, against v_fma_f32 in the same run.  One workgroup of ONE wave; every stream is 8 independent
// accumulators (issue cost), and the chain modes are one dependent accumulator (latency).
// Build: hipcc --offload-arch=gfx950 -O3 -o f64_issue f64_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>
__global__ __launch_bounds__(64) void stream(double* out, long long* cyc, int nwaves_unused) {
    double d[8];
    float a[8];
    f32x4 q[8];
    for (int i = 0; i < 8; ++i) { d[i] = out[threadIdx.x + i] + 1.0; a[i] = (float)d[i]; q[i] = f32x4{a[i], a[i], a[i], a[i]}; }
    const double b = 1.0000001, c = 1e-9;
    unsigned sc0 = 0, sc1 = 0;
    __shared__ __attribute__((aligned(16))) double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 1.0;
    const unsigned ldsaddr = threadIdx.x * 16;
    double* gp = out + 4096 + threadIdx.x * 2;   // 1 KB per store instruction, 8 distinct rows
    __syncthreads();
    long long c0 = wall_clock64();
    long long s0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"((float)b), "v"((float)c));
            if (MODE == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(b), "v"(c));
            if (MODE == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(b));
            if (MODE == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
            if (MODE == 4) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
            if (MODE == 5) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(0));
            if (MODE == 6) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            if (MODE == 7) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(ldsaddr + 1024 * i));
            if (MODE == 8) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(gp + 128 * i), "v"(q[i]) : "memory");
            if (MODE == 9) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[0]) : "v"(b), "v"(c));   // dependent chain
            if (MODE == 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[0]) : "v"(b));              // dependent chain
            if (MODE == 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"((float)b), "v"((float)c));   // dependent chain
            if (MODE == 12) {   // the recursion's chain: mul -> two-mov shift -> fma -> (next) mul
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[0]) : "v"(b));
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(d[0]), 0x138, 0xf, 0xf, true);
                const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(d[0]), 0x138, 0xf, 0xf, true);
                const double t = __hiloint2double(hi, lo);
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[0]) : "v"(t), "v"(c));
            }
            if (MODE == 14) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(gp + 128 * i), "v"(q[i]) : "memory");   // never waited on
            if (MODE == 15) {   // one store per five independent f64 fmas (the recursion's mix at two states per lane)
                asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(gp + 128 * i), "v"(q[i]) : "memory");
                asm volatile("v_fma_f64 %0, %0, %5, %6\n v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n v_fma_f64 %3, %3, %5, %6\n v_fma_f64 %4, %4, %5, %6"
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]) : "v"(b), "v"(c));
            }
            if (MODE == 16) {   // same with a 16-byte LDS read too
                asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(gp + 128 * i), "v"(q[i & 3]) : "memory");
                asm volatile("ds_read_b128 %0, %1" : "=v"(q[4 + (i & 3)]) : "v"(ldsaddr + 1024 * i));
                asm volatile("v_fma_f64 %0, %0, %5, %6\n v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n v_fma_f64 %3, %3, %5, %6\n v_fma_f64 %4, %4, %5, %6"
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]) : "v"(b), "v"(c));
            }
            if (MODE == 17) {   // mode 15 without the store
                asm volatile("v_fma_f64 %0, %0, %5, %6\n v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n v_fma_f64 %3, %3, %5, %6\n v_fma_f64 %4, %4, %5, %6"
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]) : "v"(b), "v"(c));
            }
            if (MODE == 13) asm volatile("s_add_u32 %0, %0, 1\n s_addc_u32 %1, %1, 0" : "+s"(sc0), "+s"(sc1));
        }
        if (MODE == 7 || MODE == 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    long long s1 = clock64();
    long long c1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i] + a[i] + q[i].x + sc0 + sc1;
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[0] = s1 - s0; cyc[1] = c1 - c0; }
}

int main() {
    double* out; long long* cyc;
    CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&cyc, 16)); CK(hipMemset(out, 0, 1 << 20));
    const char* names[] = {"v_fma_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_ldexp_f64", "v_mov_b32_dpp",
                           "ds_read_b128", "global_store_dwordx4", "chain v_fma_f64", "chain v_mul_f64", "chain v_fma_f32",
                           "chain mul+2dpp+fma (x4 instr)", "s_add_u32+s_addc_u32 (x2 instr)", "global_store_dwordx4, no wait",
                           "store + 5 fma_f64", "store + ds_read_b128 + 5 fma_f64", "5 fma_f64"};
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    for (int mode = 0; mode < 18; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: stream<0><<<1, 64>>>(out, cyc, 0); break; case 1: stream<1><<<1, 64>>>(out, cyc, 0); break;
                case 2: stream<2><<<1, 64>>>(out, cyc, 0); break; case 3: stream<3><<<1, 64>>>(out, cyc, 0); break;
                case 4: stream<4><<<1, 64>>>(out, cyc, 0); break; case 5: stream<5><<<1, 64>>>(out, cyc, 0); break;
                case 6: stream<6><<<1, 64>>>(out, cyc, 0); break; case 7: stream<7><<<1, 64>>>(out, cyc, 0); break;
                case 8: stream<8><<<1, 64>>>(out, cyc, 0); break; case 9: stream<9><<<1, 64>>>(out, cyc, 0); break;
                case 10: stream<10><<<1, 64>>>(out, cyc, 0); break; case 11: stream<11><<<1, 64>>>(out, cyc, 0); break;
                case 12: stream<12><<<1, 64>>>(out, cyc, 0); break; case 13: stream<13><<<1, 64>>>(out, cyc, 0); break;
                case 14: stream<14><<<1, 64>>>(out, cyc, 0); break; case 15: stream<15><<<1, 64>>>(out, cyc, 0); break;
                case 16: stream<16><<<1, 64>>>(out, cyc, 0); break; case 17: stream<17><<<1, 64>>>(out, cyc, 0); break;
            }
            CK(hipDeviceSynchronize());
        }
        long long hc[2]; CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
        printf("%-32s %7.2f clock64 ticks, %7.2f ns per group of the stream\n", names[mode], (double)hc[0] / (ITERS * 8.0),
               (double)hc[1] / (ITERS * 8.0) * 1e6 / wall_khz);
    }
    return 0;
}
