// Latency micro-benchmarks for gfx950: one wave (or a few) per CU, long dependent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int N = 4096;
template <int MODE>
__global__ void chain(float* out, long long* cyc, long long* wall, int nblk_active) {
    __shared__ float lds[256];
    lds[threadIdx.x] = threadIdx.x * 0.001f;
    __syncthreads();
    float a = out[threadIdx.x], b = 1.0001f, c = 0.0001f;
    double da = a, db = 1.0001, dc = 1e-4;
    int idx = threadIdx.x;
    long long w0 = wall_clock64();
    long long c0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        if (MODE == 0) a = fmaf(a, b, c);
        if (MODE == 1) da = fma(da, db, dc);
        if (MODE == 2) a = __builtin_amdgcn_exp2f(a);
        if (MODE == 3) a = __builtin_amdgcn_rcpf(a);
        if (MODE == 4) { idx = (int)lds[idx & 255]; }
        if (MODE == 5) a = __builtin_amdgcn_update_dpp(0.f, a, 0x138, 0xF, 0xF, true) + c;
        if (MODE == 6) { __builtin_amdgcn_s_barrier(); }
        if (MODE == 7) { a = fmaf(a, b, c); b = fmaf(b, a, c); }   // two dependent
        if (MODE == 8) da = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(da), 0x138, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, __double2loint(da), 0x138, 0xF, 0xF, true)) * db;
    }
    long long c1 = clock64();
    long long w1 = wall_clock64();
    out[threadIdx.x] = a + (float)da + idx + b;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; wall[0] = w1 - w0; }
}
int main() {
    float* out; long long *cyc, *wall;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&wall, 8));
    CK(hipMemset(out, 0, 4096));
    const char* names[] = {"f32 fma", "f64 fma", "exp2", "rcp", "ds_read dependent", "dpp+add", "s_barrier(4 waves)", "2 dep fma", "f64 dpp+mul"};
    for (int grid : {1, 512}) {
        for (int mode = 0; mode < 9; ++mode) {
            int threads = mode == 6 ? 256 : 64;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: chain<0><<<grid, threads>>>(out, cyc, wall, grid); break;
                    case 1: chain<1><<<grid, threads>>>(out, cyc, wall, grid); break;
                    case 2: chain<2><<<grid, threads>>>(out, cyc, wall, grid); break;
                    case 3: chain<3><<<grid, threads>>>(out, cyc, wall, grid); break;
                    case 4: chain<4><<<grid, threads>>>(out, cyc, wall, grid); break;
                    case 5: chain<5><<<grid, threads>>>(out, cyc, wall, grid); break;
                    case 6: chain<6><<<grid, threads>>>(out, cyc, wall, grid); break;
                    case 7: chain<7><<<grid, threads>>>(out, cyc, wall, grid); break;
                    case 8: chain<8><<<grid, threads>>>(out, cyc, wall, grid); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long hc, hw; CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hw, wall, 8, hipMemcpyDeviceToHost));
            printf("grid %4d %-22s: %7.1f clock64 ticks/iter, %7.2f ns/iter (wall_clock 100MHz), kernel %.1f us\n", grid, names[mode],
                   (double)hc / N, (double)hw * 10.0 / N, ms * 1e3);
        }
    }
    return 0;
}
