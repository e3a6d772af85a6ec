// General MFMA GEMM for every shape the fast kernels (gemm.hip) do not take: any M, N, K, either
// operand K-contiguous or row-contiguous (so NT, NN, TN and TT problems), bf16 / f16 / f32 inputs with
// f32 accumulation, f32 / bf16 / f16 output, optional accumulation into C and split-K into slabs.
// These are the remaining matmuls of net.py:61-76 (sru.SRU's projection and its gradients at small or
// odd shapes and in fp32; the classifier's nn.Linear and its gradients).
//
//   C[m][n] (+)= sum_k A(m,k) B(n,k),   A(m,k) = A[m*sam + k*sak],  B(n,k) = B[n*sbn + k*sbk]
//
// Workgroup = 4 waves on a 128 x 128 tile, K step 32, each wave 64 x 64 = 2 x 2 blocks of
// v_mfma_f32_32x32x16_{bf16,f16} (or v_mfma_f32_32x32x2_f32: exact f32 at the vector rate; gfx950 has
// no reduced-precision f32 path).  Operands go global -> registers (16-byte vectors along the
// contiguous dimension, element-wise and zero-filled at the edges) -> LDS as [row][k] with a padded
// pitch; a row-contiguous operand is transposed by the LDS stores.  The next K step's global loads are
// in flight while the current one is multiplied.  The MFMA takes the B rows as its row operand, so a
// lane ends up with 4 consecutive n of one m and stores 8 or 16 contiguous bytes.
#include "gemm.h"

namespace sp {

typedef __bf16 gg_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gg_f16x8 __attribute__((ext_vector_type(8)));
typedef float gg_f32x16 __attribute__((ext_vector_type(16)));

constexpr int GT = 128;    // tile edge
constexpr int GK = 32;     // K step

template <int DT> struct GIn;
template <> struct GIn<SP_BF16> { typedef uint16_t E; static constexpr int VEC = 8, PITCH = 40; };
template <> struct GIn<SP_F16> { typedef uint16_t E; static constexpr int VEC = 8, PITCH = 40; };
template <> struct GIn<SP_F32> { typedef float E; static constexpr int VEC = 4, PITCH = 33; };


// one operand's 128 x 32 tile slice for this thread: VEC elements along the contiguous dimension
template <int DT, bool KC>
struct GLoader {
    typedef typename GIn<DT>::E E;
    static constexpr int VEC = GIn<DT>::VEC, PITCH = GIn<DT>::PITCH;
    static constexpr int NV = GT * GK / VEC / 256;     // vectors per thread
    E v[NV][VEC];

    // X(r, k) = X[r*sr + k*sk]; rows r0.. (R in all), k from k0 (up to kend)
    __device__ __forceinline__ void load(const E* X, int64_t sr, int64_t sk, int r0, int R, int k0, int kend, int tid,
                                         bool vec) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int id = tid + 256 * i;
            int row, kk, nvalid;
            const E* p;
            if (KC) {                     // vector along k
                row = id / (GK / VEC); kk = (id % (GK / VEC)) * VEC;
                nvalid = (r0 + row < R) ? min(VEC, kend - (k0 + kk)) : 0;
                p = X + (int64_t)(r0 + row) * sr + (k0 + kk);
            } else {                      // vector along the rows
                kk = id / (GT / VEC); row = (id % (GT / VEC)) * VEC;
                nvalid = (k0 + kk < kend) ? min(VEC, R - (r0 + row)) : 0;
                p = X + (int64_t)(k0 + kk) * sk + (r0 + row);
            }
            if (vec && nvalid >= VEC) {
                const uint4 q = *reinterpret_cast<const uint4*>(p);
                *reinterpret_cast<uint4*>(v[i]) = q;
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[i][e] = (e < nvalid) ? p[e] : (E)0;
            }
        }
    }
    __device__ __forceinline__ void store(E* S, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int id = tid + 256 * i;
            if (KC) {
                const int row = id / (GK / VEC), kk = (id % (GK / VEC)) * VEC;
                if (DT != SP_F32) {
                    *reinterpret_cast<uint4*>(S + row * PITCH + kk) = *reinterpret_cast<const uint4*>(v[i]);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) S[row * PITCH + kk + e] = v[i][e];
                }
            } else {
                const int kk = id / (GT / VEC), row = (id % (GT / VEC)) * VEC;
#pragma unroll
                for (int e = 0; e < VEC; ++e) S[(row + e) * PITCH + kk] = v[i][e];
            }
        }
    }
};

template <int DT, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs g) {
    typedef typename GIn<DT>::E E;
    constexpr int PITCH = GIn<DT>::PITCH;
    __shared__ __attribute__((aligned(16))) E As[GT * PITCH];
    __shared__ __attribute__((aligned(16))) E Bs[GT * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    // this split's K range, in whole K steps
    const int nsteps = (g.K + GK - 1) / GK;
    const int per = (nsteps + g.split_k - 1) / g.split_k;
    const int kbeg = min((int)blockIdx.z * per, nsteps) * GK, kend = min(min((int)blockIdx.z * per + per, nsteps) * GK, g.K);
    const E* A = reinterpret_cast<const E*>(g.A);
    const E* B = reinterpret_cast<const E*>(g.B);

    gg_f32x16 acc[2][2];     // [n block][m block]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    GLoader<DT, AKC> la;
    GLoader<DT, BKC> lb;
    const int r = lane & 31, h = lane >> 5;
    if (kbeg < kend) {
        la.load(A, g.sam, g.sak, m0, g.M, kbeg, kend, tid, g.vec_in != 0);
        lb.load(B, g.sbn, g.sbk, n0, g.N, kbeg, kend, tid, g.vec_in != 0);
    }
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        la.store(As, tid);
        lb.store(Bs, tid);
        __syncthreads();
        if (k0 + GK < kend) {
            la.load(A, g.sam, g.sak, m0, g.M, k0 + GK, kend, tid, g.vec_in != 0);
            lb.load(B, g.sbn, g.sbk, n0, g.N, k0 + GK, kend, tid, g.vec_in != 0);
        }
        const E* sa = As + (wm * 64 + r) * PITCH;
        const E* sb = Bs + (wn * 64 + r) * PITCH;
        if constexpr (DT == SP_F32) {
#pragma unroll
            for (int kk = 0; kk < GK / 2; ++kk) {
                float fa[2], fb[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) fa[b] = sa[b * 32 * PITCH + kk * 2 + h];
#pragma unroll
                for (int a = 0; a < 2; ++a) fb[a] = sb[a * 32 * PITCH + kk * 2 + h];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[a], fa[b], acc[a][b], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < GK / 16; ++kk) {
                uint4 fa[2], fb[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) fa[b] = *reinterpret_cast<const uint4*>(sa + b * 32 * PITCH + kk * 16 + h * 8);
#pragma unroll
                for (int a = 0; a < 2; ++a) fb[a] = *reinterpret_cast<const uint4*>(sb + a * 32 * PITCH + kk * 16 + h * 8);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        if constexpr (DT == SP_BF16)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gg_bf16x8, fb[a]),
                                                                                __builtin_bit_cast(gg_bf16x8, fa[b]), acc[a][b], 0, 0, 0);
                        else
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gg_f16x8, fb[a]),
                                                                               __builtin_bit_cast(gg_f16x8, fa[b]), acc[a][b], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }

    // D[row = n][col = m]: lane = m (r), register q -> n = 8 (q >> 2) + 4 h + (q & 3)
    char* Cb = reinterpret_cast<char*>(g.C);
    const int esz = g.out_dtype == SP_F32 ? 4 : 2;
    Cb += (int64_t)blockIdx.z * g.c_split_stride * esz;
    const bool vec_ok = (g.ldc % 4) == 0 && (((uintptr_t)g.C + (uint64_t)blockIdx.z * g.c_split_stride * esz) & 15) == 0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = m0 + wm * 64 + b * 32 + r;
        if (m >= g.M) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int n = n0 + wn * 64 + a * 32 + 8 * q4 + 4 * h;
                if (n >= g.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * q4 + e];
                const int64_t off = (int64_t)m * g.ldc + n;
                const bool full = vec_ok && n + 3 < g.N;
                if (g.out_dtype == SP_F32) {
                    float* c = reinterpret_cast<float*>(Cb) + off;
                    if (full) {
                        float4 o = make_float4(v[0], v[1], v[2], v[3]);
                        if (g.accumulate) {
                            const float4 p = *reinterpret_cast<const float4*>(c);
                            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
                        }
                        *reinterpret_cast<float4*>(c) = o;
                    } else {
                        for (int e = 0; e < 4 && n + e < g.N; ++e) c[e] = g.accumulate ? c[e] + v[e] : v[e];
                    }
                } else if (g.out_dtype == SP_BF16) {
                    __hip_bfloat16* c = reinterpret_cast<__hip_bfloat16*>(Cb) + off;
                    if (full) {
                        uint2 o;
                        o.x = (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[0])) |
                              ((uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[1])) << 16);
                        o.y = (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[2])) |
                              ((uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[3])) << 16);
                        *reinterpret_cast<uint2*>(c) = o;
                    } else {
                        for (int e = 0; e < 4 && n + e < g.N; ++e) c[e] = __float2bfloat16(v[e]);
                    }
                } else {
                    __half* c = reinterpret_cast<__half*>(Cb) + off;
                    for (int e = 0; e < 4 && n + e < g.N; ++e) c[e] = __float2half(v[e]);
                }
            }
    }
}

template <int DT>
static void launch_generic(const GemmArgs& g, bool akc, bool bkc, hipStream_t s) {
    dim3 grid((g.N + GT - 1) / GT, (g.M + GT - 1) / GT, g.split_k);
    if (akc && bkc) hipLaunchKernelGGL((gemm_generic_kernel<DT, true, true>), grid, dim3(256), 0, s, g);
    else if (akc) hipLaunchKernelGGL((gemm_generic_kernel<DT, true, false>), grid, dim3(256), 0, s, g);
    else if (bkc) hipLaunchKernelGGL((gemm_generic_kernel<DT, false, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_generic_kernel<DT, false, false>), grid, dim3(256), 0, s, g);
}

// host side of the general kernel; gemm.hip's sp_gemm calls it for whatever its fast kernels do not take
int gemm_generic(const GemmArgs& g, int in_dtype, hipStream_t s) {
    const bool akc = g.sak == 1, bkc = g.sbk == 1;
    if (in_dtype == SP_BF16) launch_generic<SP_BF16>(g, akc, bkc, s);
    else if (in_dtype == SP_F16) launch_generic<SP_F16>(g, akc, bkc, s);
    else launch_generic<SP_F32>(g, akc, bkc, s);
    return check_launch("gemm_generic_kernel");
}

// out[i] (+)= scale * sum_s P[s][i]   (split-K slabs -> the caller's f32 buffer; scale read on the device)
__global__ void sum_slabs_kernel(const float* __restrict__ P, int S, int64_t n4, int64_t slab, float* __restrict__ out,
                                 int accumulate, const float* __restrict__ scale) {
    const float sc = scale ? *scale : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(P)[i];
        for (int s = 1; s < S; ++s) {
            const float4 b = reinterpret_cast<const float4*>(P + (int64_t)s * slab)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc;
        float4* o = reinterpret_cast<float4*>(out) + i;
        if (accumulate) {
            const float4 p = *o;
            a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
        }
        *o = a;
    }
}

// Linear-layer weight [V][n] f32 -> the 16-bit operands of its three products in ONE launch: out_w [Vp][n]
// (rows V..Vp zero: the logits GEMM works on whole 256-column tiles) and / or out_t [n][Vp] = scale * W^T
// (the K-contiguous B operand of the data gradient, with the incoming loss gradient folded in).  64 x 64 tiles
// through LDS: reads and both writes are row-contiguous.
template <typename T>
__global__ __launch_bounds__(256) void linear_pack_kernel(const float* __restrict__ w, int V, int n, int Vp,
                                                         const float* __restrict__ scale, T* __restrict__ out_w,
                                                         T* __restrict__ out_t) {
    __shared__ float tile[64][65];
    const int v0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float sc = scale ? *scale : 1.f;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {      // all sixteen loads of the thread in flight at once
        const int v = v0 + ty + 4 * i, c = c0 + tx;
        x[i] = (v < V && c < n) ? w[(int64_t)v * n + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ty + 4 * i, v = v0 + r, c = c0 + tx;
        tile[r][tx] = x[i];
        if (out_w && v < Vp && c < n) Elem<T>::st(out_w + (int64_t)v * n + c, x[i]);
    }
    __syncthreads();
    if (out_t) {
#pragma unroll
        for (int r = ty; r < 64; r += 4) {
            const int c = c0 + r, v = v0 + tx;
            if (c < n && v < Vp) Elem<T>::st(out_t + (int64_t)c * Vp + v, tile[tx][r] * sc);
        }
    }
}

}  // namespace sp

extern "C" int sp_sum_slabs_f32(const float* partials, int S, int64_t n, int64_t slab_stride, float* out,
                                int accumulate, const float* scale_dev, void* stream) {
    using namespace sp;
    SP_CHECK_ARG(S >= 1 && n >= 0 && slab_stride >= n, "sp_sum_slabs_f32: bad shape S=%d n=%lld", S, (long long)n);
    SP_CHECK_ARG(n % 4 == 0 && slab_stride % 4 == 0, "sp_sum_slabs_f32: n and slab_stride must be multiples of 4");
    if (n == 0) return SP_OK;
    SP_CHECK_ARG(partials && out, "sp_sum_slabs_f32: null pointer");
    SP_CHECK_ARG((((uintptr_t)partials | (uintptr_t)out) & 15) == 0, "sp_sum_slabs_f32: buffers must be 16-byte aligned");
    const int64_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partials, S, n4,
                       slab_stride, out, accumulate, scale_dev);
    return check_launch("sum_slabs_kernel");
}

extern "C" int sp_linear_pack(const float* weight, int V, int n, int Vp, const float* scale_dev, int dtype, void* out_w,
                              void* out_t, void* stream) {
    using namespace sp;
    SP_CHECK_ARG(V >= 1 && n >= 1 && Vp >= V, "sp_linear_pack: bad shape V=%d n=%d Vp=%d", V, n, Vp);
    SP_CHECK_ARG(dtype == SP_BF16 || dtype == SP_F16, "sp_linear_pack: dtype %d (bf16 or f16)", dtype);
    SP_CHECK_ARG(weight && (out_w || out_t), "sp_linear_pack: null pointer");
    const dim3 grid((n + 63) / 64, (Vp + 63) / 64);
    SP_CHECK_ARG(grid.y <= 65535, "sp_linear_pack: Vp=%d exceeds the grid limit", Vp);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SP_BF16)
        hipLaunchKernelGGL((linear_pack_kernel<__hip_bfloat16>), grid, dim3(256), 0, s, weight, V, n, Vp, scale_dev,
                           (__hip_bfloat16*)out_w, (__hip_bfloat16*)out_t);
    else
        hipLaunchKernelGGL((linear_pack_kernel<__half>), grid, dim3(256), 0, s, weight, V, n, Vp, scale_dev, (__half*)out_w,
                           (__half*)out_t);
    return check_launch("linear_pack_kernel");
}
