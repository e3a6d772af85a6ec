// Bidirectional SRU recurrence (forward and backward) for gfx950.
// Replaces the `sru` package's fused recurrence kernel that net.py:61-72 of the reference runs
// through sru.SRU(..., bidirectional=True, layer_norm=True, rescale=False, highway_bias=0).
//
// The recurrence is serial in time and independent per column (batch, direction, channel):
//   f = sigmoid(u1 + v_f c' + b_f)   r = sigmoid(u2 + v_r c' + b_r)     (c' = previous cell state)
//   c = u0 + (c' - u0) f             h = (c m - x') r + x'
// so parallelism is B*2*d columns only (32 k at B = 32): at most one or two waves per CU.  The
// kernels therefore spend registers, not occupancy, on latency hiding: one lane = one column,
// one wave = 64 adjacent channels (coalesced 128/256-byte row segments), and each wave keeps two
// register batches of UN time steps in flight (load batch n+1 while the serial chain runs over
// batch n).  U is laid out [T*B][k][2][d] so that each gate plane is contiguous across lanes.
#include <type_traits>

#include "common.h"

namespace sp {

constexpr float kL2e = 1.4426950408889634f;

// sigmoid on the raw transcendental units (v_exp_f32 / v_rcp_f32, 1 ulp each): the serial chain of
// the recurrence is c -> fma -> exp -> add -> rcp -> fma, so every instruction on it is latency.
__device__ __forceinline__ float sigmoid_neg2(float xs) {  // xs = -log2(e) * x
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(xs));
}
__device__ __forceinline__ float sigmoidf_fast(float x) { return sigmoid_neg2(-kL2e * x); }

template <typename T, int K, int UN>
__global__ __launch_bounds__(64) void sru_fwd_kernel(
    const T* __restrict__ U, const T* __restrict__ xres, const float* __restrict__ wc,
    const float* __restrict__ bias, const float* __restrict__ mask, int Tn, int B, int d,
    T* __restrict__ h, float* __restrict__ cst) {
    const int D2 = 2 * d;
    const int col = blockIdx.x * 64 + threadIdx.x;  // over [B][2][d]
    const int b = col / D2;
    const int cd = col - b * D2;                      // dir*d + j
    const int dir = cd / d;
    const float vf = wc[cd], vr = wc[D2 + cd];
    const float bf = bias[cd], br = bias[D2 + cd];
    const float m = mask ? mask[col] : 1.f;
    const int64_t ustep = (int64_t)B * K * D2;        // elements between consecutive t in U
    const int64_t hstep = (int64_t)B * D2;
    const T* Ub = U + (int64_t)b * K * D2 + cd;
    const T* Xb = (K == 3) ? xres + (int64_t)b * D2 + cd : nullptr;
    T* hb = h + (int64_t)b * D2 + cd;
    float* cb = cst + (int64_t)b * D2 + cd;

    T a0[UN], a1[UN], a2[UN], a3[UN];
    T b0[UN], b1[UN], b2[UN], b3[UN];

    auto tmap = [&](int i) { return dir ? Tn - 1 - i : i; };
    auto load = [&](T(&u0)[UN], T(&u1)[UN], T(&u2)[UN], T(&xp)[UN], int i0) {
#pragma unroll
        for (int s = 0; s < UN; ++s) {
            const int i = i0 + s;
            if (i < Tn) {
                const int64_t t = tmap(i);
                const T* p = Ub + t * ustep;
                u0[s] = p[0];
                u1[s] = p[D2];
                u2[s] = p[2 * D2];
                xp[s] = (K == 3) ? Xb[t * hstep] : p[3 * D2];
            }
        }
    };
    float c = 0.f;
    auto compute = [&](T(&u0)[UN], T(&u1)[UN], T(&u2)[UN], T(&xp)[UN], int i0) {
#pragma unroll
        for (int s = 0; s < UN; ++s) {
            const int i = i0 + s;
            if (i < Tn) {
                const int64_t t = tmap(i);
                const float v0 = Elem<T>::f(u0[s]), v1 = Elem<T>::f(u1[s]);
                const float v2 = Elem<T>::f(u2[s]), x = Elem<T>::f(xp[s]);
                const float f = sigmoidf_fast(v1 + vf * c + bf);
                const float r = sigmoidf_fast(v2 + vr * c + br);
                c = v0 + (c - v0) * f;
                const float hv = (c * m - x) * r + x;
                Elem<T>::st(hb + t * hstep, hv);
                cb[t * hstep] = c;
            }
        }
    };

    load(a0, a1, a2, a3, 0);
    for (int i0 = 0; i0 < Tn; i0 += 2 * UN) {
        load(b0, b1, b2, b3, i0 + UN);
        compute(a0, a1, a2, a3, i0);
        load(a0, a1, a2, a3, i0 + 2 * UN);
        compute(b0, b1, b2, b3, i0 + UN);
    }
}

template <typename T, int K, int UN>
__global__ __launch_bounds__(64) void sru_bwd_kernel(
    const T* __restrict__ U, const T* __restrict__ xres, const float* __restrict__ wc,
    const float* __restrict__ bias, const float* __restrict__ mask, const float* __restrict__ cst,
    const T* __restrict__ gh, int Tn, int B, int d, T* __restrict__ gU, T* __restrict__ gxres,
    float* __restrict__ gwc, float* __restrict__ gbias) {
    const int D2 = 2 * d;
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int b = col / D2;
    const int cd = col - b * D2;
    const int dir = cd / d;
    const float vf = wc[cd], vr = wc[D2 + cd];
    const float bf = bias[cd], br = bias[D2 + cd];
    const float m = mask ? mask[col] : 1.f;
    const int64_t ustep = (int64_t)B * K * D2;
    const int64_t hstep = (int64_t)B * D2;
    const T* Ub = U + (int64_t)b * K * D2 + cd;
    T* gUb = gU + (int64_t)b * K * D2 + cd;
    const T* Xb = (K == 3) ? xres + (int64_t)b * D2 + cd : nullptr;
    T* gXb = (K == 3) ? gxres + (int64_t)b * D2 + cd : nullptr;
    const T* ghb = gh + (int64_t)b * D2 + cd;
    const float* cb = cst + (int64_t)b * D2 + cd;

    struct Batch {
        T u0[UN], u1[UN], u2[UN], xp[UN], g[UN];
        float cp[UN];
    };
    Batch A, Bt;

    // walk the forward order backwards: i = Tn-1 .. 0, forward-time index t = tmap(i)
    auto tmap = [&](int i) { return dir ? Tn - 1 - i : i; };
    auto load = [&](Batch& q, int i_hi) {  // steps i_hi, i_hi-1, ..., i_hi-UN+1
#pragma unroll
        for (int s = 0; s < UN; ++s) {
            const int i = i_hi - s;
            if (i >= 0) {
                const int64_t t = tmap(i);
                const T* p = Ub + t * ustep;
                q.u0[s] = p[0];
                q.u1[s] = p[D2];
                q.u2[s] = p[2 * D2];
                q.xp[s] = (K == 3) ? Xb[t * hstep] : p[3 * D2];
                q.g[s] = ghb[t * hstep];
                q.cp[s] = (i > 0) ? cb[(int64_t)tmap(i - 1) * hstep] : 0.f;
            }
        }
    };
    float gc = 0.f, gvf = 0.f, gvr = 0.f, gbf = 0.f, gbr = 0.f;
    auto compute = [&](Batch& q, int i_hi) {
#pragma unroll
        for (int s = 0; s < UN; ++s) {
            const int i = i_hi - s;
            if (i >= 0) {
                const int64_t t = tmap(i);
                const float v0 = Elem<T>::f(q.u0[s]), v1 = Elem<T>::f(q.u1[s]);
                const float v2 = Elem<T>::f(q.u2[s]), x = Elem<T>::f(q.xp[s]);
                const float g = Elem<T>::f(q.g[s]);
                const float cp = q.cp[s];
                const float f = sigmoidf_fast(v1 + vf * cp + bf);
                const float r = sigmoidf_fast(v2 + vr * cp + br);
                const float cc = v0 + (cp - v0) * f;
                const float gr = g * (cc * m - x);         // d/dr
                const float gx = g * (1.f - r);            // d/dx'
                const float gct = gc + g * r * m;          // total d/dc_t
                const float dr = gr * r * (1.f - r);       // d/d(pre-activation of r)
                const float g0 = gct * (1.f - f);          // d/du0
                const float df = gct * (cp - v0) * f * (1.f - f);
                gc = gct * f + df * vf + dr * vr;          // d/dc_{t-1}
                gvf += df * cp; gvr += dr * cp; gbf += df; gbr += dr;
                T* gp = gUb + t * ustep;
                Elem<T>::st(gp, g0);
                Elem<T>::st(gp + D2, df);
                Elem<T>::st(gp + 2 * D2, dr);
                if (K == 3) Elem<T>::st(gXb + t * hstep, gx);
                else Elem<T>::st(gp + 3 * D2, gx);
            }
        }
    };

    load(A, Tn - 1);
    for (int i_hi = Tn - 1; i_hi >= 0; i_hi -= 2 * UN) {
        load(Bt, i_hi - UN);
        compute(A, i_hi);
        load(A, i_hi - 2 * UN);
        compute(Bt, i_hi - UN);
    }
    atomicAdd(&gwc[cd], gvf);
    atomicAdd(&gwc[D2 + cd], gvr);
    atomicAdd(&gbias[cd], gbf);
    atomicAdd(&gbias[D2 + cd], gbr);
}


// ---------------------------------------------------------------------------------------------
// Loader-ring variants (bf16, d % 128 == 0): the production path.
//
// The register-pipelined kernels above are capped by the 64 outstanding vector-memory operations a
// wave may have: 64 x 128 B per wave x (columns / 64) waves = 4 MB in flight chip-wide whatever the
// unroll, i.e. ~1.8 TB/s at the observed latency.  Here a workgroup is 2 consumer waves (128
// columns, one per lane) plus ONE loader wave that only issues LDS-DMA (global_load_lds_dwordx4,
// 1 KiB per instruction = all four streams of one time step for the 128 columns) into an LDS ring,
// so the same 64-deep queue holds 64 KiB.  Hand-off per chunk of CH steps: the loader waits with a
// COUNTED vmcnt for the oldest chunk, all three waves meet at a raw s_barrier, the consumers read
// that chunk with ds_read while the loader refills the slot the consumers released one chunk ago.
// Loads past the end of the sequence are clamped re-loads so the count per chunk stays constant.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float lds_bf16(const unsigned char* p) {
    return __uint_as_float(((uint32_t)*reinterpret_cast<const uint16_t*>(p)) << 16);
}

template <int K>
__global__ __launch_bounds__(192) void sru_fwd_ring_kernel(
    const __hip_bfloat16* __restrict__ U, const __hip_bfloat16* __restrict__ xres,
    const float* __restrict__ wc, const float* __restrict__ bias, const float* __restrict__ mask,
    int Tn, int B, int d, __hip_bfloat16* __restrict__ h, float* __restrict__ cst) {
    constexpr int CH = 8, R = 8, ROW = 1024;
    __shared__ __attribute__((aligned(16))) unsigned char ring[R * CH * ROW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int D2 = 2 * d;
    const int col0 = blockIdx.x * 128;
    const int b = col0 / D2, cd0 = col0 - b * D2, dir = cd0 / d;
    const int nch = (Tn + CH - 1) / CH;
    const int64_t ustep = (int64_t)B * K * D2, hstep = (int64_t)B * D2;

    if (wave == 2) {
        // ---- loader: lane = (stream, 16-byte piece) ----
        const int stream = lane >> 4, c16 = lane & 15;
        const __hip_bfloat16* base;
        int64_t step;
        if (K == 4 || stream < 3) {
            base = U + (int64_t)b * K * D2 + (int64_t)stream * D2 + cd0 + c16 * 8;
            step = ustep;
        } else {
            base = xres + (int64_t)b * D2 + cd0 + c16 * 8;
            step = hstep;
        }
        auto issue = [&](int c) {
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const int i = min(c * CH + s, Tn - 1);
                const int64_t t = dir ? Tn - 1 - i : i;
                __builtin_amdgcn_global_load_lds((gptr_t)(base + t * step),
                                                 (lptr_t)(ring + (((c % R) * CH) + s) * ROW), 16, 0, 0);
            }
        };
        for (int c = 0; c < R - 1; ++c) issue(c);
        for (int c = 0; c < nch; ++c) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * CH) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(c + R - 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain before the LDS is released
        return;
    }
    // ---- consumers: one column per lane ----
    const int cl = wave * 64 + lane;  // column within the workgroup
    const int cd = cd0 + cl;
    const float vf = wc[cd], vr = wc[D2 + cd];
    const float bf = bias[cd], br = bias[D2 + cd];
    const float m = mask ? mask[(int64_t)b * D2 + cd] : 1.f;
    __hip_bfloat16* hb = h + (int64_t)b * D2 + cd;
    float* cb = cst + (int64_t)b * D2 + cd;
    float c = 0.f;
    const float nvf = -kL2e * vf, nvr = -kL2e * vr, nbf = -kL2e * bf, nbr = -kL2e * br;
    auto chunk = [&](int ch, auto guard) {
        constexpr bool GUARD = decltype(guard)::value;
        const unsigned char* slot = ring + ((ch % R) * CH) * ROW + cl * 2;
        float v0[CH], a1[CH], a2[CH], x[CH];
#pragma unroll
        for (int s = 0; s < CH; ++s) {  // every LDS read of the chunk up front, off the serial chain
            const unsigned char* row = slot + s * ROW;
            v0[s] = lds_bf16(row);
            a1[s] = fmaf(-kL2e, lds_bf16(row + 256), nbf);
            a2[s] = fmaf(-kL2e, lds_bf16(row + 512), nbr);
            x[s] = lds_bf16(row + 768);
        }
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int i = ch * CH + s;
            if (!GUARD || i < Tn) {
                const int64_t t = dir ? Tn - 1 - i : i;
                const float f = sigmoid_neg2(fmaf(nvf, c, a1[s]));
                const float r = sigmoid_neg2(fmaf(nvr, c, a2[s]));
                c = fmaf(c - v0[s], f, v0[s]);
                const float hv = fmaf(fmaf(c, m, -x[s]), r, x[s]);
                hb[t * hstep] = __float2bfloat16(hv);
                cb[t * hstep] = c;
            }
        }
    };
    for (int ch = 0; ch < nch; ++ch) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if ((ch + 1) * CH <= Tn) chunk(ch, std::false_type{});
        else chunk(ch, std::true_type{});
    }
}

template <int K>
__global__ __launch_bounds__(192) void sru_bwd_ring_kernel(
    const __hip_bfloat16* __restrict__ U, const __hip_bfloat16* __restrict__ xres,
    const float* __restrict__ wc, const float* __restrict__ bias, const float* __restrict__ mask,
    const float* __restrict__ cst, const __hip_bfloat16* __restrict__ gh, int Tn, int B, int d,
    __hip_bfloat16* __restrict__ gU, __hip_bfloat16* __restrict__ gxres, float* __restrict__ gwc,
    float* __restrict__ gbias) {
    constexpr int CH = 4, R = 8, ROW = 2048;  // row: u0 u1 u2 x' (4 x 256 B) | gh (256 B) | c_prev f32 (512 B)
    __shared__ __attribute__((aligned(16))) unsigned char ring[R * CH * ROW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int D2 = 2 * d;
    const int col0 = blockIdx.x * 128;
    const int b = col0 / D2, cd0 = col0 - b * D2, dir = cd0 / d;
    const int nch = (Tn + CH - 1) / CH;
    const int64_t ustep = (int64_t)B * K * D2, hstep = (int64_t)B * D2;

    if (wave == 2) {
        const int stream = lane >> 4, c16 = lane & 15;
        const __hip_bfloat16* base;
        int64_t step;
        if (K == 4 || stream < 3) {
            base = U + (int64_t)b * K * D2 + (int64_t)stream * D2 + cd0 + c16 * 8;
            step = ustep;
        } else {
            base = xres + (int64_t)b * D2 + cd0 + c16 * 8;
            step = hstep;
        }
        // second instruction: lanes 0-15 gh (bf16, 8 columns each), lanes 16-47 c_prev (f32, 4 columns each)
        const char* base2 = lane < 16 ? reinterpret_cast<const char*>(gh + (int64_t)b * D2 + cd0 + lane * 8)
                                      : reinterpret_cast<const char*>(cst + (int64_t)b * D2 + cd0 + (lane - 16) * 4);
        const int64_t step2 = lane < 16 ? hstep * 2 : hstep * 4;  // bytes
        auto issue = [&](int c) {
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const int jr = min(c * CH + s, Tn - 1);    // walk the forward order backwards
                const int i = Tn - 1 - jr;
                const int64_t t = dir ? Tn - 1 - i : i;
                unsigned char* row = ring + (((c % R) * CH) + s) * ROW;
                __builtin_amdgcn_global_load_lds((gptr_t)(base + t * step), (lptr_t)row, 16, 0, 0);
                // c_prev: cell state of forward step i-1 (consumer substitutes 0 at i == 0)
                const int ip = max(i - 1, 0);
                const int64_t tp = dir ? Tn - 1 - ip : ip;
                const int64_t t2 = lane < 16 ? t : tp;
                if (lane < 48)
                    __builtin_amdgcn_global_load_lds((gptr_t)(base2 + t2 * step2), (lptr_t)(row + 1024), 16, 0, 0);
            }
        };
        for (int c = 0; c < R - 1; ++c) issue(c);
        for (int c = 0; c < nch; ++c) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * CH * 2) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(c + R - 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const int cl = wave * 64 + lane;
    const int cd = cd0 + cl;
    const float vf = wc[cd], vr = wc[D2 + cd];
    const float bf = bias[cd], br = bias[D2 + cd];
    const float m = mask ? mask[(int64_t)b * D2 + cd] : 1.f;
    __hip_bfloat16* gUb = gU + (int64_t)b * K * D2 + cd;
    __hip_bfloat16* gXb = (K == 3) ? gxres + (int64_t)b * D2 + cd : nullptr;
    float gc = 0.f, gvf = 0.f, gvr = 0.f, gbf = 0.f, gbr = 0.f;
    const float nvf = -kL2e * vf, nvr = -kL2e * vr, nbf = -kL2e * bf, nbr = -kL2e * br;
    auto chunk = [&](int ch, auto guard) {
        constexpr bool GUARD = decltype(guard)::value;
        const unsigned char* slot = ring + ((ch % R) * CH) * ROW;
        float v0[CH], f_[CH], r_[CH], x[CH], g[CH], cp[CH];
#pragma unroll
        for (int s = 0; s < CH; ++s) {  // gates depend only on loaded data: all off the serial chain
            const unsigned char* row = slot + s * ROW;
            const int jr = ch * CH + s;
            v0[s] = lds_bf16(row + cl * 2);
            const float u1 = lds_bf16(row + 256 + cl * 2), u2 = lds_bf16(row + 512 + cl * 2);
            x[s] = lds_bf16(row + 768 + cl * 2);
            g[s] = lds_bf16(row + 1024 + cl * 2);
            cp[s] = *reinterpret_cast<const float*>(row + 1280 + cl * 4);
            if (jr >= Tn - 1) cp[s] = 0.f;  // forward step 0 has no predecessor
            f_[s] = sigmoid_neg2(fmaf(nvf, cp[s], fmaf(-kL2e, u1, nbf)));
            r_[s] = sigmoid_neg2(fmaf(nvr, cp[s], fmaf(-kL2e, u2, nbr)));
        }
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int jr = ch * CH + s;
            if (!GUARD || jr < Tn) {
                const int i = Tn - 1 - jr;
                const int64_t t = dir ? Tn - 1 - i : i;
                const float f = f_[s], r = r_[s];
                const float cc = fmaf(cp[s] - v0[s], f, v0[s]);
                const float gr = g[s] * fmaf(cc, m, -x[s]);
                const float gx = g[s] * (1.f - r);
                const float gct = fmaf(g[s] * r, m, gc);
                const float dr = gr * r * (1.f - r);
                const float g0 = gct * (1.f - f);
                const float df = gct * (cp[s] - v0[s]) * f * (1.f - f);
                gc = fmaf(gct, f, fmaf(df, vf, dr * vr));
                gvf = fmaf(df, cp[s], gvf); gvr = fmaf(dr, cp[s], gvr); gbf += df; gbr += dr;
                __hip_bfloat16* gp = gUb + t * ustep;
                gp[0] = __float2bfloat16(g0);
                gp[D2] = __float2bfloat16(df);
                gp[2 * D2] = __float2bfloat16(dr);
                if (K == 3) gXb[t * hstep] = __float2bfloat16(gx);
                else gp[3 * D2] = __float2bfloat16(gx);
            }
        }
    };
    for (int ch = 0; ch < nch; ++ch) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if ((ch + 1) * CH <= Tn) chunk(ch, std::false_type{});
        else chunk(ch, std::true_type{});
    }
    atomicAdd(&gwc[cd], gvf);
    atomicAdd(&gwc[D2 + cd], gvr);
    atomicAdd(&gbias[cd], gbf);
    atomicAdd(&gbias[D2 + cd], gbr);
}

template <typename T>
static int sru_fwd_t(const void* U, const void* xres, const float* wc, const float* bias,
                     const float* mask, int Tn, int B, int d, int k, void* h, float* c, hipStream_t s) {
    const int blocks = B * 2 * d / 64;
    if (k == 3)
        hipLaunchKernelGGL((sru_fwd_kernel<T, 3, 16>), dim3(blocks), dim3(64), 0, s, (const T*)U,
                           (const T*)xres, wc, bias, mask, Tn, B, d, (T*)h, c);
    else
        hipLaunchKernelGGL((sru_fwd_kernel<T, 4, 16>), dim3(blocks), dim3(64), 0, s, (const T*)U,
                           (const T*)xres, wc, bias, mask, Tn, B, d, (T*)h, c);
    return check_launch("sru_fwd_kernel");
}

template <typename T>
static int sru_bwd_t(const void* U, const void* xres, const float* wc, const float* bias,
                     const float* mask, const float* c, const void* gh, int Tn, int B, int d, int k,
                     void* gU, void* gxres, float* gwc, float* gbias, hipStream_t s) {
    const int blocks = B * 2 * d / 64;
    if (k == 3)
        hipLaunchKernelGGL((sru_bwd_kernel<T, 3, 8>), dim3(blocks), dim3(64), 0, s, (const T*)U,
                           (const T*)xres, wc, bias, mask, c, (const T*)gh, Tn, B, d, (T*)gU,
                           (T*)gxres, gwc, gbias);
    else
        hipLaunchKernelGGL((sru_bwd_kernel<T, 4, 8>), dim3(blocks), dim3(64), 0, s, (const T*)U,
                           (const T*)xres, wc, bias, mask, c, (const T*)gh, Tn, B, d, (T*)gU,
                           (T*)gxres, gwc, gbias);
    return check_launch("sru_bwd_kernel");
}

static int sru_check(const char* who, int Tn, int B, int d, int k, int dtype) {
    SP_CHECK_ARG(Tn >= 0 && B >= 0 && d >= 1, "%s: bad shape T=%d B=%d d=%d", who, Tn, B, d);
    SP_CHECK_ARG(k == 3 || k == 4, "%s: k=%d must be 3 or 4", who, k);
    SP_CHECK_ARG(d % 64 == 0, "%s: d=%d must be a multiple of 64 (one wave = 64 channels)", who, d);
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16, "%s: dtype %d not supported (f32, bf16)", who, dtype);
    return SP_OK;
}

}  // namespace sp

extern "C" int sp_sru_fwd(const void* U, const void* xres, const float* wc, const float* bias,
                          const float* mask, int T, int B, int d, int k, int dtype, void* h,
                          float* c, void* stream) {
    using namespace sp;
    int rc = sru_check("sp_sru_fwd", T, B, d, k, dtype);
    if (rc) return rc;
    if (T == 0 || B == 0) return SP_OK;
    SP_CHECK_ARG(U && wc && bias && h && c && (k == 4 || xres), "sp_sru_fwd: null pointer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SP_F32) return sru_fwd_t<float>(U, xres, wc, bias, mask, T, B, d, k, h, c, s);
    if (d % 128 == 0) {  // loader-ring path
        typedef __hip_bfloat16 bf;
        const int blocks = B * 2 * d / 128;
        if (k == 3)
            hipLaunchKernelGGL((sru_fwd_ring_kernel<3>), dim3(blocks), dim3(192), 0, s, (const bf*)U, (const bf*)xres,
                               wc, bias, mask, T, B, d, (bf*)h, c);
        else
            hipLaunchKernelGGL((sru_fwd_ring_kernel<4>), dim3(blocks), dim3(192), 0, s, (const bf*)U, (const bf*)xres,
                               wc, bias, mask, T, B, d, (bf*)h, c);
        return check_launch("sru_fwd_ring_kernel");
    }
    return sru_fwd_t<__hip_bfloat16>(U, xres, wc, bias, mask, T, B, d, k, h, c, s);
}

extern "C" int sp_sru_bwd(const void* U, const void* xres, const float* wc, const float* bias,
                          const float* mask, const float* c, const void* gh, int T, int B, int d,
                          int k, int dtype, void* gU, void* gxres, float* gwc, float* gbias,
                          void* stream) {
    using namespace sp;
    int rc = sru_check("sp_sru_bwd", T, B, d, k, dtype);
    if (rc) return rc;
    if (T == 0 || B == 0) return SP_OK;
    SP_CHECK_ARG(U && wc && bias && c && gh && gU && gwc && gbias && (k == 4 || (xres && gxres)),
                 "sp_sru_bwd: null pointer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SP_F32)
        return sru_bwd_t<float>(U, xres, wc, bias, mask, c, gh, T, B, d, k, gU, gxres, gwc, gbias, s);
    if (d % 128 == 0) {  // loader-ring path
        typedef __hip_bfloat16 bf;
        const int blocks = B * 2 * d / 128;
        if (k == 3)
            hipLaunchKernelGGL((sru_bwd_ring_kernel<3>), dim3(blocks), dim3(192), 0, s, (const bf*)U, (const bf*)xres,
                               wc, bias, mask, c, (const bf*)gh, T, B, d, (bf*)gU, (bf*)gxres, gwc, gbias);
        else
            hipLaunchKernelGGL((sru_bwd_ring_kernel<4>), dim3(blocks), dim3(192), 0, s, (const bf*)U, (const bf*)xres,
                               wc, bias, mask, c, (const bf*)gh, T, B, d, (bf*)gU, (bf*)gxres, gwc, gbias);
        return check_launch("sru_bwd_ring_kernel");
    }
    return sru_bwd_t<__hip_bfloat16>(U, xres, wc, bias, mask, c, gh, T, B, d, k, gU, gxres, gwc, gbias, s);
}

// ----------------------------------------------------------------------------------------------
// Projection-weight layout conversion.  The state_dict keeps the sru package's layout
// weight[n_in][dir][d][k] (f32, rows in the reference's feature order c*F + f); the kernels and the
// projection GEMM use Wk[n_in'][k][dir][d] in the activation dtype with rows in the activations'
// order f*C + c (C = 1: same order).  One thread per (row, dir, channel): k contiguous floats on
// the state_dict side, k plane-strided elements (coalesced across channels) on the kernel side.
// ----------------------------------------------------------------------------------------------
namespace sp {
template <typename T, int K>
__global__ __launch_bounds__(256) void sru_pack_w_kernel(const float* __restrict__ w, int n_in, int d, int C,
                                                         T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (r_out, dir, dd)
    if (i >= (int64_t)n_in * 2 * d) return;
    const int dd = (int)(i % d), dir = (int)((i / d) & 1), r_out = (int)(i / (2 * d));
    const int F = n_in / C, f = r_out / C, c = r_out - f * C;
    const int r_in = c * F + f;
    const float* src = w + (((int64_t)r_in * 2 + dir) * d + dd) * K;
#pragma unroll
    for (int kk = 0; kk < K; ++kk)
        Elem<T>::st(out + (((int64_t)r_out * K + kk) * 2 + dir) * d + dd, src[kk]);
}

template <typename T, int K>
__global__ __launch_bounds__(256) void sru_unpack_dw_kernel(const T* __restrict__ part, int S, int n_in, int d,
                                                            int C, float* __restrict__ dw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n_in * 2 * d) return;
    const int dd = (int)(i % d), dir = (int)((i / d) & 1), r_out = (int)(i / (2 * d));
    const int F = n_in / C, f = r_out / C, c = r_out - f * C;
    const int r_in = c * F + f;
    const int64_t plane = (int64_t)n_in * K * 2 * d;
    float acc[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) acc[kk] = 0.f;
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            acc[kk] += Elem<T>::ld(part + s * plane + (((int64_t)r_out * K + kk) * 2 + dir) * d + dd);
    }
    float* dst = dw + (((int64_t)r_in * 2 + dir) * d + dd) * K;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) dst[kk] = acc[kk];
}

static int pack_check(const char* fn, int n_in, int d, int k, int C) {
    SP_CHECK_ARG(n_in > 0 && d > 0 && (k == 3 || k == 4), "%s: bad shape n_in=%d d=%d k=%d", fn, n_in, d, k);
    SP_CHECK_ARG(C >= 1 && n_in % C == 0, "%s: C=%d does not divide n_in=%d", fn, C, n_in);
    return SP_OK;
}
}  // namespace sp

extern "C" int sp_sru_pack_w(const float* weight, int n_in, int d, int k, int C, int dtype, void* out,
                             void* stream) {
    using namespace sp;
    int rc = pack_check("sp_sru_pack_w", n_in, d, k, C);
    if (rc) return rc;
    SP_CHECK_ARG(weight && out, "sp_sru_pack_w: null pointer");
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16, "sp_sru_pack_w: dtype %d", dtype);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)(((int64_t)n_in * 2 * d + 255) / 256);
#define SP_PACK(T, K) hipLaunchKernelGGL((sru_pack_w_kernel<T, K>), dim3(blocks), dim3(256), 0, s, weight, n_in, d, C, (T*)out)
    if (dtype == SP_F32) { if (k == 3) SP_PACK(float, 3); else SP_PACK(float, 4); }
    else { if (k == 3) SP_PACK(__hip_bfloat16, 3); else SP_PACK(__hip_bfloat16, 4); }
#undef SP_PACK
    return check_launch("sru_pack_w_kernel");
}

extern "C" int sp_sru_unpack_dw(const void* partials, int S, int n_in, int d, int k, int C, int dtype,
                                float* dweight, void* stream) {
    using namespace sp;
    int rc = pack_check("sp_sru_unpack_dw", n_in, d, k, C);
    if (rc) return rc;
    SP_CHECK_ARG(partials && dweight && S >= 1, "sp_sru_unpack_dw: null pointer or S=%d", S);
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16, "sp_sru_unpack_dw: dtype %d", dtype);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)(((int64_t)n_in * 2 * d + 255) / 256);
#define SP_UNPACK(T, K) hipLaunchKernelGGL((sru_unpack_dw_kernel<T, K>), dim3(blocks), dim3(256), 0, s, (const T*)partials, S, n_in, d, C, dweight)
    if (dtype == SP_F32) { if (k == 3) SP_UNPACK(float, 3); else SP_UNPACK(float, 4); }
    else { if (k == 3) SP_UNPACK(__hip_bfloat16, 3); else SP_UNPACK(__hip_bfloat16, 4); }
#undef SP_UNPACK
    return check_launch("sru_unpack_dw_kernel");
}
