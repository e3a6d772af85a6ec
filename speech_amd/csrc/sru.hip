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
    float* cb = cst ? cst + (int64_t)b * D2 + cd : nullptr;     // NULL: inference, nothing kept for a backward

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
                if (cb) cb[t * hstep] = c;
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
// Loader-ring kernels (bf16): the production path.
//
// The register-pipelined kernels above are capped by the 64 outstanding vector-memory operations a
// wave may have: 64 x 128 B per wave x (columns / 64) waves = 4 MB in flight chip-wide whatever the
// unroll, i.e. ~1.8 TB/s at the observed latency.  Here ONE loader wave per workgroup only issues
// LDS-DMA (global_load_lds_dwordx4, up to 1 KiB per instruction) into an LDS ring, so the same
// 64-deep queue holds up to 64 KiB.  Hand-off per chunk of CH steps: the loader waits with a COUNTED
// vmcnt for the oldest chunk, all waves meet at a raw s_barrier, the consumers read that chunk with
// ds_read while the loader refills the slot the consumers released one chunk ago.  Loads past the
// end of the sequence are clamped re-loads so the count per chunk stays constant.
//
// Cell states on this path are SAVED in bf16 (the recurrence itself runs in f32 registers): the sru
// package keeps c in the activation dtype too (its kernels are templated on one scalar type), and it
// takes 49 MB per layer off the forward's stores and the backward's loads at the reference's sizes.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float lds_bf16(const unsigned char* p) {
    return __uint_as_float(((uint32_t)*reinterpret_cast<const uint16_t*>(p)) << 16);
}

// ---------------------------------------------------------------------------------------------
// Role-split forward (bf16, the production path): one workgroup = 64 columns and three waves.
//   loader  LDS-DMA only: one global_load_lds_dwordx4 brings TWO time steps of the four streams
//           (u0 u1 u2 x', 4 x 128 B each) into the ring.
//   chain   the serial part and nothing else: f = sigmoid(u1 + v_f c' + b_f), c = u0 + (c' - u0) f
//           (fma, exp, add, rcp, sub, fma per step); c goes to an LDS ring.
//   output  one chunk behind: r = sigmoid(u2 + v_r c' + b_r), h = (c m - x') r + x' -- independent
//           across steps -- then stores the chunk transposed through LDS: 16 bytes per lane
//           (1 store of h and 1 of c per 8 steps instead of 16 narrow ones).
// Chain and output in ONE wave would be issue-bound at ~40 instructions per step with only two of a
// CU's four SIMDs occupied; here the chain wave carries ~13.
// One raw s_barrier per chunk of 8 steps hands chunk j to the chain wave, chunk j-1 to the output
// wave and slot (j-2) % R back to the loader.
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(192) void sru_fwd_split_kernel(
    const __hip_bfloat16* __restrict__ U, const __hip_bfloat16* __restrict__ xres,
    const float* __restrict__ wc, const float* __restrict__ bias, const float* __restrict__ mask,
    int Tn, int B, int d, __hip_bfloat16* __restrict__ h, __hip_bfloat16* __restrict__ cst) {
    constexpr int CH = 8, R = 7, ROW = 512, R2 = 4;   // (R-2) chunks = 20 DMA ops = 20 KB in flight per workgroup; 38 KB of LDS: four workgroups per CU
    __shared__ __attribute__((aligned(16))) unsigned char ring[R * CH * ROW];   // 28 KB
    __shared__ __attribute__((aligned(16))) float cring[R2][CH][64];            // 8 KB
    __shared__ __attribute__((aligned(16))) unsigned short htile[CH][64];       // 1 KB
    __shared__ __attribute__((aligned(16))) unsigned short ctile[CH][64];       // 1 KB
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int D2 = 2 * d;
    const int col0 = blockIdx.x * 64;
    const int b = col0 / D2, cd0 = col0 - b * D2, dir = cd0 / d;
    const int nch = (Tn + CH - 1) / CH;
    const int64_t ustep = (int64_t)B * K * D2, hstep = (int64_t)B * D2;

    if (wave == 0) {
        // ---- loader: lane = (step parity, stream, 16-byte piece) ----
        const int sp = lane >> 5, stream = (lane >> 3) & 3, c8 = lane & 7;
        const __hip_bfloat16* base;
        int64_t step;
        if (K == 4 || stream < 3) {
            base = U + (int64_t)b * K * D2 + (int64_t)stream * D2 + cd0 + c8 * 8;
            step = ustep;
        } else {
            base = xres + (int64_t)b * D2 + cd0 + c8 * 8;
            step = hstep;
        }
        auto issue = [&](int c) {
#pragma unroll
            for (int s2 = 0; s2 < CH / 2; ++s2) {
                const int i = min(c * CH + 2 * s2 + sp, Tn - 1);   // past the end: clamped re-loads
                const int64_t t = dir ? Tn - 1 - i : i;
                __builtin_amdgcn_global_load_lds((gptr_t)(base + t * step),
                                                 (lptr_t)(ring + (((c % R) * CH) + 2 * s2) * ROW), 16, 0, 0);
            }
        };
        for (int c = 0; c < R - 2; ++c) issue(c);
        for (int j = 0; j <= nch; ++j) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * (CH / 2)) : "memory");   // chunk j landed
            __builtin_amdgcn_s_barrier();
            issue(j + R - 2);   // into the slot the output wave released at this barrier
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain before the LDS is released
        return;
    }
    const int cd = cd0 + lane;
    if (wave == 1) {
        // ---- chain ----
        const float nvf = -kL2e * wc[cd], nbf = -kL2e * bias[cd];
        float c = 0.f;
        for (int j = 0; j <= nch; ++j) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (j < nch) {
                const unsigned char* slot = ring + ((j % R) * CH) * ROW + lane * 2;
                float v0[CH], a1[CH];
#pragma unroll
                for (int s = 0; s < CH; ++s) {
                    v0[s] = lds_bf16(slot + s * ROW);
                    a1[s] = fmaf(-kL2e, lds_bf16(slot + s * ROW + 128), nbf);
                }
#pragma unroll
                for (int s = 0; s < CH; ++s) {
                    const float f = sigmoid_neg2(fmaf(nvf, c, a1[s]));
                    c = fmaf(c - v0[s], f, v0[s]);
                    cring[j % R2][s][lane] = c;
                }
            }
        }
        return;
    }
    // ---- output ----
    const float nvr = -kL2e * wc[D2 + cd], nbr = -kL2e * bias[D2 + cd];
    const float m = mask ? mask[(int64_t)b * D2 + cd] : 1.f;
    __hip_bfloat16* hb = h + (int64_t)b * D2 + cd0;
    __hip_bfloat16* cb = cst ? cst + (int64_t)b * D2 + cd0 : nullptr;   // NULL: inference, c is not kept
    float cprev = 0.f;
    for (int j = 0; j <= nch; ++j) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j >= 1) {
            const int cj = j - 1;
            const unsigned char* slot = ring + ((cj % R) * CH) * ROW + lane * 2;
            float a2[CH], x[CH], cc[CH];
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                a2[s] = fmaf(-kL2e, lds_bf16(slot + s * ROW + 256), nbr);
                x[s] = lds_bf16(slot + s * ROW + 384);
                cc[s] = cring[cj % R2][s][lane];
            }
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const float r = sigmoid_neg2(fmaf(nvr, cprev, a2[s]));
                const float hv = fmaf(fmaf(cc[s], m, -x[s]), r, x[s]);
                htile[s][lane] = __builtin_bit_cast(unsigned short, __float2bfloat16(hv));
                ctile[s][lane] = __builtin_bit_cast(unsigned short, __float2bfloat16(cc[s]));
                cprev = cc[s];
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {   // h and c: lane = (step, 8-column piece), 16 bytes each
                const int s = lane >> 3, p8 = lane & 7;
                const int i = cj * CH + s;
                if (i < Tn) {
                    const int64_t t = dir ? Tn - 1 - i : i;
                    *reinterpret_cast<uint4*>(hb + t * hstep + p8 * 8) = *reinterpret_cast<const uint4*>(&htile[s][p8 * 8]);
                    if (cb) *reinterpret_cast<uint4*>(cb + t * hstep + p8 * 8) = *reinterpret_cast<const uint4*>(&ctile[s][p8 * 8]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Role-split backward (bf16).  With c' saved, everything except the cell-state gradient is
// independent across time; that gradient obeys gc' = (gc + G) A + Bv with
//   f, r as in the forward,  kappa = (c' - u0) f (1 - f),  A = f + kappa v_f,
//   dr = g (c m - x') r (1 - r),  G = g r m,  Bv = dr v_r,
// so the serial chain is an add and an fma per step.  One workgroup = 64 columns, four waves,
// chunks of 4 steps, one raw s_barrier per chunk; in interval j
//   loader  refills the ring slot of chunk j-2 with chunk j+R-2 (one LDS-DMA per step: u0 u1 u2 x'
//           gh c', 768 B);
//   f-side  chunk j:   f, 1-f, kappa, kappa c', A, q = c m - x'  -> LDS;  stores g0/df of chunk j-3;
//   r-side  chunk j-1: r, dr, gx, G, Bv (q from the f-side) -> LDS; stores dr/gx; sums for v_r, b_r;
//   chain   chunk j-2: gct = gc + G, gc = gct A + Bv, g0 = gct (1-f), df = gct kappa -> LDS tiles;
//           sums for v_f, b_f.
// Stores leave transposed through LDS: 16 bytes per lane, one vector-memory op per wave and chunk.
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void sru_bwd_split_kernel(
    const __hip_bfloat16* __restrict__ U, const __hip_bfloat16* __restrict__ xres,
    const float* __restrict__ wc, const float* __restrict__ bias, const float* __restrict__ mask,
    const __hip_bfloat16* __restrict__ cst, const __hip_bfloat16* __restrict__ gh, int Tn, int B, int d,
    __hip_bfloat16* __restrict__ gU, __hip_bfloat16* __restrict__ gxres, float* __restrict__ gwc,
    float* __restrict__ gbias) {
    constexpr int CH = 4, R = 12, ROW = 1024;   // (R-2) chunks = 40 DMA ops = 35 KB in flight per workgroup
    __shared__ __attribute__((aligned(16))) unsigned char ring[R * CH * ROW];            // 48 KB
    __shared__ __attribute__((aligned(16))) float sq[2][CH][64];                          // f-side -> r-side
    __shared__ __attribute__((aligned(16))) float sA[3][CH][64], sOmf[3][CH][64], sKap[3][CH][64], sKcp[3][CH][64];   // f-side -> chain
    __shared__ __attribute__((aligned(16))) float sG[2][CH][64], sBv[2][CH][64];          // r-side -> chain
    __shared__ __attribute__((aligned(16))) unsigned short tg[2][2][CH][64];              // chain -> f-side: g0 | df
    __shared__ __attribute__((aligned(16))) unsigned short tr[2][CH][64];                 // r-side: dr | gx
    // Measured placement (tools/ubench/hwid.hip): workgroups b and b+256 share a CU, the first one's waves
    // 0..3 sit on SIMDs 0,2,1,3 and the second one's on 2,1,3,0.  The second workgroup's roles are
    // permuted so that the SIMDs carry loader+f-side, f-side+loader, r-side+chain, chain+r-side: the two
    // heaviest roles (f-side, r-side) never share a SIMD and the f-side gets the lightest partner.
    const int w0 = threadIdx.x >> 6;
    const int wave = ((blockIdx.x >> 8) & 1) ? ((4 - w0) & 3) : w0, lane = threadIdx.x & 63;
    const int D2 = 2 * d;
    const int col0 = blockIdx.x * 64;
    const int b = col0 / D2, cd0 = col0 - b * D2, dir = cd0 / d;
    const int nch = (Tn + CH - 1) / CH;
    const int nint = nch + 3;   // barriers every wave executes
    const int64_t ustep = (int64_t)B * K * D2, hstep = (int64_t)B * D2;
    // time index of backward step jr (jr = 0 is the last forward step)
    auto t_of = [&](int jr) { const int i = Tn - 1 - jr; return (int64_t)(dir ? Tn - 1 - i : i); };

    if (wave == 0) {
        // ---- loader: lanes 0-31 (stream, piece) of U / x', 32-39 gh, 40-47 c' (bf16), 48-63 idle ----
        const char* base;
        int64_t stepb;   // bytes per time step
        if (lane < 32) {
            const int stream = lane >> 3, c8 = lane & 7;
            if (K == 4 || stream < 3) {
                base = reinterpret_cast<const char*>(U + (int64_t)b * K * D2 + (int64_t)stream * D2 + cd0 + c8 * 8);
                stepb = ustep * 2;
            } else {
                base = reinterpret_cast<const char*>(xres + (int64_t)b * D2 + cd0 + c8 * 8);
                stepb = hstep * 2;
            }
        } else if (lane < 40) {
            base = reinterpret_cast<const char*>(gh + (int64_t)b * D2 + cd0 + (lane - 32) * 8);
            stepb = hstep * 2;
        } else {
            base = reinterpret_cast<const char*>(cst + (int64_t)b * D2 + cd0 + (min(lane, 47) - 40) * 8);
            stepb = hstep * 2;
        }
        // backward step jr reads forward step i = Tn-1-jr (c': i-1, clamped at 0), i.e. time index i or
        // Tn-1-i by direction: the source pointer moves by one step per load until it reaches its last row
        const bool is_cp = lane >= 40;
        const int i0 = is_cp ? max(Tn - 2, 0) : Tn - 1;   // forward step read by backward step 0
        const char* src = base + (int64_t)(dir ? Tn - 1 - i0 : i0) * stepb;
        const int64_t delta = dir ? stepb : -stepb;
        int left = i0;   // moves still to make
        auto issue = [&](int c) {
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                if (lane < 48)
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + (((c % R) * CH) + s) * ROW), 16, 0, 0);
                const bool move = left > 0;   // past the first forward step: clamped re-loads
                src += move ? delta : 0;
                left -= move ? 1 : 0;
            }
        };
        for (int c = 0; c < R - 2; ++c) issue(c);
        for (int j = 0; j < nint; ++j) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * CH) : "memory");   // chunk j landed
            __builtin_amdgcn_s_barrier();
            issue(j + R - 2);   // slot of chunk j-2: last read by the r-side in interval j-1
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // The f-side and r-side are independent across time, so their lanes are laid out for the LDS, not
    // for the recurrence: lane = (step parity hp, column pair ci) handles columns 2ci, 2ci+1 of steps
    // hp and 2 + hp of a chunk -- one 32-bit ds_read brings two steps x 64 columns of a stream (the
    // 16-bit reads of a lane-per-column layout made the LDS pipe, ~80 % busy, the limiter of this
    // kernel) and every hand-off array is written 8 bytes per lane.
    const int hp = lane >> 5, ci = lane & 31;
    auto lo16 = [](uint32_t w) { return __uint_as_float(w << 16); };
    auto hi16 = [](uint32_t w) { return __uint_as_float(w & 0xffff0000u); };
    if (wave == 1) {
        // ---- f-side ----
        const int cA = cd0 + 2 * ci;
        const float vf[2] = {wc[cA], wc[cA + 1]};
        const float nvf[2] = {-kL2e * vf[0], -kL2e * vf[1]};
        const float nbf[2] = {-kL2e * bias[cA], -kL2e * bias[cA + 1]};
        const float m[2] = {mask ? mask[(int64_t)b * D2 + cA] : 1.f, mask ? mask[(int64_t)b * D2 + cA + 1] : 1.f};
        __hip_bfloat16* gUb = gU + (int64_t)b * K * D2 + cd0;
        for (int j = 0; j < nint; ++j) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (j < nch) {
                const unsigned char* slot = ring + ((j % R) * CH) * ROW + ci * 4;
                uint32_t w0[2], w1[2], wx[2], wp[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const unsigned char* row = slot + (2 * p + hp) * ROW;
                    w0[p] = *reinterpret_cast<const uint32_t*>(row);
                    w1[p] = *reinterpret_cast<const uint32_t*>(row + 128);
                    wx[p] = *reinterpret_cast<const uint32_t*>(row + 384);
                    wp[p] = *reinterpret_cast<const uint32_t*>(row + 640);
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int st = 2 * p + hp;
                    const bool first = j * CH + st >= Tn - 1;   // forward step 0 has no predecessor
                    float2 vA, vO, vK, vC, vQ;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float u0 = e ? hi16(w0[p]) : lo16(w0[p]);
                        const float a1 = fmaf(-kL2e, e ? hi16(w1[p]) : lo16(w1[p]), nbf[e]);
                        const float x = e ? hi16(wx[p]) : lo16(wx[p]);
                        const float cp = first ? 0.f : (e ? hi16(wp[p]) : lo16(wp[p]));
                        const float f = sigmoid_neg2(fmaf(nvf[e], cp, a1));
                        const float omf = 1.f - f;
                        const float d0 = cp - u0;
                        const float kap = d0 * f * omf;
                        (e ? vA.y : vA.x) = fmaf(kap, vf[e], f);
                        (e ? vO.y : vO.x) = omf;
                        (e ? vK.y : vK.x) = kap;
                        (e ? vC.y : vC.x) = kap * cp;
                        (e ? vQ.y : vQ.x) = fmaf(fmaf(d0, f, u0), m[e], -x);
                    }
                    *reinterpret_cast<float2*>(&sA[j % 3][st][2 * ci]) = vA;
                    *reinterpret_cast<float2*>(&sOmf[j % 3][st][2 * ci]) = vO;
                    *reinterpret_cast<float2*>(&sKap[j % 3][st][2 * ci]) = vK;
                    *reinterpret_cast<float2*>(&sKcp[j % 3][st][2 * ci]) = vC;
                    *reinterpret_cast<float2*>(&sq[j & 1][st][2 * ci]) = vQ;
                }
            }
            const int cs = j - 3;   // chunk whose g0/df tiles the chain wave finished in interval j-1
            if (cs >= 0 && cs < nch) {
                const int pl = lane >> 5, s = (lane >> 3) & 3, p8 = lane & 7;   // plane 0: g0, 1: df
                const int jr = cs * CH + s;
                if (jr < Tn)
                    *reinterpret_cast<uint4*>(gUb + t_of(jr) * ustep + (int64_t)pl * D2 + p8 * 8) =
                        *reinterpret_cast<const uint4*>(&tg[cs & 1][pl][s][p8 * 8]);
            }
        }
        return;
    }
    if (wave == 2) {
        // ---- r-side ----
        const int cA = cd0 + 2 * ci;
        const float vr[2] = {wc[D2 + cA], wc[D2 + cA + 1]};
        const float nvr[2] = {-kL2e * vr[0], -kL2e * vr[1]};
        const float nbr[2] = {-kL2e * bias[D2 + cA], -kL2e * bias[D2 + cA + 1]};
        const float m[2] = {mask ? mask[(int64_t)b * D2 + cA] : 1.f, mask ? mask[(int64_t)b * D2 + cA + 1] : 1.f};
        // lanes 0-31 store dr (plane 2 of gU), lanes 32-63 gx (gxres, or plane 3 when K == 4)
        __hip_bfloat16* sbase;
        int64_t sstep;
        if (lane < 32 || K == 4) {
            sbase = gU + (int64_t)b * K * D2 + (int64_t)(lane < 32 ? 2 : 3) * D2 + cd0;
            sstep = ustep;
        } else {
            sbase = gxres + (int64_t)b * D2 + cd0;
            sstep = hstep;
        }
        float gvr[2] = {0.f, 0.f}, gbr[2] = {0.f, 0.f};
        for (int j = 0; j < nint; ++j) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int c = j - 1;
            if (c >= 0 && c < nch) {
                const unsigned char* slot = ring + ((c % R) * CH) * ROW + ci * 4;
                uint32_t w2[2], wg[2], wp[2];
                float2 q[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const unsigned char* row = slot + (2 * p + hp) * ROW;
                    w2[p] = *reinterpret_cast<const uint32_t*>(row + 256);
                    wg[p] = *reinterpret_cast<const uint32_t*>(row + 512);
                    wp[p] = *reinterpret_cast<const uint32_t*>(row + 640);
                    q[p] = *reinterpret_cast<const float2*>(&sq[c & 1][2 * p + hp][2 * ci]);
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int st = 2 * p + hp;
                    const bool first = c * CH + st >= Tn - 1;
                    const bool live = c * CH + st < Tn;
                    float2 vG, vB;
                    uint32_t tdr = 0, tgx = 0;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float a2 = fmaf(-kL2e, e ? hi16(w2[p]) : lo16(w2[p]), nbr[e]);
                        const float g = e ? hi16(wg[p]) : lo16(wg[p]);
                        const float cp = first ? 0.f : (e ? hi16(wp[p]) : lo16(wp[p]));
                        const float r = sigmoid_neg2(fmaf(nvr[e], cp, a2));
                        const float omr = 1.f - r;
                        const float gr = g * (e ? q[p].y : q[p].x);
                        const float dr = gr * r * omr;
                        (e ? vG.y : vG.x) = g * r * m[e];
                        (e ? vB.y : vB.x) = dr * vr[e];
                        tdr |= (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(dr)) << (16 * e);
                        tgx |= (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(g * omr)) << (16 * e);
                        const float drv = live ? dr : 0.f;   // select, not a branch: the steps interleave
                        gvr[e] = fmaf(drv, cp, gvr[e]);
                        gbr[e] += drv;
                    }
                    *reinterpret_cast<float2*>(&sG[c & 1][st][2 * ci]) = vG;
                    *reinterpret_cast<float2*>(&sBv[c & 1][st][2 * ci]) = vB;
                    *reinterpret_cast<uint32_t*>(&tr[0][st][2 * ci]) = tdr;
                    *reinterpret_cast<uint32_t*>(&tr[1][st][2 * ci]) = tgx;
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                {
                    const int pl = lane >> 5, s = (lane >> 3) & 3, p8 = lane & 7;
                    const int jr = c * CH + s;
                    if (jr < Tn)
                        *reinterpret_cast<uint4*>(sbase + t_of(jr) * sstep + p8 * 8) =
                            *reinterpret_cast<const uint4*>(&tr[pl][s][p8 * 8]);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {   // the two step parities of a column pair sit 32 lanes apart
            gvr[e] += __shfl_xor(gvr[e], 32);
            gbr[e] += __shfl_xor(gbr[e], 32);
        }
        if (hp == 0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                atomicAdd(&gwc[D2 + cA + e], gvr[e]);
                atomicAdd(&gbias[D2 + cA + e], gbr[e]);
            }
        }
        return;
    }
    const int cd = cd0 + lane;
    // ---- chain ----
    float gc = 0.f, gvf = 0.f, gbf = 0.f;
    for (int j = 0; j < nint; ++j) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int c = j - 2;
        if (c >= 0 && c < nch) {
            float G[CH], A[CH], Bv[CH], omf[CH], kap[CH], kcp[CH];
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                G[s] = sG[c & 1][s][lane];
                Bv[s] = sBv[c & 1][s][lane];
                A[s] = sA[c % 3][s][lane];
                omf[s] = sOmf[c % 3][s][lane];
                kap[s] = sKap[c % 3][s][lane];
                kcp[s] = sKcp[c % 3][s][lane];
            }
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const float gct = gc + G[s];
                gc = fmaf(gct, A[s], Bv[s]);
                const float df = gct * kap[s];
                tg[c & 1][0][s][lane] = __builtin_bit_cast(unsigned short, __float2bfloat16(gct * omf[s]));
                tg[c & 1][1][s][lane] = __builtin_bit_cast(unsigned short, __float2bfloat16(df));
                const float vs = (c * CH + s < Tn) ? 1.f : 0.f;   // select, not a branch
                gvf = fmaf(gct * vs, kcp[s], gvf);
                gbf = fmaf(df, vs, gbf);
            }
        }
    }
    atomicAdd(&gwc[cd], gvf);
    atomicAdd(&gbias[cd], gbf);
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
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16 || dtype == SP_F16, "%s: dtype %d not supported (f32, bf16, f16)", who, dtype);
    return SP_OK;
}

}  // namespace sp

extern "C" int sp_sru_fwd(const void* U, const void* xres, const float* wc, const float* bias,
                          const float* mask, int T, int B, int d, int k, int dtype, void* h,
                          void* c, void* stream) {
    using namespace sp;
    int rc = sru_check("sp_sru_fwd", T, B, d, k, dtype);
    if (rc) return rc;
    if (T == 0 || B == 0) return SP_OK;
    SP_CHECK_ARG(U && wc && bias && h && (k == 4 || xres), "sp_sru_fwd: null pointer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SP_F32) return sru_fwd_t<float>(U, xres, wc, bias, mask, T, B, d, k, h, (float*)c, s);
    if (dtype == SP_F16) return sru_fwd_t<__half>(U, xres, wc, bias, mask, T, B, d, k, h, (float*)c, s);   // c stays f32
    {   // role-split path (any d % 64 == 0)
        typedef __hip_bfloat16 bf;
        const int blocks = B * 2 * d / 64;
        if (k == 3)
            hipLaunchKernelGGL((sru_fwd_split_kernel<3>), dim3(blocks), dim3(192), 0, s, (const bf*)U, (const bf*)xres,
                               wc, bias, mask, T, B, d, (bf*)h, (bf*)c);
        else
            hipLaunchKernelGGL((sru_fwd_split_kernel<4>), dim3(blocks), dim3(192), 0, s, (const bf*)U, (const bf*)xres,
                               wc, bias, mask, T, B, d, (bf*)h, (bf*)c);
        return check_launch("sru_fwd_split_kernel");
    }
}

extern "C" int sp_sru_bwd(const void* U, const void* xres, const float* wc, const float* bias,
                          const float* mask, const void* c, const void* gh, int T, int B, int d,
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
        return sru_bwd_t<float>(U, xres, wc, bias, mask, (const float*)c, gh, T, B, d, k, gU, gxres, gwc, gbias, s);
    if (dtype == SP_F16)
        return sru_bwd_t<__half>(U, xres, wc, bias, mask, (const float*)c, gh, T, B, d, k, gU, gxres, gwc, gbias, s);
    {   // role-split path (any d % 64 == 0)
        typedef __hip_bfloat16 bf;
        const int blocks = B * 2 * d / 64;
        if (k == 3)
            hipLaunchKernelGGL((sru_bwd_split_kernel<3>), dim3(blocks), dim3(256), 0, s, (const bf*)U, (const bf*)xres,
                               wc, bias, mask, (const bf*)c, (const bf*)gh, T, B, d, (bf*)gU, (bf*)gxres, gwc, gbias);
        else
            hipLaunchKernelGGL((sru_bwd_split_kernel<4>), dim3(blocks), dim3(256), 0, s, (const bf*)U, (const bf*)xres,
                               wc, bias, mask, (const bf*)c, (const bf*)gh, T, B, d, (bf*)gU, (bf*)gxres, gwc, gbias);
        return check_launch("sru_bwd_split_kernel");
    }
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

// Transposed pack: out_t[n][r_out] (n = (kk*2 + dir)*d + dd, row pitch ldo), the K-contiguous "B" operand
// of sp_gemm_nt_bf16 -- and, in the same pass over the weight, optionally sru_pack_w's layout out_w as well
// (the data gradient's library GEMM wants that one).  One workgroup moves R rows x 64 dd x K through LDS
// (R x sizeof(T) = 128 bytes): reads are 64*K consecutive floats of one weight row, writes are 128-byte
// runs of one output row.  Needs n_in % R == 0, d % 64 == 0.
constexpr int kPackBatch = 16;
struct PackBatch { const float* w[kPackBatch]; void* out_t[kPackBatch]; void* out_w[kPackBatch]; };

// blockIdx.z = 2 * layer + dir: several layers of one shape in a single launch (one layer's 256 workgroups
// leave the memory system mostly idle: 10 us for 25 MB; nine layers together run at the streaming rate)
template <typename T, int K>
__global__ __launch_bounds__(256) void sru_pack_wt_kernel(PackBatch pb, int n_in, int d, int C, int64_t ldo) {
    constexpr int R = 128 / (int)sizeof(T), W = 64 * K, E = 4 / (int)sizeof(T);   // E elements per 4-byte store
    __shared__ T tile[R][W + 2];
    const int r0 = blockIdx.x * R, dd0 = blockIdx.y * 64, dir = blockIdx.z & 1, layer = blockIdx.z >> 1;
    const float* __restrict__ w = pb.w[layer];
    T* __restrict__ out_t = reinterpret_cast<T*>(pb.out_t[layer]);
    T* __restrict__ out_w = reinterpret_cast<T*>(pb.out_w[layer]);
    const int F = n_in / C;
    // all of a thread's 16-byte loads are requested before the first is used
    constexpr int W4 = W / 4, NV = R * W4 / 256;
    static_assert(R * W4 % 256 == 0, "tile must divide over the workgroup");
    float4 v4[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + 256 * i, row = idx / W4, c4 = idx - row * W4;
        const int r_out = r0 + row, f = r_out / C, c = r_out - f * C;
        v4[i] = *reinterpret_cast<const float4*>(w + (((int64_t)(c * F + f) * 2 + dir) * d + dd0) * K + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + 256 * i, row = idx / W4, c4 = idx - row * W4;
        Elem<T>::st(&tile[row][c4 * 4 + 0], v4[i].x);
        Elem<T>::st(&tile[row][c4 * 4 + 1], v4[i].y);
        Elem<T>::st(&tile[row][c4 * 4 + 2], v4[i].z);
        Elem<T>::st(&tile[row][c4 * 4 + 3], v4[i].w);
    }
    __syncthreads();
    const int lane32 = threadIdx.x & 31, sub = threadIdx.x >> 5;      // 8 half-waves, one 128-byte run each
    // out_t: for every (kk, dd) the R rows are consecutive
    for (int pair = sub; pair < W; pair += 8) {
        const int kk = pair / 64, ddl = pair - kk * 64;
        T v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = tile[lane32 * E + e][ddl * K + kk];
        T* dst = out_t + ((int64_t)(kk * 2 + dir) * d + dd0 + ddl) * ldo + r0 + lane32 * E;
        *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(v);
    }
    if (out_w) {   // out_w: for every (row, kk) the 64 dd are consecutive; a half-wave moves 32 E of them
        constexpr int RUNS = 64 / (32 * E);                           // half-wave runs per (row, kk): 1 (bf16) or 2 (f32)
        for (int item = sub; item < R * K * RUNS; item += 8) {
            const int run = item % RUNS, rk = item / RUNS, kk = rk % K, row = rk / K;
            const int ddl = run * 32 * E + lane32 * E;
            T v[E];
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = tile[row][(ddl + e) * K + kk];
            T* dst = out_w + (((int64_t)(r0 + row) * K + kk) * 2 + dir) * d + dd0 + ddl;
            *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(v);
        }
    }
}

template <typename T, int K>
__global__ __launch_bounds__(256) void sru_unpack_dw_kernel(const T* __restrict__ part, int S, int n_in, int d,
                                                            int C, int accumulate, float* __restrict__ dw) {
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
    for (int kk = 0; kk < K; ++kk) dst[kk] = accumulate ? dst[kk] + acc[kk] : acc[kk];
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
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16 || dtype == SP_F16, "sp_sru_pack_w: dtype %d", dtype);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)(((int64_t)n_in * 2 * d + 255) / 256);
#define SP_PACK(T, K) hipLaunchKernelGGL((sru_pack_w_kernel<T, K>), dim3(blocks), dim3(256), 0, s, weight, n_in, d, C, (T*)out)
    if (dtype == SP_F32) { if (k == 3) SP_PACK(float, 3); else SP_PACK(float, 4); }
    else if (dtype == SP_F16) { if (k == 3) SP_PACK(__half, 3); else SP_PACK(__half, 4); }
    else { if (k == 3) SP_PACK(__hip_bfloat16, 3); else SP_PACK(__hip_bfloat16, 4); }
#undef SP_PACK
    return check_launch("sru_pack_w_kernel");
}

extern "C" int sp_sru_pack_wt_batched(const float* const* weights, int L, int n_in, int d, int k, int C, int dtype,
                                      void* const* out_t, int64_t ldo, void* const* out_w, void* stream) {
    using namespace sp;
    int rc = pack_check("sp_sru_pack_wt", n_in, d, k, C);
    if (rc) return rc;
    SP_CHECK_ARG(weights && out_t && L >= 1 && L <= kPackBatch, "sp_sru_pack_wt: null pointer or L=%d outside [1,%d]", L, kPackBatch);
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16 || dtype == SP_F16, "sp_sru_pack_wt: dtype %d", dtype);
    const int R = dtype == SP_F32 ? 32 : 64;
    SP_CHECK_ARG(ldo >= n_in && ldo % 2 == 0, "sp_sru_pack_wt: ldo=%lld must be even and >= n_in=%d", (long long)ldo, n_in);
    SP_CHECK_ARG(n_in % R == 0 && d % 64 == 0, "sp_sru_pack_wt: n_in %% %d and d %% 64 must be 0 (n_in=%d d=%d)", R, n_in, d);
    PackBatch pb;
    for (int l = 0; l < kPackBatch; ++l) {
        const int q = l < L ? l : 0;
        pb.w[l] = weights[q]; pb.out_t[l] = out_t[q]; pb.out_w[l] = out_w ? out_w[q] : nullptr;
        SP_CHECK_ARG(pb.w[l] && pb.out_t[l], "sp_sru_pack_wt: null pointer (layer %d)", q);
        SP_CHECK_ARG((((uintptr_t)pb.out_t[l] | (uintptr_t)pb.out_w[l]) & 3) == 0 && ((uintptr_t)pb.w[l] & 15) == 0,
                     "sp_sru_pack_wt: outputs must be 4-byte aligned, weights 16-byte (layer %d)", q);
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(n_in / R, d / 64, 2 * L);
#define SP_PACKT(T, K) hipLaunchKernelGGL((sru_pack_wt_kernel<T, K>), grid, dim3(256), 0, s, pb, n_in, d, C, ldo)
    if (dtype == SP_F32) { if (k == 3) SP_PACKT(float, 3); else SP_PACKT(float, 4); }
    else if (dtype == SP_F16) { if (k == 3) SP_PACKT(__half, 3); else SP_PACKT(__half, 4); }
    else { if (k == 3) SP_PACKT(__hip_bfloat16, 3); else SP_PACKT(__hip_bfloat16, 4); }
#undef SP_PACKT
    return check_launch("sru_pack_wt_kernel");
}

extern "C" int sp_sru_pack_wt(const float* weight, int n_in, int d, int k, int C, int dtype, void* out_t,
                              int64_t ldo, void* out_w, void* stream) {
    return sp_sru_pack_wt_batched(&weight, 1, n_in, d, k, C, dtype, &out_t, ldo, out_w ? &out_w : nullptr, stream);
}

extern "C" int sp_sru_unpack_dw(const void* partials, int S, int n_in, int d, int k, int C, int dtype,
                                int accumulate, float* dweight, void* stream) {
    using namespace sp;
    int rc = pack_check("sp_sru_unpack_dw", n_in, d, k, C);
    if (rc) return rc;
    SP_CHECK_ARG(partials && dweight && S >= 1, "sp_sru_unpack_dw: null pointer or S=%d", S);
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16, "sp_sru_unpack_dw: dtype %d", dtype);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)(((int64_t)n_in * 2 * d + 255) / 256);
#define SP_UNPACK(T, K) hipLaunchKernelGGL((sru_unpack_dw_kernel<T, K>), dim3(blocks), dim3(256), 0, s, (const T*)partials, S, n_in, d, C, accumulate, dweight)
    if (dtype == SP_F32) { if (k == 3) SP_UNPACK(float, 3); else SP_UNPACK(float, 4); }
    else { if (k == 3) SP_UNPACK(__hip_bfloat16, 3); else SP_UNPACK(__hip_bfloat16, 4); }
#undef SP_UNPACK
    return check_launch("sru_unpack_dw_kernel");
}
