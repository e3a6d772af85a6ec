// Greedy CTC decode: argmax over V + blank/repeat collapse + in-order compaction.
// Replaces decoder.py:28-57 of the reference (torch.argmax on the CPU + a Python double loop).
//
// Two launches:
//   argmax_rows_kernel   grid (frame tiles, B).  A row is scanned by a group of G lanes (G = 8..64,
//                        chosen on the host from the row length) with 16-byte loads when rows are
//                        16-byte aligned, so one wave covers 64/G rows per pass and every wave of the
//                        chip streams scores: this is the HBM-bound part (scores read exactly once).
//                        The raw argmax lands in ids[b][t].
//   collapse_kernel      one workgroup per utterance compacts ids[b][:] IN PLACE: keep/drop per frame,
//                        64-bit ballot + popcount for the in-wave rank, wave totals for the cross-wave
//                        offset, so tokens land in frame order; tail filled with -1.  4*B*T bytes.
#include "common.h"

namespace sp {

constexpr int kChunk = 256;

// Order-preserving key: larger key <=> larger float; NaN is the maximum (torch.argmax);
// -0.0 and +0.0 compare equal.
__device__ __forceinline__ uint32_t order_key(float v) {
    if (v != v) return 0xFFFFFFFFu;
    if (v == 0.0f) v = 0.0f;  // -0 -> +0
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static void ld(const float* p, float (&o)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};
template <> struct Vec16<__half> {
    static constexpr int N = 8;
    __device__ static void ld(const __half* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const __half2* h = reinterpret_cast<const __half2*>(&v);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float2 f = __half22float2(h[i]); o[2 * i] = f.x; o[2 * i + 1] = f.y; }
    }
};
template <> struct Vec16<__hip_bfloat16> {
    static constexpr int N = 8;
    __device__ static void ld(const __hip_bfloat16* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
};

// G lanes per row, NP passes per wave; VEC: rows are 16-byte aligned and V % (16/sizeof T) == 0.
template <typename T, int G, bool VEC>
__global__ __launch_bounds__(kChunk) void argmax_rows_kernel(
    const T* __restrict__ scores, int Tn, int V, int NP, const int32_t* __restrict__ frame_lengths,
    int32_t* __restrict__ ids) {
    constexpr int RPW = kWave / G;               // rows per wave pass
    constexpr int E = VEC ? Vec16<T>::N : 1;     // elements per load
    constexpr int UN = 4;                        // loads in flight per lane
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / G, l = lane % G;
    const int Tb = frame_lengths ? min(max(frame_lengths[b], 0), Tn) : Tn;
    const int t_base = blockIdx.x * (kChunk / kWave) * RPW * NP;
    for (int p = 0; p < NP; ++p) {
        const int t = t_base + (p * (kChunk / kWave) + wave) * RPW + grp;
        const bool live = t < Tb;                // uniform within a lane group
        const T* row = scores + ((int64_t)b * Tn + (live ? t : 0)) * V;
        uint32_t best = 0u;
        int arg = 0x7FFFFFFF;
        if (live) {
            for (int c0 = l * E; c0 < V; c0 += UN * G * E) {
                float v[UN][E];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int c = c0 + u * G * E;
                    if (c < V) {
                        if constexpr (VEC) Vec16<T>::ld(row + c, v[u]);
                        else v[u][0] = Elem<T>::ld(row + c);
                    }
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int c = c0 + u * G * E;
                    if (c < V) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const uint32_t k = order_key(v[u][e]);
                            // strictly greater keeps the first maximum (columns ascend per lane);
                            // the very first element must win over the initial best = 0 key
                            if (k > best || arg == 0x7FFFFFFF) { best = k; arg = c + e; }
                        }
                    }
                }
            }
        }
        // group reduce: larger key wins, ties -> smaller column
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
            const uint32_t ob = (uint32_t)__shfl_xor((int)best, o);
            const int oa = __shfl_xor(arg, o);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        if (live && l == 0) ids[(int64_t)b * Tn + t] = arg;
    }
}

__global__ __launch_bounds__(kChunk) void collapse_kernel(
    int Tn, int blank, const int32_t* __restrict__ frame_lengths, int32_t* __restrict__ ids,
    int32_t* __restrict__ lengths) {
    __shared__ int32_t s_amax[kChunk];
    __shared__ int32_t s_wtot[8];               // [0..3] wave totals, [4] previous chunk's last argmax
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int Tb = frame_lengths ? min(max(frame_lengths[b], 0), Tn) : Tn;
    int32_t* row = ids + (int64_t)b * Tn;
    int out_base = 0;                            // tokens emitted so far (uniform across the block)
    if (tid == 0) s_wtot[4] = -1;
    __syncthreads();
    for (int t0 = 0; t0 < Tb; t0 += kChunk) {
        const int rows = min(kChunk, Tb - t0);
        const int my_arg = tid < rows ? row[t0 + tid] : 0;
        if (tid < rows) s_amax[tid] = my_arg;
        __syncthreads();                         // every raw argmax of this chunk is read before any write below
        bool keep = false;
        if (tid < rows) {
            const int prev = (tid == 0) ? s_wtot[4] : s_amax[tid - 1];
            const bool first = (t0 + tid) == 0;
            keep = (my_arg != blank) && (first || my_arg != prev);
        }
        const unsigned long long m = __ballot(keep);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wtot[wave] = __popcll(m);
        __syncthreads();
        int off = out_base;
        for (int w = 0; w < wave; ++w) off += s_wtot[w];
        if (keep) row[off + rank] = my_arg;      // off + rank <= t0 + tid: never ahead of the read front
        out_base += s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
        __syncthreads();
        if (tid == rows - 1) s_wtot[4] = my_arg;
        __syncthreads();
    }
    for (int i = out_base + tid; i < Tn; i += kChunk) row[i] = -1;
    if (tid == 0) lengths[b] = out_base;
}

template <typename T, int G>
static void launch_argmax(const T* scores, int B, int Tn, int V, bool vec, const int32_t* fl, int32_t* ids,
                          hipStream_t s) {
    const int rpw = kWave / G;
    const int NP = G == 64 ? 4 : 2;
    const int rows_per_wg = (kChunk / kWave) * rpw * NP;
    dim3 grid((Tn + rows_per_wg - 1) / rows_per_wg, B);
    if (vec) hipLaunchKernelGGL((argmax_rows_kernel<T, G, true>), grid, dim3(kChunk), 0, s, scores, Tn, V, NP, fl, ids);
    else hipLaunchKernelGGL((argmax_rows_kernel<T, G, false>), grid, dim3(kChunk), 0, s, scores, Tn, V, NP, fl, ids);
}

template <typename T>
static int launch_decode(const void* scores_v, int B, int Tn, int V, int blank, const int32_t* fl,
                         int32_t* ids, int32_t* lengths, hipStream_t s) {
    const T* scores = reinterpret_cast<const T*>(scores_v);
    constexpr int EPV = 16 / (int)sizeof(T);
    const bool vec = (V % EPV) == 0;
    const int nload = vec ? V / EPV : V;         // loads per row
    if (Tn > 0) {
        if (nload >= 128) launch_argmax<T, 64>(scores, B, Tn, V, vec, fl, ids, s);
        else if (nload >= 64) launch_argmax<T, 32>(scores, B, Tn, V, vec, fl, ids, s);
        else if (nload >= 32) launch_argmax<T, 16>(scores, B, Tn, V, vec, fl, ids, s);
        else launch_argmax<T, 8>(scores, B, Tn, V, vec, fl, ids, s);
        int rc = check_launch("argmax_rows_kernel");
        if (rc != SP_OK) return rc;
    }
    hipLaunchKernelGGL(collapse_kernel, dim3(B), dim3(kChunk), 0, s, Tn, blank, fl, ids, lengths);
    return check_launch("collapse_kernel");
}

}  // namespace sp

extern "C" int sp_greedy_decode(const void* scores, int dtype, int B, int T, int V, int blank,
                                const int32_t* frame_lengths, int32_t* ids, int32_t* lengths,
                                void* stream) {
    using namespace sp;
    SP_CHECK_ARG(B >= 0 && T >= 0 && V >= 1, "sp_greedy_decode: bad shape B=%d T=%d V=%d", B, T, V);
    SP_CHECK_ARG(blank >= 0 && blank < V, "sp_greedy_decode: blank %d outside [0,%d)", blank, V);
    if (B == 0) return SP_OK;
    SP_CHECK_ARG(scores && ids && lengths, "sp_greedy_decode: null pointer");
    SP_CHECK_ARG(((uintptr_t)scores & 15) == 0, "sp_greedy_decode: scores must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (dtype) {
        case SP_F32: return launch_decode<float>(scores, B, T, V, blank, frame_lengths, ids, lengths, s);
        case SP_F16: return launch_decode<__half>(scores, B, T, V, blank, frame_lengths, ids, lengths, s);
        case SP_BF16: return launch_decode<__hip_bfloat16>(scores, B, T, V, blank, frame_lengths, ids, lengths, s);
        default: return fail(SP_EINVAL, "sp_greedy_decode: unknown dtype %d", dtype);
    }
}
