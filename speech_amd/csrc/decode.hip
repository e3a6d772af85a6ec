// Greedy CTC decode: argmax over V + blank/repeat collapse + in-order compaction.
// Replaces decoder.py:28-57 of the reference (torch.argmax on the CPU + a Python double loop).
//
// One 256-thread workgroup per utterance walks its T frames in chunks of 256.  Per chunk:
//   small V (rows fit 60 KB): the chunk's rows are copied to LDS with 16-byte coalesced loads and each
//                    thread takes the argmax of one row (odd/short row strides in LDS);
//   large V:         one wave per row, lanes strided over V, 64-lane max-reduce.
// Then each thread decides keep/drop for its frame, a 64-bit ballot + popcount gives the
// in-wave rank and the wave totals give the cross-wave offset, so tokens land in frame order.
//
// HBM traffic: scores are read exactly once; ids/lengths written once (4*B*T + 4*B bytes).
#include "common.h"

namespace sp {

constexpr int kChunk = 256;
constexpr size_t kSmallLdsBudget = 60 * 1024;  // staging bytes allowed on the small-V path

// Order-preserving key: larger key <=> larger float; NaN is the maximum (torch.argmax);
// -0.0 and +0.0 compare equal.
__device__ __forceinline__ uint32_t order_key(float v) {
    if (v != v) return 0xFFFFFFFFu;
    if (v == 0.0f) v = 0.0f;  // -0 -> +0
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <typename T>
__global__ __launch_bounds__(kChunk) void greedy_decode_kernel(
    const T* __restrict__ scores, int Tn, int V, int blank, int small_v,
    const int32_t* __restrict__ frame_lengths,
    int32_t* __restrict__ ids, int32_t* __restrict__ lengths) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // carve: [0,1024) amax int32[256]; [1024, 1024+32) wave totals + carry; then staging
    int32_t* s_amax = reinterpret_cast<int32_t*>(smem);
    int32_t* s_wtot = reinterpret_cast<int32_t*>(smem + 1024);  // [4] totals, [4] = carry amax
    T* s_rows = reinterpret_cast<T*>(smem + 1024 + 64);

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int Tb = frame_lengths ? min(max(frame_lengths[b], 0), Tn) : Tn;
    constexpr int EPV = 16 / (int)sizeof(T);  // elements per 16-byte vector
    const int64_t total_elems = (int64_t)gridDim.x * Tn * V;

    int out_base = 0;       // tokens emitted so far (uniform across the block)
    if (tid == 0) s_wtot[4] = -1;
    __syncthreads();

    for (int t0 = 0; t0 < Tb; t0 += kChunk) {
        const int rows = min(kChunk, Tb - t0);
        int my_arg = 0;
        if (small_v) {
            // ---- stage rows [t0, t0+rows) into LDS with aligned 16-byte loads ----
            const int64_t g0 = ((int64_t)b * Tn + t0) * V;
            const int64_t a0 = g0 & ~(int64_t)(EPV - 1);
            const int shift = (int)(g0 - a0);
            const int nvec = (shift + rows * V + EPV - 1) / EPV;
            const uint4* src = reinterpret_cast<const uint4*>(scores + a0);
            uint4* dst = reinterpret_cast<uint4*>(s_rows);
            for (int i = tid; i < nvec; i += kChunk) {
                if (a0 + (int64_t)(i + 1) * EPV <= total_elems) {
                    dst[i] = src[i];
                } else {  // last vector of the whole tensor: stay in bounds
                    for (int e = 0; e < EPV; ++e) {
                        int64_t gi = a0 + (int64_t)i * EPV + e;
                        if (gi < total_elems) s_rows[(int64_t)i * EPV + e] = scores[gi];
                    }
                }
            }
            __syncthreads();
            if (tid < rows) {
                const T* row = s_rows + shift + tid * V;
                uint32_t best = order_key(Elem<T>::ld(row));
                int arg = 0;
                for (int c = 1; c < V; ++c) {
                    uint32_t k = order_key(Elem<T>::ld(row + c));
                    if (k > best) { best = k; arg = c; }
                }
                my_arg = arg;
            }
        } else {
            // ---- one wave per row, lanes strided over V ----
            for (int r = wave; r < rows; r += kChunk / kWave) {
                const T* row = scores + ((int64_t)b * Tn + t0 + r) * V;
                unsigned long long best = 0ull;
                for (int c = lane; c < V; c += kWave) {
                    unsigned long long k = ((unsigned long long)order_key(Elem<T>::ld(row + c)) << 32) |
                                           (unsigned long long)(0xFFFFFFFFu - (uint32_t)c);
                    best = k > best ? k : best;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    unsigned long long other = __shfl_xor(best, o);
                    best = other > best ? other : best;
                }
                if (lane == 0) s_amax[r] = (int)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull));
            }
            __syncthreads();
            if (tid < rows) my_arg = s_amax[tid];
            __syncthreads();
        }
        // ---- collapse: need the previous frame's argmax ----
        if (tid < rows) s_amax[tid] = my_arg;
        __syncthreads();
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
        if (keep) ids[(int64_t)b * Tn + off + rank] = my_arg;
        const int chunk_total = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
        out_base += chunk_total;
        __syncthreads();  // everyone has read s_wtot / s_amax before they are rewritten
        if (tid == rows - 1) s_wtot[4] = my_arg;
        __syncthreads();
    }
    // ---- tail fill and length ----
    for (int i = out_base + tid; i < Tn; i += kChunk) ids[(int64_t)b * Tn + i] = -1;
    if (tid == 0) lengths[b] = out_base;
}

template <typename T>
static int launch_decode(const void* scores, int B, int Tn, int V, int blank, const int32_t* fl,
                         int32_t* ids, int32_t* lengths, hipStream_t s) {
    size_t lds = 1024 + 64;
    const size_t stage = align_up((size_t)kChunk * V * sizeof(T) + 32, 16);
    const int small_v = stage <= kSmallLdsBudget;
    if (small_v) lds += stage;
    hipLaunchKernelGGL(greedy_decode_kernel<T>, dim3(B), dim3(kChunk), lds, s,
                       reinterpret_cast<const T*>(scores), Tn, V, blank, small_v, fl, ids, lengths);
    return check_launch("greedy_decode_kernel");
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
