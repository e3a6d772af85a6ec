// Fused log-softmax + CTC loss + gradient for gfx950.
// Replaces train.py:69-77 of the reference (F.log_softmax -> transpose -> torch.nn.CTCLoss).
//
// Three launches on the caller's stream (v_exp_f32/v_log_f32 are base-2 natively, so log-probabilities
// are kept in log2 units):
//   K1 ctc_rows    (whole chip, one wave per (t,b) row): lse2 = log2 sum_c 2^(z_c*log2e) and the
//                  compact per-frame row the lattice needs -- either the normalised row itself
//                  (small vocabulary, V <= Lmax+1: indexed by class) or the gathered
//                  (blank, l_1..l_L) log-probs (large vocabulary: indexed by label position).
//   K2 ctc_lattice (2 workgroups per utterance: alpha forward in time, beta backward in time,
//                  running concurrently).  The recursion is serial in time, so the kernel is
//                  latency-bound and everything is done to keep the per-frame dependency chain
//                  short: states are PROBABILITIES in f64 registers scaled by an exact power of two
//                  (new = (a[s] + a[s-1] + skip*a[s-2]) * y: DPP move, add, fma, mul -- no exp/log
//                  on the chain); the emission y = 2^(log2-prob) is built off the chain from an f32
//                  v_exp_f32 of the fraction and an f64 ldexp of the integer part, so rows as
//                  improbable as 2^-1000 stay exact; every 8th frame the wave maximum's exponent
//                  (measured 7 frames earlier, off the chain) is folded into that ldexp and added
//                  to the row's integer scale E_t.  Each thread owns NS consecutive lattice states;
//                  neighbours come from the adjacent lane by DPP wave shift, and only the
//                  wave-boundary state crosses LDS (one barrier per frame, none for a single-wave
//                  lattice).  The per-frame rows are staged through LDS a chunk of
//                  frames ahead (issue loads early, write to LDS late).  beta is the same
//                  recursion on the reversed label sequence over reversed time, so there is one
//                  code path.  Lattice rows are spilled to the workspace in coalesced rows.
//   K3 gradient.   Vocabularies of <= 64 classes with class-indexed rows (the reference's 28 characters + blank), lattices
//                  up to 1024 states: ctc_char_grad writes whole gradient rows after the lattice (p - occupancy for
//                  every class, 32-bit posterior arithmetic on the packed lattice words, no dense half, the logits
//                  are not read again).  Otherwise single-wave lattices (S <= 512) split it in two.  The DENSE half, softmax(z) * scale over
//                  the whole vocabulary, needs K1's log-sum-exp only: it rides in K2's launch, in workgroups
//                  behind the 2B lattice workgroups on the CUs the lattice leaves idle (one workgroup per CU,
//                  so no streaming wave shares a lattice CU).  The SPARSE half (ctc_sparse) runs after it:
//                  posterior per state 2^(alpha+beta-lp+nll), summed per class in compact LDS slots, and
//                  the <= L + 1 touched classes of a frame are overwritten with (p - occ) * scale.
//                  Larger lattices keep the fused ctc_grad (whole chip, one wave per (t,b) row:
//                  occupancies per class in LDS, grad = (softmax - occ) * scale).
//
// Workspace: lse2[B*T] | rows[B][T][RW] (f32), then alpha[B][T][SP] | beta_rev[B][T][SP] | nll2[B] (f64)
// | aux.  True alpha_t(s) = alpha[b][t][s] * 2^E_t.  Single-wave lattices (S <= 1024: 2, 4, 8 or 16 states per lane,
// SP = 64 x states per lane) spill PACKED rows, four bytes a state (lat_pack / lat_unpack, spill_col order) and write
// E once per 8-frame block into their aux block; the multi-wave kernel (S > 1024) spills f64 rows of 2*Lmax+1 rounded
// up to 8 > S slots and keeps E_t in slot SP-1 of the row.
//
// Precision: log alpha/beta reach thousands of log2 units, where one f32 ulp is ~2e-4: a plain f32
// log-space recursion (torch's included) random-walks to ~3e-4 relative error in the posteriors.
// Scaled f64 probabilities carry 53 bits through every add; the only f32 roundings are the
// emission exp2 (1 ulp) and K1's log-sum-exp.  The spilled lattice keeps f64's EXPONENT (for tightly
// constrained alignments the states that carry the posterior sit hundreds of binary orders below
// the row maximum: f64 reaches 2^-1022 below it; anything smaller is a posterior of zero) and, packed,
// 21 bits of mantissa (2.4e-7 relative per factor, unbiased; the likelihood itself is summed from the
// f64 registers).  K3 multiplies the mantissas and applies 2^(e_a + e_b + E_a + E_b - log2 y - log2 P) by ldexp.
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "common.h"

namespace sp {

// Scratch of sp_ctc_fwd_bwd_noalloc (the entry point for callers that cannot allocate): one copy per GPU,
// held from the moment the code object is loaded on that GPU until the process ends.  In host code the
// array's NAME is only a handle for hipGetSymbolAddress; every use of its memory goes through the
// per-device pointer obtained below.
constexpr size_t kCtcScratchBytes = (size_t)64 << 20;
static __device__ __align__(256) unsigned char scratch[kCtcScratchBytes];
constexpr int kMaxDevices = 64;
static std::atomic<void*> g_scratch_ptr[kMaxDevices];

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kMaxV = 4000;  // K3 keeps 4 x V floats (fused) or ints + a few label-sized rows (sparse half) in LDS: <= 72 KB
constexpr int kMaxL = 1023;
constexpr int kPF = 8;  // float4 prefetch registers per thread in K2
constexpr int kRescale = 8;  // K2: frames per rescale block (the exponent measured at its start is applied in its last frame)
// The rescale parks the row maximum at 2^kTopBits, not at 1: what the recursion needs is room BELOW the maximum (states
// that carry the posterior can sit hundreds of binary orders under it, and a block's emissions shrink everything for up
// to 2 * kRescale frames before the next correction lands), while above it a row can only grow 3x per frame
// (3^16 = 2^26).  f64 then offers ~1900 binary orders below the maximum instead of 1022.
constexpr int kTopBits = 900;
constexpr int kCUs = 256;        // MI355X
constexpr int kAuxDummy = 512;   // doubles of the single-wave kernel's dummy spill row (64 lanes x up to 8 states)
constexpr int kLat1Threads = 512;      // workgroup of ctc_lattice1_kernel: with kLat1DynLds one workgroup owns a CU
constexpr int kLat1DynLds = 20 * 1024;  // unused dynamic LDS on top of the 64 KB of emissions: a second workgroup does not fit
constexpr int kSingleWaveMaxS = 1024;  // lattices up to this many states run on ctc_lattice1_kernel
constexpr bool kPackAll = true;        // every single-wave lattice spills packed rows (lat_pack), not only eight / sixteen states per lane
constexpr int kSplitMaxS = 512;        // ... and up to this many have a sparse-half kernel (dense half beside the lattice)

// Lattices of eight and sixteen states per lane (256 < S <= 1024) are spilled PACKED: exponent and the upper 21 bits
// of the mantissa of each f64 state (states are never negative: the sign bit's place goes to the mantissa), four bytes
// a state, truncated by the producer and read back as the middle of the truncation interval.  At these sizes the
// spill and its read-back are what the lattice and the gradient kernel wait for (30 s of characters, 64 utterances:
// 1.4 GB each way in f64); the posterior keeps 2^-21 relative precision, two hundred times inside the 1e-4 bound,
// and the likelihood itself is taken from the registers in full f64.
// Column of lattice state p in a packed row: a lane's states in quads, the quads 1 KB apart ([quad][lane][4]: each
// store of the recursion wave is one contiguous KB; with a lane's bytes side by side in the row every store touched
// all of the row's sectors a fraction each and the recursion paid 35 us of 141 for it); 64 consecutive states are
// then whole 64-byte pieces of NS / 4 planes, so the consumers' reads stay coalesced.
__device__ __forceinline__ int spill_col(int p, int perm) {   // perm = log2(states per lane) of the packed producer, or 0
    // (a producer of several waves would put wave w's block of 64 << perm states behind wave w - 1's)
    const int q = p & ((64 << perm) - 1);
    return perm > 2 ? (p - q) + ((q & ((1 << perm) - 4)) << 6) + ((q >> perm) << 2) + (q & 3) : p;   // up to four states per lane: state order
}
__device__ __forceinline__ uint32_t lat_pack(double x) {   // one instruction on the recursion wave (v_alignbit_b32)
    return __builtin_amdgcn_alignbit((uint32_t)__double2hiint(x), (uint32_t)__double2loint(x), 31);
}
__device__ __forceinline__ double lat_unpack(uint32_t w) {   // the middle of the interval the truncation left; 0 stays 0
    return __hiloint2double((int)(w >> 1), w ? (int)((w << 31) | 0x40000000u) : 0);
}
// state (row, col) of a lattice block: packed words (perm != 0) or doubles
__device__ __forceinline__ double lat_load(const void* base, int64_t row, int SP, int col, int perm) {
    if (perm) return lat_unpack(reinterpret_cast<const uint32_t*>(base)[row * SP + col]);
    return reinterpret_cast<const double*>(base)[row * SP + col];
}

struct CtcDims {
    int T, B, V, Lmax;
    int RW;    // row width in floats (multiple of 4)
    int SP;    // padded lattice row (multiple of 8)
    int perm;  // != 0: PACKED rows in the store order of the kernel with 2^perm states per lane (spill_col); 0: f64, in state order
    int modeB; // 1: rows indexed by class, 0: by label position
    int small, split_ok;   // route (see ctc_dims)
    size_t off_lse, off_rows, total_floats;  // float offsets
    size_t off_dbl;  // float offset of alpha[B][T][SP] (f64 or packed words), 16-byte aligned like the next two
    size_t off_beta, off_nll2;   // beta_rev[B][T][SP], nll2[B] (f64)
    size_t off_aux;  // float offset of the aux f64 block (see ctc_dims)
    size_t total_bytes;
};

static CtcDims ctc_dims(int T, int B, int V, int Lmax) {
    CtcDims d;
    d.T = T; d.B = B; d.V = V; d.Lmax = Lmax;
    d.modeB = V <= Lmax + 1;
    d.RW = (int)align_up((size_t)(d.modeB ? V : Lmax + 1), 4);
    const int S = 2 * Lmax + 1;
    // 256 < S <= 1024: packed rows of the single-wave kernel with eight / sixteen states per lane
    d.small = d.modeB && V <= 64;
    d.split_ok = !d.small && S <= kSplitMaxS && 4 * B <= kCUs;
    d.perm = S > kSingleWaveMaxS ? 0 : S <= 256 ? (kPackAll ? (S <= 128 ? 1 : 2) : 0) : S <= 512 ? 3 : 4;
    // permuted rows and the lattices of two states per lane: every lane has its columns (pitch 64 x states per lane),
    // so the recursion wave's row pointer is one scalar walked by a scalar stride
    d.SP = d.perm ? 64 << d.perm : S <= 128 ? 128 : (int)align_up((size_t)S, 8);
    size_t o = 0;
    d.off_lse = o; o += align_up((size_t)B * T, 4);
    d.off_rows = o; o += (size_t)B * T * d.RW;
    o = align_up(o, 4);
    const size_t lat = align_up((size_t)B * T * d.SP * (d.perm ? 1 : 2), 4);   // one lattice block, in floats
    d.off_dbl = o; o += lat;
    d.off_beta = o; o += lat;
    d.off_nll2 = o; o += align_up((size_t)2 * B, 4);
    d.off_aux = o;                            // single-wave lattice kernel: per (dir, b) a dummy spill
    o += 2 * ((size_t)2 * B * (kAuxDummy + T / kRescale + 2));   // row (kAuxDummy doubles) and the scale E after each block
    d.total_floats = o;
    d.total_bytes = o * sizeof(float);
    return d;
}

// -------------------------------------------------------------------------------------------
// K1: per-row log-sum-exp and compact row
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ctc_rows_kernel(
    const float* __restrict__ logits, int64_t st, int64_t sb, const int64_t* __restrict__ labels,
    int label_stride, const int32_t* __restrict__ in_len, const int32_t* __restrict__ lab_len,
    int T, int B, int V, int blank, int Lmax, int RW, int modeB, float* __restrict__ lse2,
    float* __restrict__ rows) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int64_t nrows = (int64_t)T * B;
    for (int64_t r = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); r < nrows;
         r += (int64_t)gridDim.x * wpb) {
        // r enumerates (b, t) batch-major so a block's waves write neighbouring rows
        const int b = (int)(r / T), t = (int)(r % T);
        const int Tb = min(max(in_len[b], 0), T);
        if (t >= Tb) continue;
        const float* z = logits + t * st + b * sb;
        float m = -INFINITY;
        for (int c = lane; c < V; c += 64) m = fmaxf(m, z[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < V; c += 64) s += exp2f((z[c] - m) * kLog2e);
        s = wave_sum(s);
        const float l2 = m * kLog2e + log2f(s);
        if (lane == 0) lse2[(int64_t)b * T + t] = l2;
        float* out = rows + ((int64_t)b * T + t) * RW;
        if (modeB) {
            for (int c = lane; c < V; c += 64) out[c] = z[c] * kLog2e - l2;
        } else {
            const int L = min(max(lab_len[b], 0), Lmax);
            for (int i = lane; i <= L; i += 64) {
                int idx = blank;
                if (i > 0) idx = (int)labels[(int64_t)b * label_stride + i - 1];
                idx = min(max(idx, 0), V - 1);
                out[i] = z[idx] * kLog2e - l2;
            }
        }
    }
}

// K1, small vocabularies (V <= HALF, HALF = 32 or 64: the reference's 29 characters, text.py:1-12): a row is at most
// 256 bytes, so one wave per row spends a memory round trip and two wave reductions on 29 numbers (35 us for 64 x 1501
// rows).  Here a wave takes 64 / HALF rows side by side (lane = class within its row, reductions within HALF lanes) and
// kIt of such groups at once: every load of the 4 (8) rows is in flight before the first reduction starts.
template <int HALF>
__global__ __launch_bounds__(256) void ctc_rows_small_kernel(
    const float* __restrict__ logits, int64_t st, int64_t sb, const int64_t* __restrict__ labels,
    int label_stride, const int32_t* __restrict__ in_len, const int32_t* __restrict__ lab_len,
    int T, int B, int V, int blank, int Lmax, int RW, int modeB, float* __restrict__ lse2,
    float* __restrict__ rows) {
    constexpr int PER = 64 / HALF, kIt = 4, NR = PER * kIt;
    const int lane = threadIdx.x & 63, sub = lane / HALF, c = lane % HALF;
    const int wpb = blockDim.x >> 6;
    const int64_t nrows = (int64_t)T * B;
    auto red_max = [](float v) {
#pragma unroll
        for (int o = HALF / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        return v;
    };
    auto red_sum = [](float v) {
#pragma unroll
        for (int o = HALF / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    for (int64_t r0 = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * NR; r0 < nrows; r0 += (int64_t)gridDim.x * wpb * NR) {
        float z[kIt];
        int bb[kIt], tt[kIt];
        bool live[kIt];
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const int64_t r = r0 + i * PER + sub;      // (b, t) batch-major, as the other row kernels
            const int64_t rc = min(r, nrows - 1);
            bb[i] = (int)(rc / T);
            tt[i] = (int)(rc % T);
            live[i] = r < nrows && tt[i] < min(max(in_len[bb[i]], 0), T);
            z[i] = (live[i] && c < V) ? logits[tt[i] * st + bb[i] * sb + c] : -INFINITY;
        }
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const float m = red_max(z[i]);
            const float e = (c < V) ? exp2f((z[i] - m) * kLog2e) : 0.f;
            const float sum = red_sum(live[i] ? e : 0.f);
            const float l2 = m * kLog2e + log2f(sum);
            if (!live[i]) continue;       // uniform within the HALF lanes of a row
            const int64_t row = (int64_t)bb[i] * T + tt[i];
            if (c == 0) lse2[row] = l2;
            float* out = rows + row * RW;
            if (modeB) {
                if (c < V) out[c] = z[i] * kLog2e - l2;
            } else {
                // by label position: lane c fetches class idx's score from the lane that holds it
                const int L = min(max(lab_len[bb[i]], 0), Lmax);
                for (int p0 = 0; p0 <= L; p0 += HALF) {
                    const int pos = p0 + c;
                    int idx = blank;
                    if (pos > 0 && pos <= L) idx = (int)labels[(int64_t)bb[i] * label_stride + pos - 1];
                    idx = min(max(idx, 0), V - 1);
                    const float zi = __shfl(z[i], sub * HALF + idx);
                    if (pos <= L) out[pos] = zi * kLog2e - l2;
                }
            }
        }
    }
}

// K1, vector path (V % 4 == 0, 16-byte aligned rows, V <= 1024 * NV / 4): the row is read ONCE with
// float4 loads and stays in registers for the max, the sum and (mode B) the normalised row.
template <int NV>
__global__ __launch_bounds__(256) void ctc_rows_vec_kernel(
    const float* __restrict__ logits, int64_t st, int64_t sb, const int64_t* __restrict__ labels,
    int label_stride, const int32_t* __restrict__ in_len, const int32_t* __restrict__ lab_len,
    int T, int B, int V, int blank, int Lmax, int RW, int modeB, float* __restrict__ lse2,
    float* __restrict__ rows) {
    __shared__ __attribute__((aligned(16))) float srow[4][NV * 256];   // one row per wave (launched with 256 threads)
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int V4 = V >> 2;
    const int64_t nrows = (int64_t)T * B;
    for (int64_t r = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); r < nrows;
         r += (int64_t)gridDim.x * wpb) {
        const int b = (int)(r / T), t = (int)(r % T);
        const int Tb = min(max(in_len[b], 0), T);
        if (t >= Tb) continue;
        const float* z = logits + t * st + b * sb;
        const float4* z4 = reinterpret_cast<const float4*>(z);
        float4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = lane + 64 * i;
            v[i] = q < V4 ? z4[q] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
        m = wave_max(m);
        const float mk = m * kLog2e;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            sum += exp2f(fmaf(v[i].x, kLog2e, -mk)) + exp2f(fmaf(v[i].y, kLog2e, -mk)) +
                   exp2f(fmaf(v[i].z, kLog2e, -mk)) + exp2f(fmaf(v[i].w, kLog2e, -mk));
        sum = wave_sum(sum);
        const float l2 = mk + log2f(sum);
        if (lane == 0) lse2[(int64_t)b * T + t] = l2;
        float* out = rows + ((int64_t)b * T + t) * RW;
        if (modeB) {
            float4* out4 = reinterpret_cast<float4*>(out);   // RW == V here (V % 4 == 0)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = lane + 64 * i;
                if (q < V4)
                    out4[q] = make_float4(fmaf(v[i].x, kLog2e, -l2), fmaf(v[i].y, kLog2e, -l2),
                                          fmaf(v[i].z, kLog2e, -l2), fmaf(v[i].w, kLog2e, -l2));
            }
        } else {
            // gather the label positions from the row the wave already holds: registers -> the wave's LDS row ->
            // indexed reads (re-reading z[idx] from memory fetched 1.4x the logits: PMC, round 1)
            float4* s4 = reinterpret_cast<float4*>(srow[threadIdx.x >> 6]);
#pragma unroll
            for (int i = 0; i < NV; ++i) s4[lane + 64 * i] = v[i];
            __builtin_amdgcn_wave_barrier();
            const int L = min(max(lab_len[b], 0), Lmax);
            for (int i = lane; i <= L; i += 64) {
                int idx = blank;
                if (i > 0) idx = (int)labels[(int64_t)b * label_stride + i - 1];
                idx = min(max(idx, 0), V - 1);
                out[i] = fmaf(srow[threadIdx.x >> 6][idx], kLog2e, -l2);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// -------------------------------------------------------------------------------------------
// K2: alpha / beta lattice
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_shr1_i(int v) {
    // lane i receives lane i-1's value; lane 0 receives 0 (bound_ctrl: no source lane -> 0)
    return __builtin_amdgcn_update_dpp(0, v, 0x138 /*wave_shr:1*/, 0xF, 0xF, true);
}
__device__ __forceinline__ double lane_shr1(double v) {  // lane 0 receives +0.0
    return __hiloint2double(lane_shr1_i(__double2hiint(v)), lane_shr1_i(__double2loint(v)));
}
// wave-wide max of a non-negative int on the VALU only (DPP within each row of 16 lanes, then four
// v_readlane): no LDS round trips, so it can sit beside the recursion without stalling it
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141 /*row_half_mirror*/, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140 /*row_mirror*/, 0xF, 0xF, false));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// wave-wide f32 sum on the VALU only, same association as common.h's butterfly wave_sum (same bits)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});
    v += dpp(v, std::integral_constant<int, 0x4E>{});
    v += dpp(v, std::integral_constant<int, 0x141>{});
    v += dpp(v, std::integral_constant<int, 0x140>{});
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}
// 2^(x - shift) as f64 for x <= 0 (a log2-probability; -inf allowed) and integer shift: f32 exp2 of
// the fraction (in [1,2), 1 ulp), exact f64 ldexp of the integer part.
__device__ __forceinline__ double exp2_wide(float x, int shift) {
    x = fmaxf(x, -1200.f);
    const float n = floorf(x);
    const double m = (double)__builtin_amdgcn_exp2f(x - n);
    return ldexp(m, (int)n - shift);
}
// log2 of a non-negative f64 of any magnitude (0 -> -inf)
__device__ __forceinline__ double log2_wide(double p) {
    if (!(p > 0.0)) return p == 0.0 ? -(double)INFINITY : p;
    int ex;
    const double mant = frexp(p, &ex);
    return (double)ex + (double)log2f((float)mant);
}


template <int W, int NS>
__global__ __launch_bounds__(64 * W) void ctc_lattice_kernel(
    const float* __restrict__ rows, const int64_t* __restrict__ labels, int label_stride,
    const int32_t* __restrict__ in_len, const int32_t* __restrict__ lab_len, int T, int V,
    int blank, int Lmax, int RW, int SP, int modeB, int TB, double* __restrict__ alpha,
    double* __restrict__ beta_rev, double* __restrict__ nll2, float* __restrict__ nll) {
    constexpr int NT = 64 * W;
    // LDS: two row-chunk buffers of kPF*NT float4 each, then boundary slots, then final pair
    __shared__ __attribute__((aligned(16))) float4 s_chunk[2][kPF * NT];
    __shared__ double s_bnd[2][W];   // per wave: its last state (previous frame)
    __shared__ int s_emax[2][W];
    __shared__ int s_rep[NT];
    __shared__ double s_fin[2];

    const int b = blockIdx.x, dir = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Tb = min(max(in_len[b], 0), T);
    const int L = min(max(lab_len[b], 0), Lmax);
    const int S = 2 * L + 1;
    const int64_t* lab = labels + (int64_t)b * label_stride;
    double* out = (dir ? beta_rev : alpha) + (int64_t)b * T * SP;
    const float* rows_b = rows + (int64_t)b * T * RW;

    if (Tb == 0) {
        if (dir == 0 && tid == 0) {
            nll[b] = (L == 0) ? 0.f : INFINITY;
            nll2[b] = (L == 0) ? 0.0 : (double)INFINITY;
        }
        return;
    }

    // ---- per-state constants ----
    // lab'[k]: the label sequence in this recursion's order (reversed for beta)
    auto labp = [&](int k) { return min(max((int)lab[dir ? L - 1 - k : k], 0), V - 1); };
    int ridx[NS];
    int need[NS];    // frames still required after this one to reach the end from this position
    double skm[NS];  // 1.0 where the s-2 transition exists
    {
        // thread tid owns label indices [tid*NS/2, tid*NS/2 + NS/2) (positions 2k and 2k+1)
        constexpr int NK = NS / 2;
        int rflag[NK], cnt = 0;
#pragma unroll
        for (int q = 0; q < NK; ++q) {
            const int k = tid * NK + q;
            rflag[q] = (k >= 1 && k < L && labp(k) == labp(k - 1)) ? 1 : 0;   // mandatory blank before k
            cnt += rflag[q];
        }
        s_rep[tid] = cnt;
        __syncthreads();
        int after = 0;  // repeats among labels owned by higher threads
        for (int u = tid + 1; u < NT; ++u) after += s_rep[u];
#pragma unroll
        for (int q = NK - 1; q >= 0; --q) {
            const int k = tid * NK + q;
            // R(k) = #repeat pairs (j-1, j) with j > k
            const int Rk = after;
            const int need_label = (L - 1 - k) + Rk;            // position 2k+1
            need[2 * q + 1] = need_label;
            need[2 * q] = (k < L) ? need_label + 1 : 0;          // blank before label k; final blank: 0
            after += rflag[q];
        }
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int sp_ = tid * NS + j;           // position in this recursion's order
        const int s = dir ? S - 1 - sp_ : sp_;   // real lattice state
        ridx[j] = 0;
        skm[j] = 0.0;
        if (sp_ < S) {
            int cls = blank;
            if (s & 1) cls = min(max((int)lab[(s - 1) >> 1], 0), V - 1);
            ridx[j] = modeB ? cls : ((s & 1) ? (s + 1) >> 1 : 0);
            if ((s & 1) && sp_ >= 2) {
                const int s2 = dir ? s + 2 : s - 2;  // the state two steps earlier in this order
                const int cls2 = min(max((int)lab[(s2 - 1) >> 1], 0), V - 1);
                skm[j] = cls != cls2 ? 1.0 : 0.0;
            }
        } else {
            need[j] = 0x3FFFFFFF;  // never live
        }
    }

    const int nchunks = (Tb + TB - 1) / TB;
    const float4* src_base = reinterpret_cast<const float4*>(rows_b);
    const int rw4 = RW >> 2;

    float4 pf[kPF];
    auto chunk_lo = [&](int c) { return dir ? max(0, Tb - (c + 1) * TB) : c * TB; };
    auto chunk_hi = [&](int c) { return dir ? Tb - c * TB : min(Tb, (c + 1) * TB); };
#define SP_CTC_ISSUE(c_)                                                              \
    {                                                                                 \
        const int lo_ = chunk_lo(c_), n4_ = (chunk_hi(c_) - lo_) * rw4;               \
        const float4* src_ = src_base + (int64_t)lo_ * rw4;                           \
        _Pragma("unroll") for (int n = 0; n < kPF; ++n) {                             \
            const int q = tid + n * NT;                                               \
            pf[n] = (q < n4_) ? src_[q] : make_float4(0.f, 0.f, 0.f, 0.f);            \
        }                                                                             \
    }
#define SP_CTC_COMMIT(c_)                                                             \
    {                                                                                 \
        _Pragma("unroll") for (int n = 0; n < kPF; ++n) s_chunk[(c_) & 1][tid + n * NT] = pf[n]; \
    }

    SP_CTC_ISSUE(0);
    SP_CTC_COMMIT(0);
    __syncthreads();

    // a(-1) = delta(position 0): one step of the general recurrence then yields the CTC start
    // condition (positions 0 and 1 take their emission, everything else 0)
    double a[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) a[j] = (tid * NS + j == 0) ? 1.0 : 0.0;
    int E = 0;  // alpha_true = a * 2^E
    const bool spill_ok = tid * NS < SP;
    const bool e_slot = (tid + 1) * NS == SP;   // this lane's last slot is row[SP-1]
    if (W > 1) {
        if (tid < W) s_bnd[0][tid] = 0.0;   // a(-1) of every wave's last position
        __syncthreads();
    }

    int i = 0;
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) SP_CTC_ISSUE(c + 1);
        const float* buf = reinterpret_cast<const float*>(s_chunk[c & 1]);
        const int lo = chunk_lo(c), hi = chunk_hi(c);
        const int nsteps = hi - lo;
        // blocks of kRescale frames: emissions of the whole block first (LDS reads, exp2, ldexp --
        // all independent of the recursion), then the frames' dependency chains back to back
        for (int k0 = 0; k0 < nsteps; k0 += kRescale) {
            const int nb = min(kRescale, nsteps - k0);
            // scale: exponent of the current maximum, folded into the block's last frame
            int dp = 0;
            if (nb == kRescale) {
                double m = a[0];
#pragma unroll
                for (int j = 1; j < NS; ++j) m = fmax(m, a[j]);
                const int e = wave_max_i((__double2hiint(m) >> 20) & 0x7FF);
                if (W > 1) {
                    if (lane == 0) s_emax[(i / kRescale) & 1][wave] = e;   // read after >= 1 barrier
                } else {
                    dp = e ? e - (1023 + kTopBits) : 0;
                }
            }
            double y[kRescale][NS];
#pragma unroll
            for (int q = 0; q < kRescale; ++q) {
                const int iq = i + min(q, nb - 1);
                const int t = dir ? Tb - 1 - iq : iq;
                const float* row = buf + (t - lo) * RW;
                const int rem = Tb - 1 - iq;  // frames left: positions that need more are dead (exact
                                              // pruning: a dead position only ever feeds dead positions)
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const float lp = row[ridx[j]];
                    float x = fmaxf(lp, -1200.f);
                    const float n = floorf(x);
                    const double mnt = (double)__builtin_amdgcn_exp2f(x - n);
                    // the block's scale shift is folded in below for q == kRescale-1 (W == 1) or
                    // applied to y after the barrier-protected read (W > 1)
                    const int sh = (W == 1 && q == kRescale - 1) ? dp : 0;
                    const double v = ldexp(mnt, (int)n - sh);
                    y[q][j] = need[j] <= rem ? v : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < kRescale; ++q) {
                if (q < nb) {
                    const int t = dir ? Tb - 1 - i : i;
                    if (W > 1 && q == kRescale - 1) {
                        int e = 0;
#pragma unroll
                        for (int w = 0; w < W; ++w) e = max(e, s_emax[(i / kRescale) & 1][w]);
                        dp = e ? e - (1023 + kTopBits) : 0;
#pragma unroll
                        for (int j = 0; j < NS; ++j) y[q][j] = ldexp(y[q][j], -dp);
                    }
                    double p1 = lane_shr1(a[NS - 1]);
                    if (W > 1) {
                        if (lane == 0 && wave > 0) p1 = s_bnd[i & 1][wave - 1];
                    }
                    double n_[NS];
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        // NS is even: position parity == state parity; s-1 and s-2 of an odd (label)
                        // position j are a[j-1], a[j-2] (the previous lane's last state for j == 1)
                        const double x1 = (j >= 1) ? a[j - 1] : p1;
                        const double x2 = (j >= 2) ? a[j - 2] : p1;
                        const double base = (j >= 1) ? a[j] + x1 : a[j];   // j >= 1: off the DPP chain
                        n_[j] = (j >= 1) ? fma(skm[j], x2, base) * y[q][j] : (base + x1) * y[q][j];
                    }
#pragma unroll
                    for (int j = 0; j < NS; ++j) a[j] = n_[j];
                    if (q == kRescale - 1) E += dp;
                    // spill the lattice row (coalesced 16-byte stores) and its scale
                    // (one vector-memory op per lane and frame: the 64-deep vmem queue of the wave
                    // is what the write latency is divided by).  Slot SP-1 >= S is never a state: it
                    // carries the row's scale E.
                    if (spill_ok) {
                        double* orow = out + (int64_t)t * SP + tid * NS;
#pragma unroll
                        for (int j = 0; j < NS; j += 2) {
                            const double hi_ = (e_slot && j == NS - 2) ? (double)E : a[j + 1];
                            *reinterpret_cast<double2*>(orow + j) = make_double2(a[j], hi_);
                        }
                    }
                    if (W > 1) {
                        if (lane == 63) s_bnd[(i + 1) & 1][wave] = a[NS - 1];
                        __syncthreads();
                    }
                    ++i;
                }
            }
        }
        if (c + 1 < nchunks) SP_CTC_COMMIT(c + 1);
        __syncthreads();
    }

    if (dir == 0) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int sp_ = tid * NS + j;
            if (sp_ == S - 1) s_fin[0] = a[j];
            if (sp_ == S - 2) s_fin[1] = a[j];
        }
        __syncthreads();
        if (tid == 0) {
            const double p = (S > 1) ? s_fin[0] + s_fin[1] : s_fin[0];
            const double ll2 = log2_wide(p) + (double)E;  // log2 likelihood
            nll2[b] = -ll2;
            nll[b] = (float)(-ll2 * 0.69314718055994530942);
        }
    }
}

__device__ __forceinline__ void store4(float* g, int c, float a, float b, float x, float y) {
    reinterpret_cast<float4*>(g)[c] = make_float4(a, b, x, y);
}
__device__ __forceinline__ void store4(__hip_bfloat16* g, int c, float a, float b, float x, float y) {
    uint2 o;
    o.x = (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(a)) |
          ((uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(b)) << 16);
    o.y = (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(x)) |
          ((uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(y)) << 16);
    reinterpret_cast<uint2*>(g)[c] = o;
}

__device__ __forceinline__ void store4(__half* g, int c, float a, float b, float x, float y) {
    uint2 o;
    o.x = (uint32_t)__builtin_bit_cast(unsigned short, __float2half(a)) |
          ((uint32_t)__builtin_bit_cast(unsigned short, __float2half(b)) << 16);
    o.y = (uint32_t)__builtin_bit_cast(unsigned short, __float2half(x)) |
          ((uint32_t)__builtin_bit_cast(unsigned short, __float2half(y)) << 16);
    reinterpret_cast<uint2*>(g)[c] = o;
}

// The DENSE half of the gradient: softmax(z) * scale over the whole vocabulary, zero rows beyond an utterance's
// length.  It needs K1's log-sum-exp only, so it does not wait for the lattice: the workgroups behind the 2B lattice
// workgroups of ctc_lattice1_kernel's grid run it on the CUs (and the memory bandwidth) the lattice leaves idle.
// (id, n) = this wave's index among the n waves that share the rows.
template <typename G>
__device__ __forceinline__ void ctc_dense_rows(
    const float* __restrict__ logits, int64_t st, int64_t sb, const int32_t* __restrict__ in_len,
    const int32_t* __restrict__ lab_len, int T, int B, int V, int Lmax, const float* __restrict__ lse2,
    float grad_scale, G* __restrict__ grad, int id, int n, int lane) {
    const bool vec = ((V & 3) == 0) && ((st & 3) == 0) && ((sb & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0;
    const int64_t nrows = (int64_t)T * B;
    const bool tmaj = st > sb;   // rows are walked in memory order: neighbouring waves then touch neighbouring lines
    // the arguments are log2-probabilities (<= 0 up to rounding): the bare v_exp_f32 (probabilities below 2^-126
    // flush to zero) instead of exp2f's denormal-safe sequence, which made this half VALU-bound at two waves per SIMD
    if (vec && V <= 1024) {
        // few waves run this (two workgroups per CU): each keeps kR rows = up to 16 KB in flight
        constexpr int kR = 4;
        const int nc = V >> 2;
        for (int64_t r0 = (int64_t)id * kR; r0 < nrows; r0 += (int64_t)n * kR) {
            float4 zz[kR][4];
            float l2[kR], scale[kR];
            bool live[kR];
            G* g[kR];
#pragma unroll
            for (int q = 0; q < kR; ++q) {
                const int64_t r = min(r0 + q, nrows - 1);
                const int b = tmaj ? (int)(r % B) : (int)(r / T), t = tmaj ? (int)(r / B) : (int)(r % T);
                const float4* z4 = reinterpret_cast<const float4*>(logits + t * st + b * sb);
#pragma unroll
                for (int c = 0; c < 4; ++c) zz[q][c] = z4[min(lane + 64 * c, nc - 1)];
                const int Tb = min(max(in_len[b], 0), T);
                const int L = min(max(lab_len[b], 0), Lmax);
                live[q] = t < Tb;   // rows beyond the utterance's length are zeros
                scale[q] = grad_scale / ((float)B * (float)max(L, 1));
                l2[q] = lse2[(int64_t)b * T + t];
                g[q] = grad + t * st + b * sb;
            }
#pragma unroll
            for (int q = 0; q < kR; ++q) {
                if (r0 + q >= nrows) break;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (lane + 64 * c >= nc) continue;
                    if (!live[q]) {
                        store4(g[q], lane + 64 * c, 0.f, 0.f, 0.f, 0.f);
                    } else {
                        const float4 v = zz[q][c];
                        store4(g[q], lane + 64 * c, __builtin_amdgcn_exp2f(fmaf(v.x, kLog2e, -l2[q])) * scale[q],
                               __builtin_amdgcn_exp2f(fmaf(v.y, kLog2e, -l2[q])) * scale[q], __builtin_amdgcn_exp2f(fmaf(v.z, kLog2e, -l2[q])) * scale[q],
                               __builtin_amdgcn_exp2f(fmaf(v.w, kLog2e, -l2[q])) * scale[q]);
                    }
                }
            }
        }
        return;
    }
    for (int64_t r = id; r < nrows; r += n) {
        const int b = tmaj ? (int)(r % B) : (int)(r / T), t = tmaj ? (int)(r / B) : (int)(r % T);
        const int Tb = min(max(in_len[b], 0), T);
        G* g = grad + t * st + b * sb;
        if (t >= Tb) {
            if (vec) {
                for (int c = lane; c < (V >> 2); c += 64) store4(g, c, 0.f, 0.f, 0.f, 0.f);
            } else {
                for (int c = lane; c < V; c += 64) Elem<G>::st(g + c, 0.f);
            }
            continue;
        }
        const int L = min(max(lab_len[b], 0), Lmax);
        const float scale = grad_scale / ((float)B * (float)max(L, 1));
        const float* z = logits + t * st + b * sb;
        const float l2 = lse2[(int64_t)b * T + t];
        if (vec) {
            const float4* z4 = reinterpret_cast<const float4*>(z);
            for (int c = lane; c < (V >> 2); c += 64) {
                const float4 zz = z4[c];
                store4(g, c, exp2f(fmaf(zz.x, kLog2e, -l2)) * scale, exp2f(fmaf(zz.y, kLog2e, -l2)) * scale,
                       exp2f(fmaf(zz.z, kLog2e, -l2)) * scale, exp2f(fmaf(zz.w, kLog2e, -l2)) * scale);
            }
        } else {
            for (int c = lane; c < V; c += 64) Elem<G>::st(g + c, exp2f(fmaf(z[c], kLog2e, -l2)) * scale);
        }
    }
}

// -------------------------------------------------------------------------------------------
// K2, single-wave lattices (S <= 512: every sentencepiece-sized transcript, and 15 s of characters): the recursion wave is
// issue-bound once the chain is short, so the emission work moves to H HELPER waves (2 at two states per lane, 4 at four and eight) of the same
// workgroup.  Helpers turn the compact log2-prob rows of a chunk of frames into pruned f64
// emissions y[frame][position] in LDS (double-buffered, one barrier per chunk); the recursion wave
// only reads them back (ds_read_b128), shifts, adds, multiplies and spills.
// -------------------------------------------------------------------------------------------
template <int NS, int H>   // H helper waves (the emissions of a chunk's frames are dealt to them round-robin)
__global__ __launch_bounds__(kLat1Threads) void ctc_lattice1_kernel(
    const float* __restrict__ rows, const int64_t* __restrict__ labels, int label_stride,
    const int32_t* __restrict__ in_len, const int32_t* __restrict__ lab_len, int T, int V,
    int blank, int Lmax, int RW, int SP, int modeB, double* __restrict__ alpha,
    double* __restrict__ beta_rev, double* __restrict__ aux, double* __restrict__ nll2,
    float* __restrict__ nll, int B, const float* __restrict__ logits, int64_t st, int64_t sb,
    const float* __restrict__ lse2, float grad_scale, void* __restrict__ grad, int grad_bf16) {
    // grid = 2B lattice workgroups (placed first), then the workgroups of the gradient's dense half
    if ((int)blockIdx.x >= 2 * B) {
        constexpr int nw = kLat1Threads / 64;
        const int id = ((int)blockIdx.x - 2 * B) * nw + (int)(threadIdx.x >> 6), n = ((int)gridDim.x - 2 * B) * nw;
        if (grad_bf16 == 1)
            ctc_dense_rows(logits, st, sb, in_len, lab_len, T, B, V, Lmax, lse2, grad_scale, (__hip_bfloat16*)grad, id, n,
                           (int)(threadIdx.x & 63));
        else if (grad_bf16 == 2)
            ctc_dense_rows(logits, st, sb, in_len, lab_len, T, B, V, Lmax, lse2, grad_scale, (__half*)grad, id, n,
                           (int)(threadIdx.x & 63));
        else
            ctc_dense_rows(logits, st, sb, in_len, lab_len, T, B, V, Lmax, lse2, grad_scale, (float*)grad, id, n,
                           (int)(threadIdx.x & 63));
        return;
    }
    constexpr int TB = NS == 16 ? kRescale : 64 / NS;   // frames per chunk (at least one rescale block)
    constexpr int PW = 64 * NS;   // positions per frame
    constexpr int NK = NS / 2;    // labels per lane
    constexpr int HF = TB / H;    // frames per helper wave and chunk
    static_assert(TB % H == 0, "helpers share a chunk's frames evenly");
    // emissions as [buffer][frame][state pair][lane][2]: a lane's pairs lie 1 KB apart, so every 16-byte read or write
    // of a wave is one contiguous KB (lane * NS doubles apart, the reads of four / eight states per lane hit the same
    // banks two / four times over)
    __shared__ __attribute__((aligned(16))) double s_y[2][TB][PW];   // 64 KB (128 KB at sixteen states per lane)
    __shared__ int s_rep[64];
    __shared__ double s_fin[2];

    const int b = (int)blockIdx.x % B, dir = (int)blockIdx.x / B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Tb = min(max(in_len[b], 0), T);
    const int L = min(max(lab_len[b], 0), Lmax);
    const int S = 2 * L + 1;
    const int64_t* lab = labels + (int64_t)b * label_stride;
    const float* rows_b = rows + (int64_t)b * T * RW;

    if (Tb == 0) {
        if (dir == 0 && tid == 0) {
            nll[b] = (L == 0) ? 0.f : INFINITY;
            nll2[b] = (L == 0) ? 0.0 : (double)INFINITY;
        }
        return;
    }
    if (wave > H) {
        // The workgroup is as large as the dense half wants it (eight waves, one workgroup per CU: a lattice CU then
        // holds no streaming waves, which slowed the recursion to half its pace through the CU's memory pipeline).
        // The lattice's spare waves only keep the barrier count: one for the set-up below, one per chunk.
        const int nch = (Tb + TB - 1) / TB;
        for (int c = 0; c <= nch; ++c) __syncthreads();
        return;
    }

    // ---- per-position constants of the positions lane*NS .. lane*NS+NS-1 (every wave computes them) ----
    auto labp = [&](int k) { return min(max((int)lab[dir ? L - 1 - k : k], 0), V - 1); };
    int need[NS];    // frames still required after this one to reach the end from this position
    int rlab[NK];    // row index of the lane's labels
    double skm[NK];  // 1.0 where the s-2 transition into the label position exists
    {
        int rflag[NK], cnt = 0;
#pragma unroll
        for (int q = 0; q < NK; ++q) {
            const int k = lane * NK + q;
            rflag[q] = (k >= 1 && k < L && labp(k) == labp(k - 1)) ? 1 : 0;   // mandatory blank before k
            cnt += rflag[q];
            rlab[q] = 0;
            skm[q] = 0.0;
            if (k < L) {
                // real label index of recursion-order label k is (dir ? L-1-k : k); rows are indexed by
                // class (modeB) or by real label position + 1
                rlab[q] = modeB ? labp(k) : (dir ? L - k : k + 1);
                skm[q] = (k >= 1 && !rflag[q]) ? 1.0 : 0.0;
            }
        }
        if (wave == 0) s_rep[lane] = cnt;
        __syncthreads();
        int after = 0;  // repeats among labels owned by higher lanes
        for (int u = lane + 1; u < 64; ++u) after += s_rep[u];
#pragma unroll
        for (int q = NK - 1; q >= 0; --q) {
            const int k = lane * NK + q;
            const int need_label = (L - 1 - k) + after;           // position 2k+1
            need[2 * q + 1] = (k < L) ? need_label : 0x3FFFFFFF;
            need[2 * q] = (k < L) ? need_label + 1 : (k == L ? 0 : 0x3FFFFFFF);   // final blank: 0
            after += rflag[q];
        }
    }
    const int rblank = modeB ? blank : 0;
    const int nchunks = (Tb + TB - 1) / TB;

    if (wave > 0) {
        // ------------------------------------------------------------------ helpers
        const int hw = wave - 1;
        // rows of chunk c+1 are requested before chunk c is converted: a helper never waits on HBM
        float lpb[HF], lpl[HF][NK], nxb[HF], nxl[HF][NK];
        auto fetch = [&](int c, float (&pb)[HF], float (&pl)[HF][NK]) {
#pragma unroll
            for (int f = 0; f < HF; ++f) {
                const int i = min(c * TB + H * f + hw, Tb - 1);
                const int t = dir ? Tb - 1 - i : i;
                const float* row = rows_b + (int64_t)t * RW;
                pb[f] = row[rblank];
#pragma unroll
                for (int q = 0; q < NK; ++q) pl[f][q] = row[rlab[q]];
            }
        };
        fetch(0, lpb, lpl);
        for (int c = 0; c < nchunks; ++c) {
            fetch(min(c + 1, nchunks - 1), nxb, nxl);
#pragma unroll
            for (int f = 0; f < HF; ++f) {
                const int i = c * TB + H * f + hw;
                const int rem = Tb - 1 - i;  // frames left: positions that need more are dead (exact
                                             // pruning: a dead position only ever feeds dead positions)
                const float xb = fmaxf(lpb[f], -1200.f);
                const float nb_ = floorf(xb);
                const double yb = ldexp((double)__builtin_amdgcn_exp2f(xb - nb_), (int)nb_);
                double yv[NS];
#pragma unroll
                for (int q = 0; q < NK; ++q) {
                    const float xl = fmaxf(lpl[f][q], -1200.f);
                    const float nl = floorf(xl);
                    const double yl = ldexp((double)__builtin_amdgcn_exp2f(xl - nl), (int)nl);
                    yv[2 * q] = need[2 * q] <= rem ? yb : 0.0;
                    yv[2 * q + 1] = need[2 * q + 1] <= rem ? yl : 0.0;
                }
                double* dst = &s_y[c & 1][H * f + hw][lane * 2];
#pragma unroll
                for (int j = 0; j < NS; j += 2)
                    *reinterpret_cast<double2*>(dst + j * 64) = make_double2(yv[j], yv[j + 1]);
            }
#pragma unroll
            for (int f = 0; f < HF; ++f) {
                lpb[f] = nxb[f];
#pragma unroll
                for (int q = 0; q < NK; ++q) lpl[f][q] = nxl[f][q];
            }
            __syncthreads();  // chunk c is ready; the recursion wave released buffer (c+1)&1 at this barrier
        }
        return;
    }

    // ---------------------------------------------------------------------- recursion wave
    // the serial chain outranks the streaming waves of the dense half that may share its SIMD
    __builtin_amdgcn_s_setprio(3);
    constexpr bool kPack = NS >= 8 || kPackAll;   // four-byte states (lat_pack)
    constexpr int kEsz = kPack ? 4 : 8;
    char* outb = reinterpret_cast<char*>(dir ? beta_rev : alpha) + (int64_t)b * T * SP * kEsz;
    double* out = reinterpret_cast<double*>(outb);
    // a(-1) = delta(position 0): one step of the general recurrence then yields the CTC start
    // condition (positions 0 and 1 take their emission, everything else 0)
    double a[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) a[j] = (lane * NS + j == 0) ? 1.0 : 0.0;
    int E = 0;         // alpha_true = a * 2^E
    // Everything in the frame loop is straight-line: an exec-mask branch costs the serial wave ~30
    // cycles per frame.  Lanes beyond the padded row keep storing into a dummy row (step 0), and the
    // scale is not stored per row but once per rescale block (all lanes write the same value):
    // aux block of this (dir, b) = dummy[kAuxDummy] | Ecum[T/kRescale + 2].
    const int aux_stride = kAuxDummy + T / kRescale + 2;
    double* aux_b = aux + ((int64_t)dir * B + b) * aux_stride;
    double* ecum = aux_b + kAuxDummy;
    constexpr bool kAllLanes = NS != 4 || kPack;   // the row pitch is 64 NS (ctc_dims): every lane has its columns
    const bool spill_ok = kAllLanes || lane * NS < SP;
    // row pointer of the frame being written: walks forward (alpha) or backward (beta) by one row
    double* orow = spill_ok ? out + (int64_t)(dir ? Tb - 1 : 0) * SP + lane * NS : aux_b + lane * NS;   // NS == 4 only
    const int64_t ostep = spill_ok ? (dir ? -(int64_t)SP : (int64_t)SP) : 0;
    // kAllLanes: buffer stores -- the utterance's rows behind one descriptor, the row as a scalar byte offset walked by
    // scalar adds, the lane's column as a fixed vector offset: no vector instruction per frame goes into addressing
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)outb, 0, (int)((int64_t)T * SP * kEsz), 0x00020000);
    int osoff = (dir ? Tb - 1 : 0) * SP * kEsz;
    const int ostep_b = (dir ? -SP : SP) * kEsz;
    // lanes whose states all lie beyond the utterance's lattice store nothing: their offset is out of the descriptor's
    // range and the hardware drops the lanes (the consumers read columns < S only)
    const int loff = (kPack && lane * NS >= S) ? 0x7F000000 : lane * (kPack && NS == 2 ? 8 : 16);
    auto step = [&](const double (&yq)[NS]) {
        const double p1 = lane_shr1(a[NS - 1]);
        double n_[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            // position parity == state parity; s-1 and s-2 of an odd (label) position j are
            // a[j-1], a[j-2] (the previous lane's last state for j == 1)
            const double x1 = (j >= 1) ? a[j - 1] : p1;
            const double x2 = (j >= 2) ? a[j - 2] : p1;
            const double base = (j >= 1) ? a[j] + x1 : a[j];   // j >= 1: off the DPP chain
            n_[j] = (j & 1) ? fma(skm[j >> 1], x2, base) * yq[j] : (a[j] + x1) * yq[j];
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) a[j] = n_[j];
    };
    auto spill = [&]() {   // one vector-memory op per lane and frame (NS == 2), NS / 4 of them packed
        using u4 = __attribute__((ext_vector_type(4))) unsigned;
#ifdef SP_LAT1_NOSTORE   // lab: timing only
        if (T >= 0) return;
#endif
        if constexpr (kPack && NS == 2) {
            using u2 = __attribute__((ext_vector_type(2))) unsigned;
            const u2 v = {lat_pack(a[0]), lat_pack(a[1])};
            __builtin_amdgcn_raw_buffer_store_b64(v, orsrc, loff, osoff, 0);
        } else if constexpr (kPack) {
#pragma unroll
            for (int j = 0; j < NS; j += 4) {
                const u4 v = {lat_pack(a[j]), lat_pack(a[j + 1]), lat_pack(a[j + 2]), lat_pack(a[j + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, loff + j * 256, osoff, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NS; j += 2) {
                if constexpr (kAllLanes) {
                    using d2 = __attribute__((ext_vector_type(2))) double;
                    const d2 v = {a[j], a[j + 1]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), orsrc, loff + j * 8, osoff, 0);
                } else
                    *reinterpret_cast<double2*>(orow + j) = make_double2(a[j], a[j + 1]);
            }
        }
        if constexpr (kAllLanes) osoff += ostep_b;
        else orow += ostep;
    };
    int i = 0;
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        const int nsteps = min(TB, Tb - c * TB);
        int k0 = 0;
        for (; k0 + kRescale <= nsteps; k0 += kRescale) {   // full blocks: straight-line code
            // exponent of the current maximum, folded into the block's last frame.  The states are non-negative, so
            // their high words order like the values: one integer max per state instead of canonicalising f64
            // maxima, and the exponent field is cut out of the wave's maximum on the scalar unit
            int mh = __double2hiint(a[0]);
#pragma unroll
            for (int j = 1; j < NS; ++j) mh = max(mh, __double2hiint(a[j]));
            const int e = (wave_max_i(mh) >> 20) & 0x7FF;
            const int dp = e ? e - (1023 + kTopBits) : 0;
            // the block's emissions: all eight frames up front, or (sixteen states per lane: 32 registers a frame) kAhead
            // frames ahead of the one being computed
            constexpr int kAhead = NS == 16 ? 3 : kRescale;
            double y[kRescale][NS];
            auto fetch_y = [&](int q) {
                const double* src = &s_y[c & 1][k0 + q][lane * 2];
#pragma unroll
                for (int j = 0; j < NS; j += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(src + j * 64);
                    y[q][j] = v.x;
                    y[q][j + 1] = v.y;
                }
            };
#pragma unroll
            for (int q = 0; q < kAhead; ++q) fetch_y(q);
#pragma unroll
            for (int q = 0; q < kRescale; ++q) {
                if (q + kAhead < kRescale) fetch_y(q + kAhead);
                if (q == kRescale - 1) {
#pragma unroll
                    for (int j = 0; j < NS; ++j) y[q][j] = ldexp(y[q][j], -dp);
                }
                step(y[q]);
                spill();
                if constexpr (NS == 16) __builtin_amdgcn_sched_barrier(0);   // keeps the fetches where they are written
            }
            E += dp;
            ecum[i / kRescale] = (double)E;   // the scale of the block's last frame and of the next block's first seven
            i += kRescale;
        }
        for (; k0 < nsteps; ++k0, ++i) {   // tail frames of the sequence: no rescale
            double yq[NS];
            const double* src = &s_y[c & 1][k0][lane * 2];
#pragma unroll
            for (int j = 0; j < NS; ++j) yq[j] = src[(j >> 1) * 128 + (j & 1)];
            step(yq);
            spill();
        }
    }
    if (dir == 0) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int sp_ = lane * NS + j;
            if (sp_ == S - 1) s_fin[0] = a[j];
            if (sp_ == S - 2) s_fin[1] = a[j];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) {
            const double p = (S > 1) ? s_fin[0] + s_fin[1] : s_fin[0];
            const double ll2 = log2_wide(p) + (double)E;  // log2 likelihood
            nll2[b] = -ll2;
            nll[b] = (float)(-ll2 * 0.69314718055994530942);
        }
    }
}

// -------------------------------------------------------------------------------------------
// K3: gradient rows (+ the scalar loss in block 0)
// -------------------------------------------------------------------------------------------
// G = element type of the gradient (float, or bf16 when the consumer is a bf16 GEMM: the rounding the
// caller would apply in a pass of its own happens in the store)
template <typename G>
__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float* __restrict__ logits, int64_t st, int64_t sb, const int64_t* __restrict__ labels,
    int label_stride, const int32_t* __restrict__ in_len, const int32_t* __restrict__ lab_len,
    int T, int B, int V, int blank, int Lmax, int RW, int SP, int modeB,
    const float* __restrict__ lse2, const float* __restrict__ rows, const void* __restrict__ alpha,
    const void* __restrict__ beta_rev, const double* __restrict__ aux,
    const double* __restrict__ nll2d, const float* __restrict__ nll, float grad_scale,
    G* __restrict__ grad, float* __restrict__ loss, int perm) {
    extern __shared__ __attribute__((aligned(16))) float s_occ[];  // [waves][V]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row pointers stay scalar
    const int wpb = blockDim.x >> 6;

    if (blockIdx.x == 0 && wave == 0 && loss) {
        float acc = 0.f;
        // the same (clamped) target length the lattice scored
        for (int b = lane; b < B; b += 64) acc += nll[b] / (float)max(min(lab_len[b], Lmax), 1);
        acc = wave_sum(acc);
        if (lane == 0) loss[0] = acc / (float)B;
    }
    if (!grad) return;

    float* occ = s_occ + wave * V;
    const bool vec = ((V & 3) == 0) && ((st & 3) == 0) && ((sb & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0;
    const int64_t nrows = (int64_t)T * B;
    for (int64_t r = (int64_t)blockIdx.x * wpb + wave; r < nrows; r += (int64_t)gridDim.x * wpb) {
        const int b = (int)(r / T), t = (int)(r % T);
        const int Tb = min(max(in_len[b], 0), T);
        G* g = grad + t * st + b * sb;
        if (t >= Tb) {
            if (vec) {
                for (int c = lane; c < (V >> 2); c += 64) store4(g, c, 0.f, 0.f, 0.f, 0.f);
            } else {
                for (int c = lane; c < V; c += 64) Elem<G>::st(g + c, 0.f);
            }
            continue;
        }
        if (vec) {
            for (int c = lane; c < (V >> 2); c += 64) reinterpret_cast<float4*>(occ)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int c = lane; c < V; c += 64) occ[c] = 0.f;
        }
        const int L = min(max(lab_len[b], 0), Lmax);
        const int S = 2 * L + 1;
        const float scale = grad_scale / ((float)B * (float)max(L, 1));
        const int64_t lrow_i = (int64_t)b * T + t;
        // 2^gexp * alpha_hat * beta_hat / y = the state posterior; slot SP-1 of a row is its scale E
        // (multi-wave lattices) or, for the single-wave kernel, Ecum[block] of its aux block
        double gexp = nll2d[b];
        if (aux) {
            const int aux_stride = kAuxDummy + T / kRescale + 2;
            const int ka = (t + 1) / kRescale - 1, kb = (Tb - t) / kRescale - 1;   // frame index i: t and Tb-1-t
            if (ka >= 0) gexp += aux[(int64_t)b * aux_stride + kAuxDummy + ka];
            if (kb >= 0) gexp += aux[((int64_t)B + b) * aux_stride + kAuxDummy + kb];
        } else {
            gexp += lat_load(alpha, lrow_i, SP, SP - 1, 0) + lat_load(beta_rev, lrow_i, SP, SP - 1, 0);
        }
        const float* lrow = rows + ((int64_t)b * T + t) * RW;
        const int64_t* lab = labels + (int64_t)b * label_stride;

        float blank_acc = 0.f;
        for (int s = lane; s < S; s += 64) {
            int cls = blank;
            if (s & 1) cls = min(max((int)lab[(s - 1) >> 1], 0), V - 1);
            const int ri = modeB ? cls : ((s & 1) ? (s + 1) >> 1 : 0);
            // alpha_hat and beta_hat can each sit ~1000 binary orders below their row maxima where the
            // posterior peaks: multiply mantissas, add exponents, one ldexp
            const double gq = gexp - (double)lrow[ri];
            const double gn = floor(gq);
            int xa, xb;
            const double ma = frexp(lat_load(alpha, lrow_i, SP, spill_col(s, perm), perm), &xa);
            const double mb = frexp(lat_load(beta_rev, lrow_i, SP, spill_col(S - 1 - s, perm), perm), &xb);
            const double pr = ma * mb * (double)__builtin_amdgcn_exp2f((float)(gq - gn));
            const double ge = fmax(fmin(gn + (double)(xa + xb), 4000.0), -4000.0);
            const float gam = (float)ldexp(pr, (int)ge);
            if (s & 1) atomicAdd(&occ[cls], gam);
            else blank_acc += gam;
        }
        blank_acc = wave_sum(blank_acc);
        if (lane == 0) atomicAdd(&occ[blank], blank_acc);
        __builtin_amdgcn_wave_barrier();
        const float* z = logits + t * st + b * sb;
        const float l2 = lse2[(int64_t)b * T + t];
        if (vec) {
            const float4* z4 = reinterpret_cast<const float4*>(z);
            const float4* o4 = reinterpret_cast<const float4*>(occ);
            for (int c = lane; c < (V >> 2); c += 64) {
                const float4 zz = z4[c], oo = o4[c];
                store4(g, c, (exp2f(fmaf(zz.x, kLog2e, -l2)) - oo.x) * scale,
                       (exp2f(fmaf(zz.y, kLog2e, -l2)) - oo.y) * scale,
                       (exp2f(fmaf(zz.z, kLog2e, -l2)) - oo.z) * scale,
                       (exp2f(fmaf(zz.w, kLog2e, -l2)) - oo.w) * scale);
            }
        } else {
            for (int c = lane; c < V; c += 64) {
                const float sm = exp2f(fmaf(z[c], kLog2e, -l2));
                Elem<G>::st(g + c, (sm - occ[c]) * scale);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// K3 in two halves (single-wave lattices).  The DENSE half (ctc_dense_rows) has already run inside the lattice
// launch.  The SPARSE half runs after it: per frame it forms the state posteriors exactly as ctc_grad_kernel does and
// overwrites the <= L + 1 classes that carry any (label classes and blank) with (p - occupancy) * scale, p taken
// from K1's compact log2-prob row: the same f32 expression the fused kernel evaluates, so the two routes agree bit
// for bit (up to the association of the blank states' sum).  One wave = kR consecutive frames of one utterance: the label classes are read once, the alpha / beta /
// log-prob values of all kR frames are in flight together (a single row per wave left the kernel latency-bound:
// 26 us for 46 MB), and the occupancies are summed in compact per-frame slots (slot 0 = blank, slot 1 + k = one
// label position that carries position k's class), so nothing V-sized is cleared.
// NI = lattice states per lane (S <= 64 NI); NI >= 8: packed rows (lat_pack), the lanes take the states in the
// producer's store order, so that every load of the wave is 256 contiguous bytes of an alpha row.
// kDense (vocabularies of <= 64 classes with class-indexed rows, the reference's characters): the slots are the
// classes themselves and the wave writes the frame's WHOLE gradient row, p - occupancy for every class and zeros
// beyond the utterance's length -- there is no dense half, the logits are not read again.
template <int NI, bool kDense>
constexpr int kSparseGroups = 1;   // groups of kR frames per item, the next group's loads in flight behind the current one's arithmetic (measured: no gain at the sizes this kernel still serves; ctc_char_grad_kernel is the pipelined form)

template <typename G, int NI, int kR, bool kDense>
__global__ __launch_bounds__(256) void ctc_sparse_kernel(
    int64_t st, int64_t sb, const int64_t* __restrict__ labels, int label_stride,
    const int32_t* __restrict__ in_len, const int32_t* __restrict__ lab_len, int T, int B, int V, int blank,
    int Lmax, int RW, int SP, int modeB, const float* __restrict__ rows, const void* __restrict__ alpha,
    const void* __restrict__ beta_rev, const double* __restrict__ aux, const double* __restrict__ nll2d,
    const float* __restrict__ nll, float grad_scale, G* __restrict__ grad, float* __restrict__ loss, int perm) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];  // per wave: first[V] (int) | acc[kR][AL]; kDense: acc[kR][V]
    constexpr bool kPack = NI >= 8 || kPackAll;
    using LT = typename std::conditional<kPack, uint32_t, double>::type;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row pointers stay scalar
    const int wpb = blockDim.x >> 6;

    if (blockIdx.x == 0 && wave == 0 && loss) {
        float acc = 0.f;
        for (int b = lane; b < B; b += 64) acc += nll[b] / (float)max(min(lab_len[b], Lmax), 1);
        acc = wave_sum(acc);
        if (lane == 0) loss[0] = acc / (float)B;
    }
    if (!grad) return;

    const int AL = kDense ? V : Lmax + 2;
    int* first = reinterpret_cast<int*>(s_dyn + (size_t)wave * ((kDense ? 0 : V) + kR * AL));
    float* acc = reinterpret_cast<float*>(first + (kDense ? 0 : V));
    // One item = kG groups of kR consecutive frames of one utterance.  kG > 1 (whole-row mode at eight and sixteen
    // states per lane, where an item's set-up -- up to sixteen label gathers per lane -- and its load latency were what
    // the kernel spent its time on): the classes are set up once, and group g + 1's alpha / beta / log-prob loads are
    // in flight while group g is computed.
    constexpr int kG = kSparseGroups<NI, kDense>;
    const int groups = (T + kR * kG - 1) / (kR * kG);
    const int64_t nitems = (int64_t)B * groups;
    const int aux_stride = kAuxDummy + T / kRescale + 2;
    // state of (lane, i).  Packed rows are [NI / 4 planes][64 producer lanes][4]: load i of the wave takes plane
    // i % (NI / 4) of the producer lanes 16 (i / (NI / 4)) .. + 15, four consecutive states per producer lane
    auto state_of = [&](int i) {
        if constexpr (NI >= 8) return (((i / (NI / 4)) * 16 + (lane >> 2)) * NI) + (i % (NI / 4)) * 4 + (lane & 3);
        else return lane + 64 * i;
    };
    for (int64_t it = (int64_t)blockIdx.x * wpb + wave; it < nitems; it += (int64_t)gridDim.x * wpb) {
        const int b = (int)(it / groups), t0 = (int)(it % groups) * (kR * kG);
        const int Tb = min(max(in_len[b], 0), T);
        auto zero_rows = [&](int ta, int te) {   // rows beyond the utterance's length are zeros
            for (int t = ta; t < min(te, T); ++t)
                if (lane < V) Elem<G>::st(grad + (int64_t)t * st + b * sb + lane, 0.f);
        };
        if (t0 >= Tb) {
            if constexpr (kDense) zero_rows(t0, t0 + kR * kG);
            continue;
        }
        const int L = min(max(lab_len[b], 0), Lmax);
        const int S = 2 * L + 1;
        const float scale = grad_scale / ((float)B * (float)max(L, 1));
        const int64_t* lab = labels + (int64_t)b * label_stride;
        const double gbase = nll2d[b];

        // ---- every load of a group is issued before anything waits: classes, then kR frames of alpha / beta / log-probs
        int cls[NI], ri[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            // one unconditional gather per state (a load under a per-lane condition compiles to a branch and a full wait each)
            const int s_ = state_of(i);
            const bool has = (s_ & 1) && s_ < S;
            const int c = L > 0 ? (int)lab[has ? (s_ - 1) >> 1 : 0] : blank;
            cls[i] = has ? min(max(c, 0), V - 1) : blank;
        }
        LT avn[kR][NI], bvn[kR][NI];
        double gexpn[kR];
        float lrn[kDense ? 1 : kR][kDense ? 1 : NI], palln[kR];   // kDense: log2-prob of class `lane`, the states' by shuffle
        auto issue_lattice = [&](int tg) {
#pragma unroll
            for (int q = 0; q < kR; ++q) {
                const int t = min(tg + q, Tb - 1);
                const LT* arow = reinterpret_cast<const LT*>(alpha) + ((int64_t)b * T + t) * SP;
                const LT* brow = reinterpret_cast<const LT*>(beta_rev) + ((int64_t)b * T + t) * SP;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int s_ = min(state_of(i), S - 1);
                    avn[q][i] = arow[spill_col(s_, perm)];
                    bvn[q][i] = brow[spill_col(S - 1 - s_, perm)];
                }
                // 2^gexp * alpha_hat * beta_hat / y = the state posterior (Ecum of the two rescale blocks, see ctc_grad_kernel)
                const int ka = (t + 1) / kRescale - 1, kb = (Tb - t) / kRescale - 1;
                double ge = gbase;
                if (ka >= 0) ge += aux[(int64_t)b * aux_stride + kAuxDummy + ka];
                if (kb >= 0) ge += aux[((int64_t)B + b) * aux_stride + kAuxDummy + kb];
                gexpn[q] = ge;
            }
        };
        auto issue_rows = [&](int tg) {
#pragma unroll
            for (int q = 0; q < kR; ++q) {
                const float* lrow = rows + ((int64_t)b * T + min(tg + q, Tb - 1)) * RW;
                if constexpr (kDense) palln[q] = lrow[min(lane, V - 1)];
                else {
                    palln[q] = 0.f;
#pragma unroll
                    for (int i = 0; i < NI; ++i) lrn[q][i] = lrow[ri[i]];
                }
            }
        };
        issue_lattice(t0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int s_ = state_of(i);
            ri[i] = modeB ? cls[i] : ((s_ & 1) ? (s_ + 1) >> 1 : 0);
            if (s_ >= S) ri[i] = 0;
        }
        issue_rows(t0);

        int slot[NI];
        if constexpr (kDense) {
#pragma unroll
            for (int i = 0; i < NI; ++i) slot[i] = cls[i];
        } else {
            // ---- slot of every label position: ONE position of its class (whichever write the LDS ordered last: every
            // position of the class reads the same survivor; the wave's LDS operations execute in order)
            __builtin_amdgcn_wave_barrier();   // the previous item's reads of first[] / acc[] are done
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int s_ = state_of(i);
                if ((s_ & 1) && s_ < S) first[cls[i]] = (s_ - 1) >> 1;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int s_ = state_of(i);
                slot[i] = ((s_ & 1) && s_ < S) ? 1 + first[cls[i]] : 0;
            }
        }

#pragma unroll 1
        for (int g = 0; g < kG; ++g) {
            const int tg = t0 + g * kR;
            if (tg >= T) break;
            if (tg >= Tb) {   // kG > 1 only: the rest of the item lies beyond the utterance
                if constexpr (kDense) zero_rows(tg, t0 + kR * kG);
                break;
            }
            LT av[kR][NI], bv[kR][NI];
            double gexp[kR];
            float lr[kDense ? 1 : kR][kDense ? 1 : NI], pall[kR];
#pragma unroll
            for (int q = 0; q < kR; ++q) {
                gexp[q] = gexpn[q];
                pall[q] = palln[q];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    av[q][i] = avn[q][i];
                    bv[q][i] = bvn[q][i];
                    if constexpr (!kDense) lr[q][i] = lrn[q][i];
                }
            }
            if (kG > 1 && g + 1 < kG && tg + kR < Tb) {
                issue_lattice(tg + kR);
                issue_rows(tg + kR);
            }
            __builtin_amdgcn_wave_barrier();   // the previous group's reads of acc[] are done
            for (int c = lane; c < kR * AL; c += 64) acc[c] = 0.f;
            __builtin_amdgcn_wave_barrier();

            // ---- posteriors of the kR frames into their slots
            float blank_acc[kR];
#pragma unroll
            for (int q = 0; q < kR; ++q) {
                blank_acc[q] = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int s_ = state_of(i);
                    float lrv;
                    if constexpr (kDense) lrv = __shfl(pall[q], cls[i]);
                    else lrv = lr[q][i];
                    // alpha_hat and beta_hat can each sit ~1000 binary orders below their row maxima where the
                    // posterior peaks: multiply mantissas, add exponents, one ldexp
                    const double gq = gexp[q] - (double)lrv;
                    const double gn = floor(gq);
                    int xa, xb;
                    double ad, bd;
                    if constexpr (kPack) { ad = lat_unpack(av[q][i]); bd = lat_unpack(bv[q][i]); }
                    else { ad = av[q][i]; bd = bv[q][i]; }
                    const double ma = frexp(ad, &xa), mb = frexp(bd, &xb);
                    const double pr = ma * mb * (double)__builtin_amdgcn_exp2f((float)(gq - gn));
                    const double ge = fmax(fmin(gn + (double)(xa + xb), 4000.0), -4000.0);
                    float gam = (float)ldexp(pr, (int)ge);
                    if (s_ >= S || tg + q >= Tb) gam = 0.f;
                    if (s_ & 1) {
                        if (gam != 0.f) atomicAdd(&acc[q * AL + slot[i]], gam);
                    } else {
                        blank_acc[q] += gam;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < kR; ++q) blank_acc[q] = wave_sum_dpp(blank_acc[q]);   // kR independent chains
            if (lane < kR) {
                float v = blank_acc[0];
#pragma unroll
                for (int q = 1; q < kR; ++q) v = (lane == q) ? blank_acc[q] : v;
                if constexpr (kDense) atomicAdd(&acc[lane * AL + blank], v);   // a label may (wrongly) name the blank class
                else acc[lane * AL] = v;   // slot 0 is the blank's alone
            }
            __builtin_amdgcn_wave_barrier();

            if constexpr (kDense) {
                // ---- the whole row: class `lane`
#pragma unroll
                for (int q = 0; q < kR; ++q) {
                    if (tg + q >= T) break;
                    if (lane < V) {
                        const float gv = tg + q < Tb ? (exp2f(pall[q]) - acc[q * AL + lane]) * scale : 0.f;
                        Elem<G>::st(grad + (int64_t)(tg + q) * st + b * sb + lane, gv);
                    }
                }
            } else {
                // ---- overwrite the touched classes (every position of a class stores the same value)
#pragma unroll
                for (int q = 0; q < kR; ++q) {
                    if (tg + q >= Tb) break;
                    G* g_ = grad + (int64_t)(tg + q) * st + b * sb;
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int s_ = state_of(i);
                        if (s_ >= S || (!(s_ & 1) && s_ != 0)) continue;   // one store for the blank class
                        Elem<G>::st(g_ + cls[i], (exp2f(lr[q][i]) - acc[q * AL + slot[i]]) * scale);
                    }
                }
            }
        }
    }
}

// K3 for the reference's character vocabulary at eight and sixteen states per lane (V <= 64 classes, class-indexed
// rows, packed lattices: 15 s and 30 s of characters).  One wave walks kCharFrames consecutive frames of one utterance
// and writes their WHOLE gradient rows, (p - occupancy) * scale for every class and zeros beyond the utterance's
// length; there is no dense half and the logits are not read again (p = exp2 of K1's row).  The classes of the lane's
// states are gathered once per item; a frame's states are taken in two halves, each half's alpha / beta words
// requested one half-frame ahead of its arithmetic (registers for one frame in flight, four waves per SIMD).  The
// lanes take the states in the producer's store order: every alpha load of the wave is 256 contiguous bytes.
constexpr int kCharFrames = 8;
template <typename G, int NI, int PS>   // NI states per lane here, PS per lane of the producer (NI / PS recursion waves)
__global__ __launch_bounds__(256, NI == 16 ? 3 : 4) void ctc_char_grad_kernel(
    int64_t st, int64_t sb, const int64_t* __restrict__ labels, int label_stride,
    const int32_t* __restrict__ in_len, const int32_t* __restrict__ lab_len, int T, int B, int V, int blank,
    int Lmax, int RW, int SP, const float* __restrict__ rows, const uint32_t* __restrict__ alpha,
    const uint32_t* __restrict__ beta_rev, const double* __restrict__ aux, const double* __restrict__ nll2d,
    const float* __restrict__ nll, float grad_scale, G* __restrict__ grad, float* __restrict__ loss, int perm) {
    __shared__ float s_acc[4][128];   // per wave: the classes' occupancies | a dummy slot per lane
    constexpr int NH = NI / 2, Q = PS / 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row pointers stay scalar
    const int wpb = blockDim.x >> 6;
    if (blockIdx.x == 0 && wave == 0 && loss) {
        float a_ = 0.f;
        for (int b = lane; b < B; b += 64) a_ += nll[b] / (float)max(min(lab_len[b], Lmax), 1);
        a_ = wave_sum(a_);
        if (lane == 0) loss[0] = a_ / (float)B;
    }
    if (!grad) return;
    float* acc = s_acc[wave];
    const int groups = (T + kCharFrames - 1) / kCharFrames;
    const int64_t nitems = (int64_t)B * groups;
    const int aux_stride = kAuxDummy + T / kRescale + 2;
    // load i of the wave: of producer wave i / PS, plane (i % PS) % Q of its lanes 16 ((i % PS) / Q) .. + 15, four
    // consecutive states per producer lane: 256 contiguous bytes at word woff(i) + lane of the row
    auto state_of = [&](int i) {
        return (i / PS) * 64 * PS + ((((i % PS) / Q) * 16 + (lane >> 2)) * PS) + ((i % PS) % Q) * 4 + (lane & 3);
    };
    auto woff = [](int i) { return (i / PS) * 64 * PS + ((i % PS) % Q) * 256 + ((i % PS) / Q) * 64; };
    const int lc = min(lane, V - 1);
    for (int64_t it = (int64_t)blockIdx.x * wpb + wave; it < nitems; it += (int64_t)gridDim.x * wpb) {
        const int b = (int)(it / groups), t0 = (int)(it % groups) * kCharFrames;
        const int Tb = min(max(in_len[b], 0), T);
        const int te = min(t0 + kCharFrames, Tb);
        if (t0 < Tb) {
            const int L = min(max(lab_len[b], 0), Lmax);
            const int S = 2 * L + 1;
            const float scale = grad_scale / ((float)B * (float)max(L, 1));
            const int64_t* lab = labels + (int64_t)b * label_stride;
            const double gbase = nll2d[b];
            int cls[NI];
            uint32_t valid = 0;   // bit i: state_of(i) < S
            uint32_t boff[NI];   // unsigned 32-bit offsets from wave-uniform row pointers: no 64-bit address registers
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                // one unconditional gather per state (a load under a per-lane condition compiles to a branch and a full wait each)
                const int s_ = state_of(i);
                const bool has = (s_ & 1) && s_ < S;
                const int c = L > 0 ? (int)lab[has ? (s_ - 1) >> 1 : 0] : blank;
                // blank lanes: the blank class (for its log-prob); label lanes: the label's class, or beyond S a slot of
                // the lane's own, so no two lanes ever meet on an occupancy slot they do not share
                cls[i] = (s_ & 1) ? (has ? min(max(c, 0), V - 1) : 64 + lane) : blank;
                valid |= (s_ < S ? 1u : 0u) << i;
                boff[i] = (uint32_t)spill_col(max(S - 1 - s_, 0), perm);
            }
            const uint32_t* abase = alpha + (int64_t)b * T * SP;
            const uint32_t* bbase = beta_rev + (int64_t)b * T * SP;
            const float* rbase = rows + (int64_t)b * T * RW;
            const uint32_t ulane = (uint32_t)lane, ulc = (uint32_t)lc;
            uint32_t a0[NH], b0[NH], a1[NH], b1[NH];
            // states beyond S read columns nobody wrote: whatever they hold is discarded below
            auto issue = [&](int t, int h, uint32_t (&a)[NH], uint32_t (&bb)[NH]) {
                const uint32_t* ar = abase + (int64_t)t * SP;
                const uint32_t* br = bbase + (int64_t)t * SP;
#pragma unroll
                for (int k = 0; k < NH; ++k) {
                    const int i = h * NH + k;
                    a[k] = ar[ulane + (uint32_t)woff(i)];
                    bb[k] = br[boff[i]];
                }
            };
            float blank_acc = 0.f, pall = 0.f, gfrac = 0.f;
            int gint = 0;
            const bool odd = lane & 1;   // label states: the state's parity is the lane's
            // The posterior 2^gexp * alpha_hat * beta_hat / y of a state in 32-bit arithmetic on the packed words:
            // mantissas as floats in [1, 2) (21 bits + the half-step of lat_unpack), multiplied; exponents as integers;
            // gexp = gint + gfrac split once per frame, the row's log2-prob into floor and fraction exactly; one v_exp
            // of a fraction in (-1, 1) and one v_ldexp.  Denormal and zero states (biased exponent 0) count as zero.
            auto half = [&](int h, const uint32_t (&a)[NH], const uint32_t (&bb)[NH]) {
                float gam[NH];
#pragma unroll
                for (int k = 0; k < NH; ++k) {
                    const int i = h * NH + k;
                    int c = cls[i];
                    asm volatile("" : "+v"(c));   // addresses derived from the class are formed here, not kept per state across the frame loop
                    const float lrv = __shfl(pall, c);
                    const float li = floorf(lrv);
                    const float sfr = gfrac - (lrv - li);
                    const uint32_t wa = a[k], wb = bb[k];
                    const float ma = __uint_as_float(0x3F800002u | ((wa & 0x1FFFFFu) << 2));
                    const float mb = __uint_as_float(0x3F800002u | ((wb & 0x1FFFFFu) << 2));
                    const int ea = (int)(wa >> 21), eb = (int)(wb >> 21);
                    const int e = min(max(ea + eb + (gint - 2046 - (int)li), -300), 300);
                    float g_ = ldexpf(ma * mb * __builtin_amdgcn_exp2f(sfr), e);
                    if (ea == 0 || eb == 0 || !((valid >> i) & 1u)) g_ = 0.f;
                    gam[k] = g_;
                }
                if (odd) {
#pragma unroll
                    for (int k = 0; k < NH; ++k) {
                        const int i = h * NH + k;
                        if (gam[k] != 0.f) atomicAdd(&acc[cls[i]], gam[k]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NH; ++k) blank_acc += gam[k];
                }
            };
            issue(t0, 0, a0, b0);
            float palln = (rbase + (int64_t)t0 * RW)[ulc];
#pragma unroll 1
            for (int t = t0; t < te; ++t) {
                issue(t, 1, a1, b1);
                pall = palln;
                palln = (rbase + (int64_t)min(t + 1, te - 1) * RW)[ulc];
                // 2^gexp * alpha_hat * beta_hat / y = the state posterior (Ecum of the two rescale blocks, see ctc_grad_kernel)
                const int ka = (t + 1) / kRescale - 1, kb = (Tb - t) / kRescale - 1;
                double gexp = gbase;
                if (ka >= 0) gexp += aux[(int64_t)b * aux_stride + kAuxDummy + ka];
                if (kb >= 0) gexp += aux[((int64_t)B + b) * aux_stride + kAuxDummy + kb];
                const double gfl = floor(gexp);
                gint = (int)gfl;
                gfrac = (float)(gexp - gfl);
                __builtin_amdgcn_wave_barrier();   // the previous frame's reads of acc[] are done
                acc[lane] = 0.f;
                __builtin_amdgcn_wave_barrier();
                blank_acc = 0.f;
                half(0, a0, b0);
                if (t + 1 < te) issue(t + 1, 0, a0, b0);
                half(1, a1, b1);
                blank_acc = wave_sum_dpp(blank_acc);
                if (lane == 0) atomicAdd(&acc[blank], blank_acc);   // a label may (wrongly) name the blank class
                __builtin_amdgcn_wave_barrier();
                if (lane < V) Elem<G>::st(grad + (int64_t)t * st + b * sb + lane, (exp2f(pall) - acc[lane]) * scale);
            }
        }
        // rows beyond the utterance's length are zeros
        for (int t = max(t0, Tb); t < min(t0 + kCharFrames, T); ++t)
            if (lane < V) Elem<G>::st(grad + (int64_t)t * st + b * sb + lane, 0.f);
    }
}

template <int W, int NS>
static void launch_lattice(const CtcDims& d, float* ws, const int64_t* labels, int label_stride,
                           const int32_t* in_len, const int32_t* lab_len, int blank, float* nll,
                           hipStream_t s) {
    double* dbl = reinterpret_cast<double*>(ws + d.off_dbl);
    int TB = (kPF * 64 * W * 4) / d.RW;   // RW <= 1024 and RW <= 128 W (see ctc_dims): TB >= 8
    if (TB > 64) TB = 64;
    TB &= ~(kRescale - 1);
    if (TB < kRescale) TB = kRescale;
    hipLaunchKernelGGL((ctc_lattice_kernel<W, NS>), dim3(d.B, 2), dim3(64 * W), 0, s,
                       ws + d.off_rows, labels, label_stride, in_len, lab_len, d.T, d.V, blank,
                       d.Lmax, d.RW, d.SP, d.modeB, TB, dbl, reinterpret_cast<double*>(ws + d.off_beta),
                       reinterpret_cast<double*>(ws + d.off_nll2), nll);
}

}  // namespace sp

extern "C" size_t sp_ctc_workspace_bytes(int T, int B, int V, int max_label_len) {
    if (T < 0 || B < 0 || V < 1 || max_label_len < 0) return 0;
    return sp::ctc_dims(T, B, V, max_label_len).total_bytes + 256;
}

extern "C" int sp_ctc_fwd_bwd(const float* logits, int64_t stride_t, int64_t stride_b,
                              const int64_t* labels, int label_stride, const int32_t* input_lengths,
                              const int32_t* label_lengths, int T, int B, int V, int blank,
                              int max_label_len, float grad_scale, float* nll, float* loss,
                              void* grad, int grad_dtype, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sp;
    SP_CHECK_ARG(!grad || grad_dtype == SP_F32 || grad_dtype == SP_BF16 || grad_dtype == SP_F16,
                 "sp_ctc_fwd_bwd: grad_dtype %d", grad_dtype);
    SP_CHECK_ARG(T >= 0 && B >= 0 && V >= 1, "sp_ctc_fwd_bwd: bad shape T=%d B=%d V=%d", T, B, V);
    SP_CHECK_ARG(V <= kMaxV, "sp_ctc_fwd_bwd: V=%d exceeds the supported maximum %d", V, kMaxV);
    SP_CHECK_ARG(blank >= 0 && blank < V, "sp_ctc_fwd_bwd: blank %d outside [0,%d)", blank, V);
    SP_CHECK_ARG(max_label_len >= 0 && max_label_len <= kMaxL,
                 "sp_ctc_fwd_bwd: max_label_len=%d outside [0,%d]", max_label_len, kMaxL);
    SP_CHECK_ARG(label_stride >= max_label_len, "sp_ctc_fwd_bwd: label_stride %d < max_label_len %d",
                 label_stride, max_label_len);
    if (B == 0) return SP_OK;
    SP_CHECK_ARG(logits && input_lengths && label_lengths && nll, "sp_ctc_fwd_bwd: null pointer");
    SP_CHECK_ARG(labels || max_label_len == 0, "sp_ctc_fwd_bwd: null labels");
    const CtcDims d = ctc_dims(T, B, V, max_label_len);
    if (workspace_bytes < d.total_bytes || !workspace)
        return fail(SP_EWORKSPACE, "sp_ctc_fwd_bwd: workspace %zu < required %zu", workspace_bytes,
                    d.total_bytes);
    SP_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "sp_ctc_fwd_bwd: workspace must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    const double* dbl = reinterpret_cast<const double*>(ws + d.off_dbl);
    const double* dblb = reinterpret_cast<const double*>(ws + d.off_beta);
    const double* dbln2 = reinterpret_cast<const double*>(ws + d.off_nll2);
    const int64_t nrows = (int64_t)T * B;
    const int row_blocks = (int)((nrows + 3) / 4 > 8192 ? 8192 : (nrows + 3) / 4);

    if (T > 0) {
        const bool vec = (V % 4) == 0 && (stride_t % 4) == 0 && (stride_b % 4) == 0 &&
                         ((uintptr_t)logits & 15) == 0 && V <= 2048;
#define SP_ROWS_ARGS logits, stride_t, stride_b, labels, label_stride, input_lengths, label_lengths, T, B, V, \
                     blank, max_label_len, d.RW, d.modeB, ws + d.off_lse, ws + d.off_rows
        const int small_blocks = (int)std::min<int64_t>(((int64_t)T * B + 31) / 32, 16384);
        if (V <= 32) hipLaunchKernelGGL((ctc_rows_small_kernel<32>), dim3(small_blocks), dim3(256), 0, s, SP_ROWS_ARGS);
        else if (V <= 64) hipLaunchKernelGGL((ctc_rows_small_kernel<64>), dim3(small_blocks), dim3(256), 0, s, SP_ROWS_ARGS);
        else if (vec && V <= 256) hipLaunchKernelGGL((ctc_rows_vec_kernel<1>), dim3(row_blocks), dim3(256), 0, s, SP_ROWS_ARGS);
        else if (vec && V <= 512) hipLaunchKernelGGL((ctc_rows_vec_kernel<2>), dim3(row_blocks), dim3(256), 0, s, SP_ROWS_ARGS);
        else if (vec && V <= 1024) hipLaunchKernelGGL((ctc_rows_vec_kernel<4>), dim3(row_blocks), dim3(256), 0, s, SP_ROWS_ARGS);
        else if (vec) hipLaunchKernelGGL((ctc_rows_vec_kernel<8>), dim3(row_blocks), dim3(256), 0, s, SP_ROWS_ARGS);
        else hipLaunchKernelGGL(ctc_rows_kernel, dim3(row_blocks), dim3(256), 0, s, SP_ROWS_ARGS);
#undef SP_ROWS_ARGS
        int rc = check_launch("ctc_rows_kernel");
        if (rc) return rc;
    }
    const int S = 2 * max_label_len + 1;
    // (waves, states per thread): smallest lattice that holds S states
    double* dblw = reinterpret_cast<double*>(ws + d.off_dbl);
    // single-wave lattices carry the dense half of the gradient in the same launch: workgroups behind the 2B lattice
    // workgroups, one per CU the lattice leaves free (rows are dealt statically, so a workgroup that had to wait for
    // a CU would finish last); the sparse half follows.  Only while the lattice leaves at least half of the chip:
    // larger batches run the lattice two workgroups to a CU and the fused gradient kernel on the whole chip after it.
    const bool have_grad = grad && T > 0;
    // Small vocabularies with class-indexed rows (the reference's 28 characters + blank): the sparse-half kernel writes
    // whole gradient rows (kDense), at any batch size and any single-wave lattice; no dense half, no fused kernel.
    const bool small = have_grad && d.small && S <= kSingleWaveMaxS;
    const bool split = have_grad && d.split_ok;
    // How many: the dense half should END with the lattice, not race it -- with all free CUs streaming, the recursion
    // lost 10 % (32 x 751: 41.8 us with 96 dense workgroups, 43.8 with 160, 45.5 with 192) although no streaming wave
    // shares its CUs.  A CU streams ~32 GB/s of rows that the Infinity Cache still holds and ~22 GB/s from HBM; the
    // lattice takes 52 / 76 / 125 ns per frame at 2 / 4 / 8 states per lane.  Only the four-rows-in-flight path is
    // sized this way: the generic one (odd or large V) is latency-bound per wave and takes every free CU.
    int dense_blocks = split ? kCUs - 2 * B : 0;
    if (split && V % 4 == 0 && V <= 1024) {
        const double bytes = (double)T * B * V * (4 + (grad_dtype == SP_F32 ? 4 : 2));
        const double t_lat = (double)T * (S <= 128 ? 52e-9 : S <= 256 ? 76e-9 : 125e-9);
        const double per_cu = bytes > 200e6 ? 22e9 : 32e9;
        dense_blocks = std::min(dense_blocks, std::max(32, (int)(1.3 * bytes / (t_lat * per_cu)) + 1));
    }
#define SP_LAT1_ARGS ws + d.off_rows, labels, label_stride, input_lengths, label_lengths, T, V, blank, max_label_len, d.RW, \
                     d.SP, d.modeB, dblw, reinterpret_cast<double*>(ws + d.off_beta), reinterpret_cast<double*>(ws + d.off_aux), \
                     reinterpret_cast<double*>(ws + d.off_nll2), nll, B, logits, stride_t, stride_b, ws + d.off_lse, grad_scale, grad,  \
                     grad_dtype == SP_BF16 ? 1 : (grad_dtype == SP_F16 ? 2 : 0)
#define SP_LAT1(NS, H)                                                                                                       \
    hipLaunchKernelGGL((ctc_lattice1_kernel<NS, H>), dim3(2 * B + dense_blocks), dim3(split ? kLat1Threads : 64 * (1 + H)),  \
                       split ? kLat1DynLds : 0, s, SP_LAT1_ARGS)
    if (S <= 128) {
        // four helper waves when the workgroup has a CU to itself (two could not keep up with the recursion: 43.4 ->
        // 39.7 us at 32 x 751 x 1000); two where two workgroups share a CU's registers
        if (split || (small && 2 * B <= kCUs)) SP_LAT1(2, 4);
        else SP_LAT1(2, 2);
    }
    else if (S <= 256) SP_LAT1(4, 4);
    else if (S <= 512) SP_LAT1(8, 4);   // 15 s of characters (L <= 255): eight states per lane, still no barrier per frame
    else if (S <= kSingleWaveMaxS) SP_LAT1(16, 4);   // 30 s of characters (L <= 511): sixteen states per lane
    else launch_lattice<4, 8>(d, ws, labels, label_stride, input_lengths, label_lengths, blank, nll, s);
#undef SP_LAT1
#undef SP_LAT1_ARGS
    int rc = check_launch("ctc_lattice_kernel");
    if (rc) return rc;

    if (grad || loss) {
        const int gb = have_grad ? row_blocks : 1;
        const size_t lds = (size_t)4 * V * sizeof(float);
        const double* auxp = S <= kSingleWaveMaxS ? reinterpret_cast<const double*>(ws + d.off_aux) : nullptr;
        if (split || small) {
#define SP_GRAD_ARGS stride_t, stride_b, labels, label_stride, input_lengths, label_lengths, T, B, V, blank, max_label_len,  \
                     d.RW, d.SP, d.modeB, ws + d.off_rows, dbl, dblb, auxp, dbln2, nll, grad_scale
#define SP_SPARSE(NI, R)                                                                                                     \
    do {                                                                                                                     \
        const int fpi = R * kSparseGroups<NI, DENSE>;                                                                         \
        const int64_t items = (int64_t)B * ((T + fpi - 1) / fpi);                                                            \
        const int sg = (int)std::min<int64_t>((items + 3) / 4, 8192);                                                        \
        const size_t sl = (size_t)4 * (DENSE ? R * V : V + R * (max_label_len + 2)) * sizeof(float);                         \
        if (grad_dtype == SP_BF16)                                                                                           \
            hipLaunchKernelGGL((ctc_sparse_kernel<__hip_bfloat16, NI, R, DENSE>), dim3(sg), dim3(256), sl, s, SP_GRAD_ARGS,  \
                               (__hip_bfloat16*)grad, loss, d.perm);                                                         \
        else if (grad_dtype == SP_F16)                                                                                       \
            hipLaunchKernelGGL((ctc_sparse_kernel<__half, NI, R, DENSE>), dim3(sg), dim3(256), sl, s, SP_GRAD_ARGS,          \
                               (__half*)grad, loss, d.perm);                                                                 \
        else                                                                                                                 \
            hipLaunchKernelGGL((ctc_sparse_kernel<float, NI, R, DENSE>), dim3(sg), dim3(256), sl, s, SP_GRAD_ARGS,           \
                               (float*)grad, loss, d.perm);                                                                  \
    } while (0)
#define SP_CHAR(NI, PS)                                                                                                      \
    do {                                                                                                                     \
        const int64_t items = (int64_t)B * ((T + kCharFrames - 1) / kCharFrames);                                            \
        const int sg = (int)std::min<int64_t>((items + 3) / 4, 16384);                                                       \
        const uint32_t* al = reinterpret_cast<const uint32_t*>(dbl);                                                         \
        const uint32_t* be = reinterpret_cast<const uint32_t*>(dblb);                                                        \
        if (grad_dtype == SP_BF16)                                                                                           \
            hipLaunchKernelGGL((ctc_char_grad_kernel<__hip_bfloat16, NI, PS>), dim3(sg), dim3(256), 0, s, SP_CHAR_ARGS,          \
                               (__hip_bfloat16*)grad, loss, d.perm);                                                         \
        else if (grad_dtype == SP_F16)                                                                                       \
            hipLaunchKernelGGL((ctc_char_grad_kernel<__half, NI, PS>), dim3(sg), dim3(256), 0, s, SP_CHAR_ARGS, (__half*)grad,   \
                               loss, d.perm);                                                                                \
        else                                                                                                                 \
            hipLaunchKernelGGL((ctc_char_grad_kernel<float, NI, PS>), dim3(sg), dim3(256), 0, s, SP_CHAR_ARGS, (float*)grad,     \
                               loss, d.perm);                                                                                \
    } while (0)
#define SP_CHAR_ARGS stride_t, stride_b, labels, label_stride, input_lengths, label_lengths, T, B, V, blank, max_label_len,  \
                     d.RW, d.SP, ws + d.off_rows, al, be, auxp, dbln2, nll, grad_scale
            if (small) {
#define DENSE true
                if (S <= 128) SP_SPARSE(2, 2);
                else if (S <= 256) SP_SPARSE(4, 2);
                else if (S <= 512) SP_CHAR(8, 8);
                else SP_CHAR(16, 16);
#undef DENSE
            } else {
#define DENSE false
                if (S <= 128) SP_SPARSE(2, 2);   // measured at 32 x 751 x 1000: 19.9 / 18.5 / 21.4 us at 1 / 2 / 4 frames per wave
                else if (S <= 256) SP_SPARSE(4, 2);
                else SP_SPARSE(8, 1);   // registers: one frame per wave at eight states per lane (47 against 60 us)
#undef DENSE
            }
#undef SP_SPARSE
#undef SP_CHAR
#undef SP_CHAR_ARGS
#undef SP_GRAD_ARGS
        } else {
#define SP_GRAD_ARGS logits, stride_t, stride_b, labels, label_stride, input_lengths, label_lengths, T, B, V, blank,   \
                     max_label_len, d.RW, d.SP, d.modeB, ws + d.off_lse, ws + d.off_rows, dbl, dblb, auxp, dbln2, nll, grad_scale
            if (grad && grad_dtype == SP_BF16)
                hipLaunchKernelGGL((ctc_grad_kernel<__hip_bfloat16>), dim3(gb), dim3(256), lds, s, SP_GRAD_ARGS,
                                   T > 0 ? (__hip_bfloat16*)grad : nullptr, loss, d.perm);
            else if (grad && grad_dtype == SP_F16)
                hipLaunchKernelGGL((ctc_grad_kernel<__half>), dim3(gb), dim3(256), lds, s, SP_GRAD_ARGS,
                                   T > 0 ? (__half*)grad : nullptr, loss, d.perm);
            else
                hipLaunchKernelGGL((ctc_grad_kernel<float>), dim3(gb), dim3(256), lds, s, SP_GRAD_ARGS,
                                   T > 0 ? (float*)grad : nullptr, loss, d.perm);
#undef SP_GRAD_ARGS
        }
        rc = check_launch("ctc_grad_kernel");
        if (rc) return rc;
    }
    return SP_OK;
}

extern "C" size_t sp_ctc_noalloc_max_bytes(void) { return sp::kCtcScratchBytes; }

extern "C" int sp_ctc_fwd_bwd_noalloc(const float* logits, int64_t stride_t, int64_t stride_b,
                                      const int64_t* labels, int label_stride, const int32_t* input_lengths,
                                      const int32_t* label_lengths, int T, int B, int V, int blank,
                                      int max_label_len, float grad_scale, float* nll, float* loss,
                                      void* grad, int grad_dtype, void* stream) {
    using namespace sp;
    // set-up, once per device, with that device current: each GPU holds its own copy of `scratch` at its
    // own address
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(SP_ELAUNCH, "sp_ctc_fwd_bwd_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    SP_CHECK_ARG(dev >= 0 && dev < kMaxDevices, "sp_ctc_fwd_bwd_noalloc: device ordinal %d not supported", dev);
    void* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return fail(SP_ELAUNCH, "sp_ctc_fwd_bwd_noalloc: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);
    }
    if (T >= 0 && B >= 0 && V >= 1 && max_label_len >= 0 && max_label_len <= kMaxL) {
        const size_t need = ctc_dims(T, B, V, max_label_len).total_bytes;
        if (need > kCtcScratchBytes)
            return fail(SP_EWORKSPACE,
                        "sp_ctc_fwd_bwd_noalloc: T=%d B=%d V=%d max_label_len=%d needs %zu bytes of scratch, the "
                        "built-in buffer holds %zu (use sp_ctc_fwd_bwd with a caller-owned workspace)",
                        T, B, V, max_label_len, need, kCtcScratchBytes);
    }
    // the kernels overwrite every workspace word they later read, so nothing has to start at zero
    return sp_ctc_fwd_bwd(logits, stride_t, stride_b, labels, label_stride, input_lengths, label_lengths, T, B, V,
                          blank, max_label_len, grad_scale, nll, loss, grad, grad_dtype, ptr, kCtcScratchBytes, stream);
}
