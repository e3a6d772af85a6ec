// Fused log-softmax + CTC loss + gradient for gfx950.
// Replaces train.py:69-77 of the reference (F.log_softmax -> transpose -> torch.nn.CTCLoss).
//
// Three launches on one stream, all arithmetic in the log2 domain (v_exp_f32/v_log_f32 are
// base-2 natively):
//   K1 ctc_rows    (whole chip, one wave per (t,b) row): lse2 = log2 sum_c 2^(z_c*log2e) and the
//                  compact per-frame row the lattice needs -- either the normalised row itself
//                  (small vocabulary, V <= Lmax+1: indexed by class) or the gathered
//                  (blank, l_1..l_L) log-probs (large vocabulary: indexed by label position).
//   K2 ctc_lattice (2 workgroups per utterance: alpha forward in time, beta backward in time,
//                  running concurrently).  Each thread owns NS consecutive lattice states in
//                  registers; neighbours come from the adjacent lane by DPP wave shift, and only
//                  the wave-boundary pair crosses LDS (one barrier per frame, none for a
//                  single-wave lattice).  The per-frame rows are staged through LDS a chunk of
//                  frames ahead (issue loads early, write to LDS late).  beta is the same
//                  recursion on the reversed label sequence over reversed time, so there is one
//                  code path.  Lattice rows are spilled to the workspace in coalesced rows.
//   K3 ctc_grad    (whole chip, one wave per (t,b) row): posterior per state
//                  2^(alpha+beta-lp+nll), summed per class in LDS, grad = (softmax - occ)*scale.
//
// Workspace: lse2[B*T] | rows[B][T][RW] (f32), then f64 alpha[B][T][SP] | beta_rev[B][T][SP] | nll2[B].
//
// Precision: alpha/beta reach thousands of log2 units, where one f32 ulp is ~2e-4 and a plain
// f32 log-space recursion (torch's included) random-walks to ~3e-4 relative error in the
// posteriors.  K2 keeps the lattice states in f64 registers (the adds; exp2/log2 of the small
// terms stay f32) and spills the rows as f64: for tightly constrained alignments the states that
// carry the posterior sit hundreds of units below the row maximum, so an f32 spill relative to
// any per-row offset loses the same digits again.  K3 forms alpha + beta - lp - log-likelihood
// in f64 and exponentiates in f32.
#include "common.h"

namespace sp {

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kMaxV = 4000;  // K3 keeps 4 x V floats of per-class sums in LDS (<= 64 KB)
constexpr int kMaxL = 1023;
constexpr int kPF = 8;  // float4 prefetch registers per thread in K2

struct CtcDims {
    int T, B, V, Lmax;
    int RW;    // row width in floats (multiple of 4)
    int SP;    // padded lattice row (multiple of 8)
    int modeB; // 1: rows indexed by class, 0: by label position
    size_t off_lse, off_rows, total_floats;  // float offsets
    size_t off_dbl;  // float offset of the f64 block: alpha[B][T][SP] | beta_rev[B][T][SP] | nll2[B]
    size_t total_bytes;
};

static CtcDims ctc_dims(int T, int B, int V, int Lmax) {
    CtcDims d;
    d.T = T; d.B = B; d.V = V; d.Lmax = Lmax;
    d.modeB = V <= Lmax + 1;
    d.RW = (int)align_up((size_t)(d.modeB ? V : Lmax + 1), 4);
    d.SP = (int)align_up((size_t)(2 * Lmax + 1), 8);
    size_t o = 0;
    d.off_lse = o; o += align_up((size_t)B * T, 4);
    d.off_rows = o; o += (size_t)B * T * d.RW;
    o = align_up(o, 4);
    d.off_dbl = o;
    o += 2 * ((size_t)2 * B * T * d.SP + B);  // doubles, counted in floats
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

// -------------------------------------------------------------------------------------------
// K2: alpha / beta lattice
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_shr1_i(int v, int fill) {
    // lane i receives lane i-1's value; lane 0 receives `fill`
    return __builtin_amdgcn_update_dpp(fill, v, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
}
__device__ __forceinline__ double lane_shr1(double v) {
    // -inf as f64 = 0xFFF00000'00000000
    const int lo = lane_shr1_i(__double2loint(v), 0);
    const int hi = lane_shr1_i(__double2hiint(v), (int)0xFFF00000);
    return __hiloint2double(hi, lo);
}

// log2(2^a + 2^b + 2^c): differences and the final add in f64 (alpha reaches thousands of log2
// units; an f32 add there costs 2e-4 per step), exp2/log2 of the small terms in f32.
__device__ __forceinline__ double lse2_3(double a, double b, double c) {
    const double m = fmax(a, fmax(b, c));
    const double m2 = (m == -(double)INFINITY) ? 0.0 : m;
    const float s = exp2f((float)(a - m2)) + exp2f((float)(b - m2)) + exp2f((float)(c - m2));
    return m2 + (double)log2f(s);
}
__device__ __forceinline__ double lse2_2(double a, double b) {
    const double m = fmax(a, b);
    const double m2 = (m == -(double)INFINITY) ? 0.0 : m;
    const float s = exp2f((float)(a - m2)) + exp2f((float)(b - m2));
    return m2 + (double)log2f(s);
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
    __shared__ double s_bnd[2][W][2];  // per wave: last state, second-to-last
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
    int ridx[NS];
    bool valid[NS], skip[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int sp_ = tid * NS + j;           // position in this recursion's order
        const int s = dir ? S - 1 - sp_ : sp_;   // real lattice state
        valid[j] = sp_ < S;
        ridx[j] = 0;
        skip[j] = false;
        if (valid[j]) {
            int cls = blank;
            if (s & 1) cls = min(max((int)lab[(s - 1) >> 1], 0), V - 1);
            ridx[j] = modeB ? cls : ((s & 1) ? (s + 1) >> 1 : 0);
            if ((s & 1) && sp_ >= 2) {
                const int s2 = dir ? s + 2 : s - 2;  // the state two steps earlier in this order
                const int cls2 = min(max((int)lab[(s2 - 1) >> 1], 0), V - 1);
                skip[j] = cls != cls2;
            }
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

    const double kNegInf = -(double)INFINITY;
    double a[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) a[j] = kNegInf;

    int i = 0;
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) SP_CTC_ISSUE(c + 1);
        const float* buf = reinterpret_cast<const float*>(s_chunk[c & 1]);
        const int lo = chunk_lo(c), hi = chunk_hi(c);
        for (int k = 0; k < hi - lo; ++k, ++i) {
            const int t = dir ? Tb - 1 - i : i;
            const float* row = buf + (t - lo) * RW;
            float lp[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) lp[j] = row[ridx[j]];
            if (i == 0) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const int sp_ = tid * NS + j;
                    a[j] = (valid[j] && sp_ < 2) ? (double)lp[j] : kNegInf;
                }
            } else {
                double p1 = lane_shr1(a[NS - 1]);
                double p2 = lane_shr1(a[NS - 2]);
                if (W > 1) {
                    if (lane == 0 && wave > 0) {
                        p1 = s_bnd[i & 1][wave - 1][0];
                        p2 = s_bnd[i & 1][wave - 1][1];
                    }
                }
                double n_[NS];
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const double x1 = (j >= 1) ? a[j - 1] : p1;
                    const double x2 = (j >= 2) ? a[j - 2] : (j == 1 ? p1 : p2);
                    // j == 1: s'-2 is the previous lane's last state (p1); j == 0: its second-to-last
                    const double x2s = skip[j] ? x2 : kNegInf;
                    n_[j] = valid[j] ? lse2_3(a[j], x1, x2s) + (double)lp[j] : kNegInf;
                }
#pragma unroll
                for (int j = 0; j < NS; ++j) a[j] = n_[j];
            }
            // spill the lattice row (coalesced 16-byte stores)
            double* orow = out + (int64_t)t * SP + tid * NS;
#pragma unroll
            for (int j = 0; j < NS; j += 2)
                if (tid * NS + j < SP) *reinterpret_cast<double2*>(orow + j) = make_double2(a[j], a[j + 1]);
            if (W > 1) {
                if (lane == 63) {
                    s_bnd[(i + 1) & 1][wave][0] = a[NS - 1];
                    s_bnd[(i + 1) & 1][wave][1] = a[NS - 2];
                }
                __syncthreads();
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
            const double ll2 = (S > 1) ? lse2_2(s_fin[0], s_fin[1]) : s_fin[0];  // log2 likelihood
            nll2[b] = -ll2;
            nll[b] = (float)(-ll2 * 0.69314718055994530942);
        }
    }
}

// -------------------------------------------------------------------------------------------
// K3: gradient rows (+ the scalar loss in block 0)
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float* __restrict__ logits, int64_t st, int64_t sb, const int64_t* __restrict__ labels,
    int label_stride, const int32_t* __restrict__ in_len, const int32_t* __restrict__ lab_len,
    int T, int B, int V, int blank, int Lmax, int RW, int SP, int modeB,
    const float* __restrict__ lse2, const float* __restrict__ rows, const double* __restrict__ alpha,
    const double* __restrict__ beta_rev, const double* __restrict__ nll2d,
    const float* __restrict__ nll, float grad_scale, float* __restrict__ grad,
    float* __restrict__ loss) {
    extern __shared__ __attribute__((aligned(16))) float s_occ[];  // [waves][V]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;

    if (blockIdx.x == 0 && wave == 0 && loss) {
        float acc = 0.f;
        for (int b = lane; b < B; b += 64) acc += nll[b] / (float)max(lab_len[b], 1);
        acc = wave_sum(acc);
        if (lane == 0) loss[0] = acc / (float)B;
    }
    if (!grad) return;

    float* occ = s_occ + wave * V;
    const int64_t nrows = (int64_t)T * B;
    for (int64_t r = (int64_t)blockIdx.x * wpb + wave; r < nrows; r += (int64_t)gridDim.x * wpb) {
        const int b = (int)(r / T), t = (int)(r % T);
        const int Tb = min(max(in_len[b], 0), T);
        float* g = grad + t * st + b * sb;
        if (t >= Tb) {
            for (int c = lane; c < V; c += 64) g[c] = 0.f;
            continue;
        }
        const int L = min(max(lab_len[b], 0), Lmax);
        const int S = 2 * L + 1;
        const double nll2 = nll2d[b];
        const float scale = grad_scale / ((float)B * (float)max(L, 1));
        const double* arow = alpha + ((int64_t)b * T + t) * SP;
        const double* brow = beta_rev + ((int64_t)b * T + t) * SP;
        const float* lrow = rows + ((int64_t)b * T + t) * RW;
        const int64_t* lab = labels + (int64_t)b * label_stride;

        for (int c = lane; c < V; c += 64) occ[c] = 0.f;
        float blank_acc = 0.f;
        for (int s = lane; s < S; s += 64) {
            int cls = blank;
            if (s & 1) cls = min(max((int)lab[(s - 1) >> 1], 0), V - 1);
            const int ri = modeB ? cls : ((s & 1) ? (s + 1) >> 1 : 0);
            const float gam = exp2f((float)(arow[s] + brow[S - 1 - s] - (double)lrow[ri] + nll2));
            if (s & 1) atomicAdd(&occ[cls], gam);
            else blank_acc += gam;
        }
        blank_acc = wave_sum(blank_acc);
        if (lane == 0) atomicAdd(&occ[blank], blank_acc);
        __builtin_amdgcn_wave_barrier();
        const float* z = logits + t * st + b * sb;
        const float l2 = lse2[(int64_t)b * T + t];
        for (int c = lane; c < V; c += 64) {
            const float sm = exp2f(z[c] * kLog2e - l2);
            g[c] = (sm - occ[c]) * scale;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int W, int NS>
static void launch_lattice(const CtcDims& d, float* ws, const int64_t* labels, int label_stride,
                           const int32_t* in_len, const int32_t* lab_len, int blank, float* nll,
                           hipStream_t s) {
    double* dbl = reinterpret_cast<double*>(ws + d.off_dbl);
    int TB = (kPF * 64 * W * 4) / d.RW;
    if (TB > 64) TB = 64;
    if (TB < 1) TB = 1;
    hipLaunchKernelGGL((ctc_lattice_kernel<W, NS>), dim3(d.B, 2), dim3(64 * W), 0, s,
                       ws + d.off_rows, labels, label_stride, in_len, lab_len, d.T, d.V, blank,
                       d.Lmax, d.RW, d.SP, d.modeB, TB, dbl, dbl + (size_t)d.B * d.T * d.SP,
                       dbl + (size_t)2 * d.B * d.T * d.SP, nll);
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
                              float* grad, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sp;
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
    const int64_t nrows = (int64_t)T * B;
    const int row_blocks = (int)((nrows + 3) / 4 > 4096 ? 4096 : (nrows + 3) / 4);

    if (T > 0) {
        hipLaunchKernelGGL(ctc_rows_kernel, dim3(row_blocks), dim3(256), 0, s, logits, stride_t,
                           stride_b, labels, label_stride, input_lengths, label_lengths, T, B, V,
                           blank, max_label_len, d.RW, d.modeB, ws + d.off_lse, ws + d.off_rows);
        int rc = check_launch("ctc_rows_kernel");
        if (rc) return rc;
    }
    const int S = 2 * max_label_len + 1;
    // (waves, states per thread): smallest lattice that holds S states
    if (S <= 128) launch_lattice<1, 2>(d, ws, labels, label_stride, input_lengths, label_lengths, blank, nll, s);
    else if (S <= 256) launch_lattice<1, 4>(d, ws, labels, label_stride, input_lengths, label_lengths, blank, nll, s);
    else if (S <= 512) launch_lattice<2, 4>(d, ws, labels, label_stride, input_lengths, label_lengths, blank, nll, s);
    else if (S <= 1024) launch_lattice<4, 4>(d, ws, labels, label_stride, input_lengths, label_lengths, blank, nll, s);
    else launch_lattice<4, 8>(d, ws, labels, label_stride, input_lengths, label_lengths, blank, nll, s);
    int rc = check_launch("ctc_lattice_kernel");
    if (rc) return rc;

    if (grad || loss) {
        const int gb = (grad && T > 0) ? row_blocks : 1;
        const size_t lds = (size_t)4 * V * sizeof(float);
        hipLaunchKernelGGL(ctc_grad_kernel, dim3(gb), dim3(256), lds, s, logits, stride_t, stride_b,
                           labels, label_stride, input_lengths, label_lengths, T, B, V, blank,
                           max_label_len, d.RW, d.SP, d.modeB, ws + d.off_lse, ws + d.off_rows,
                           dbl, dbl + (size_t)B * T * d.SP, dbl + (size_t)2 * B * T * d.SP, nll,
                           grad_scale, T > 0 ? grad : nullptr, loss);
        rc = check_launch("ctc_grad_kernel");
        if (rc) return rc;
    }
    return SP_OK;
}
