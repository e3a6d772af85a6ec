// sp_gemm: the one GEMM entry point (include/speech_hip.h).  Which kernel runs is decided here by shape,
// layout and dtype alone:
//   * bf16 / f16, both operands K-contiguous ("NT": forward projection on the packed W^T, data gradients,
//     classifier logits), whole 256- or 128-column tiles, K % 64 == 0   -> gemm_rs.hip, NT role-split kernel
//   * bf16, both operands row-contiguous ("TN": weight gradients), f32 split-K slabs, 256 x 256 tiles
//                                                                       -> gemm_rs.hip, TN role-split kernel
//   * everything else (fp32, fp16, odd or small shapes, NN / TT forms)   -> gemm_generic.hip
#include "gemm.h"

namespace sp {
constexpr int kMinFastTiles = 64;     // below this many tiles a persistent launch has nothing to stream
}

extern "C" int sp_gemm_tn_splits(int M, int N, int K, int in_dtype) {
    using namespace sp;
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    if (in_dtype == SP_BF16 && M % 256 == 0 && N % 256 == 0 && K >= 2048) return tn_rs_splits(M, N, K);
    return K >= 8192 ? 8 : (K >= 2048 ? 4 : 1);
}

extern "C" int sp_gemm(const void* A, const void* B, void* C, int M, int N, int K, int64_t sam, int64_t sak,
                       int64_t sbn, int64_t sbk, int64_t ldc, int in_dtype, int out_dtype, int accumulate,
                       int split_k, int64_t c_split_stride, void* stream) {
    using namespace sp;
    SP_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "sp_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    SP_CHECK_ARG(in_dtype == SP_F32 || in_dtype == SP_F16 || in_dtype == SP_BF16, "sp_gemm: in_dtype %d", in_dtype);
    SP_CHECK_ARG(out_dtype == SP_F32 || out_dtype == SP_F16 || out_dtype == SP_BF16, "sp_gemm: out_dtype %d", out_dtype);
    SP_CHECK_ARG(sak == 1 || sam == 1, "sp_gemm: A must be contiguous along k (sak == 1) or along m (sam == 1)");
    SP_CHECK_ARG(sbk == 1 || sbn == 1, "sp_gemm: B must be contiguous along k (sbk == 1) or along n (sbn == 1)");
    SP_CHECK_ARG(!(sak == 1 && sam == 1 && M > 1 && K > 1) && !(sbk == 1 && sbn == 1 && N > 1 && K > 1),
                 "sp_gemm: overlapping operand strides");
    SP_CHECK_ARG(ldc >= N, "sp_gemm: ldc %lld < N %d", (long long)ldc, N);
    SP_CHECK_ARG(split_k >= 1 && split_k <= 64, "sp_gemm: split_k %d outside [1, 64]", split_k);
    SP_CHECK_ARG(!accumulate || out_dtype == SP_F32, "sp_gemm: accumulation needs an f32 output");
    SP_CHECK_ARG(split_k == 1 || c_split_stride >= (int64_t)M * ldc,
                 "sp_gemm: split-K slabs must not overlap (c_split_stride >= M * ldc)");
    if (M == 0 || N == 0) return SP_OK;
    SP_CHECK_ARG(C && (K == 0 || (A && B)), "sp_gemm: null pointer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int vec = in_dtype == SP_F32 ? 4 : 8;
    const int64_t a_ld = sak == 1 ? sam : sak, b_ld = sbk == 1 ? sbn : sbk;
    const bool aligned_in = (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && a_ld % vec == 0 && b_ld % vec == 0;
    if ((in_dtype == SP_BF16 || in_dtype == SP_F16) && !accumulate && aligned_in && K >= 64) {
        // NT: both operands K-contiguous; the output keeps the operands' 16-bit type or is f32
        if (sak == 1 && sbk == 1 && split_k == 1 && (M > 1 || sam >= K) && (out_dtype == in_dtype || out_dtype == SP_F32)) {
            const int bn = N % 256 == 0 ? 256 : 128, bm = bn == 256 ? 192 : 256;
            if ((int64_t)((M + bm - 1) / bm) * (N / bn) >= kMinFastTiles &&
                nt_rs_ok(M, N, K, sam, sbn, ldc, A, B, C, out_dtype, bn))
                return gemm_nt_rs(A, B, C, M, N, K, sam, sbn, ldc, in_dtype, out_dtype, bn == 256 ? 0 : 3, s);
        }
        // TN: both operands row-contiguous, f32 slabs (bf16: the training type)
        if (in_dtype == SP_BF16 && sam == 1 && sbn == 1 && sak > 1 && sbk > 1 && out_dtype == SP_F32 && ldc % 4 == 0 &&
            c_split_stride % 4 == 0 && (int64_t)(M / 256) * (N / 256) * split_k >= kMinFastTiles &&
            (split_k - 1) * (((K + 63) / 64 + split_k - 1) / split_k) < (K + 63) / 64 &&      // no empty K range
            tn_rs_ok(M, N, K, sak, sbk, A, B, C))
            return gemm_tn_rs(A, B, (float*)C, M, N, K, sak, sbk, ldc, split_k > 1 ? c_split_stride : 0, split_k, 0, s);
    }
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
    g.sam = sam; g.sak = sak; g.sbn = sbn; g.sbk = sbk; g.ldc = ldc;
    g.out_dtype = out_dtype; g.accumulate = accumulate; g.split_k = split_k; g.c_split_stride = c_split_stride;
    g.vec_in = aligned_in;
    SP_CHECK_ARG((int64_t)((M + 127) / 128) < 65536, "sp_gemm: M too large for the general kernel");
    return gemm_generic(g, in_dtype, s);
}
