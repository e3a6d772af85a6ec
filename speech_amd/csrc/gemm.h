// Shared between gemm.hip (the sp_gemm entry point), gemm_rs.hip (persistent role-split kernels) and
// gemm_generic.hip (the general kernel).
#pragma once
#include "common.h"

namespace sp {

struct GemmArgs {
    const void* A; const void* B; void* C;
    int M, N, K;
    int64_t sam, sak, sbn, sbk, ldc;
    int out_dtype, accumulate, split_k;
    int64_t c_split_stride;
    int vec_in;      // 16-byte operand loads are legal (aligned bases, non-unit strides multiples of a vector)
};

int gemm_generic(const GemmArgs& g, int in_dtype, hipStream_t s);

// role-split persistent kernels (bf16 operands)
bool nt_rs_ok(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, const void* A, const void* B, const void* C,
              int out_dtype, int BN);
int gemm_nt_rs(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
               int in_dtype, int out_dtype, int variant, hipStream_t s);
bool tn_rs_ok(int Mi, int Nj, int K, int64_t ldx, int64_t ldg, const void* X, const void* G, const void* P);
int tn_rs_splits(int Mi, int Nj, int K);
int gemm_tn_rs(const void* X, const void* G, float* P, int Mi, int Nj, int K, int64_t ldx, int64_t ldg, int64_t ldp,
               int64_t slab_stride, int S, int variant, hipStream_t s);

}  // namespace sp
