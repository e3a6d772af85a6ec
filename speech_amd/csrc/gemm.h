// Shared between gemm.hip (fast kernels + the sp_gemm entry point) and gemm_generic.hip.
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

}  // namespace sp
