// Argument block shared by the GEMM kernels of gemm.hip (128x128 tile, every dtype / alignment) and
// gemm_wide.hip (256x256 tile, bf16, LDS-DMA staged: the gene-width layers).
#pragma once
#include "common.h"

namespace c2 {

// C[M,N] = alpha * sum_r Aop[i][r] * Bop[j][r] (+ bias[j]).  Each operand is "RC" (reduction index contiguous:
// element (x,r) at p[x*ld + r]) or "XC" (output index contiguous: element (x,r) at p[r*ld + x]).
// layout 0 = (RC,RC) forward, 1 = (RC,XC) dgrad, 2 = (XC,XC) wgrad.
struct GemmArgs {
    const void* A;
    const void* B;
    float* C;
    const float* bias;
    const float* alpha_dev;
    int64_t lda, ldb, ldc, slab_stride;
    int M, N, R, r_per_split;
};

// Tile edge the library will use for a bf16 GEMM of this shape (256 when gemm_wide takes it, else 128).
bool wide_eligible(int layout, int dtype, const GemmArgs& a);
int launch_gemm_wide(int layout, const GemmArgs& a, int nslab, hipStream_t stream);

}  // namespace c2
