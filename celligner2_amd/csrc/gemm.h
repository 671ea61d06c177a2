// Argument block shared by the GEMM kernels of gemm.hip (128x128 tile, every dtype / alignment) and
// gemm_wide.hip (256x256 tile, bf16, LDS-DMA staged: the gene-width layers).
#pragma once
#include "common.h"

namespace c2 {

// C[M,N] = alpha * sum_r Aop[i][r] * Bop[j][r] (+ bias[j]).  Each operand is "RC" (reduction index contiguous:
// element (x,r) at p[x*ld + r]) or "XC" (output index contiguous: element (x,r) at p[r*ld + x]).
// layout 0 = (RC,RC) forward, 1 = (RC,XC) dgrad, 2 = (XC,XC) wgrad, 3 = (XC,RC) wgrad with the second operand transposed
// (256x256 LDS-DMA kernel only); row-indexed launches also take 4 = (RC, XC with indexed reduction rows).
struct GemmArgs {
    const void* A;
    const void* B;
    float* C;
    const float* bias;
    const float* alpha_dev;
    int64_t lda, ldb, ldc, slab_stride;
    int M, N, R, r_per_split;
    const void* A_lo;  // C2_BF16X3P: the low bf16 planes of the two operands (same layout / leading dimension)
    const void* B_lo;
    // Row-indexed operand (a batch read in place from a resident matrix, no gathered copy):
    //   layout 0 (forward): output row m reads row rows[m] of A;  layout 2 (wgrad): reduction index r reads row rows[r]
    //   of B.  int32, readable up to the reduction length rounded up to 64 (layout 2).  NULL: plain operands.
    const int32_t* rows;
};

// Block-tile edge gemm_wide.hip uses for this problem (256 or 128), or 0 when it does not take it.
int wide_tile(int dtype, const GemmArgs& a);
int launch_gemm_wide(int layout, int tile, const GemmArgs& a, int nslab, hipStream_t stream);
// Row-indexed instances (256x256 tile, bf16, layouts 0 and 2 only); 0 = launched, 3 = shape not supported.
int launch_gemm_wide_rows(int layout, const GemmArgs& a, int nslab, hipStream_t stream);

// The same kernels on IEEE half operands (gemm_wide_f16.hip; dtype C2_F16).
int wide_tile_f16(int dtype, const GemmArgs& a);
int launch_gemm_wide_f16(int layout, int tile, const GemmArgs& a, int nslab, hipStream_t stream);
int launch_gemm_wide_rows_f16(int layout, const GemmArgs& a, int nslab, hipStream_t stream);

}  // namespace c2
