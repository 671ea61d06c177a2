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
    // Output as IEEE halves instead of fp32 (C then addresses [M][ldc] halves; unsplit launches only): the decoder's
    // logits in bf16 mode -- the 311 MB fp32 store of the 4096 x 19 000 forward becomes 155 MB, and the loss kernel reads
    // 155 MB less.
    int c_f16;
    // Rank-one term of the epilogue: C[row][col] += rowscale[row] * bias[col] (the first split of a split reduction adds
    // it, once).  The first layer's weight gradient with a centred input: dW = dh^T (x - c) + (sum_b dh_b) (x) c.
    const float* rowscale;
};

// One 4-column piece of an output row: alpha * v + bias, as fp32 or halves.  dst_row: the row's first element (fp32 view of
// C); col: first of the four columns; vec: the 4 columns exist and the address is aligned for one store.
__device__ __forceinline__ void store_out4(const GemmArgs& a, float* C, int64_t row, int col, const f32x4& v, float alpha,
                                           const float (&bv0)[4], bool vec, bool first_split = true) {
    float bv[4] = {bv0[0], bv0[1], bv0[2], bv0[3]};
    if (a.rowscale) {
        const float rs = first_split ? a.rowscale[row] : 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c] *= rs;
    }
    if (a.c_f16) {
        f16_t* dst = reinterpret_cast<f16_t*>(C) + row * a.ldc + col;
        if (vec) {
            u32x2 pk;
            pk[0] = pack2<f16_t>(alpha * v[0] + bv[0], alpha * v[1] + bv[1]);
            pk[1] = pack2<f16_t>(alpha * v[2] + bv[2], alpha * v[3] + bv[3]);
            *reinterpret_cast<u32x2*>(dst) = pk;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (col + c < a.N) dst[c] = (f16_t)(alpha * v[c] + bv[c]);
        }
        return;
    }
    float* dst = C + row * a.ldc + col;
    if (vec) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = alpha * v[c] + bv[c];
        *reinterpret_cast<f32x4*>(dst) = o;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (col + c < a.N) dst[c] = alpha * v[c] + bv[c];
    }
}

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
