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
// RS: the launch has a rank-one term, `rs` = rowscale[row] (0 on every split but the first), fetched by the CALLER before
// its store loop.  A load inside the loop -- even one behind a branch that is never taken -- makes the compiler open every
// iteration with s_waitcnt vmcnt(0), and on this architecture that counter also holds the stores: each 16-byte store then
// waited for the one before it to be acknowledged by memory (16-32 round trips per wave and epilogue, ~0.25 us each).
template <bool RS>
__device__ __forceinline__ void store_out4t(const GemmArgs& a, float* C, int64_t row, int col, const f32x4& v, float alpha,
                                            const float (&bv0)[4], bool vec, float rs) {
    float bv[4] = {bv0[0], bv0[1], bv0[2], bv0[3]};
    if (RS) {
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

// NR 16-byte row pieces of an fp32 output, rows `step` elements apart, stored with EVERY value and address formed before
// the first store issues.  On gfx950 a store's data and address registers count as busy until vmcnt has seen the store
// complete: a loop that forms piece r + 1 in the registers of piece r (what the allocator does with a read - scale - store
// loop) opens every iteration with s_waitcnt vmcnt(0) and so waits out the previous store's round trip to memory --
// measured 0.25 us per store, 4-8 us per wave and epilogue, the whole fixed cost of a small GEMM launch.  The caller
// guarantees (wave-uniformly) that all NR pieces exist and are aligned; pieces are alpha * v + bv (* rs[r] with RS).
template <int NR, bool RS, typename LdsRead>
__device__ __forceinline__ void store_rows16(float* dst, int64_t step, float alpha, const float (&bv)[4], const float* rs,
                                             LdsRead&& piece) {
    f32x4 o[NR];
    float* p[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const f32x4 v = piece(r);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[r][c] = alpha * v[c] + (RS ? bv[c] * rs[r] : bv[c]);
        p[r] = dst + r * step;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NR; ++r) *reinterpret_cast<f32x4*>(p[r]) = o[r];
    __builtin_amdgcn_sched_barrier(0);
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
