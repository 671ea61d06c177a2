// The LDS-DMA GEMM kernels of gemm_wide.hip on IEEE half operands (v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16):
// namespace c2::wide_f16, entry points wide_tile_f16 / launch_gemm_wide_f16 / launch_gemm_wide_rows_f16 (gemm.h).
#define C2_WIDE_F16 1
#include "gemm_wide.hip"
