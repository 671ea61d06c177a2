// Gene-width bf16 GEMMs (encoder L0 / decoder output layer: forward, dgrad, wgrad) for gfx950.
//
// 256x256 block tile, BK = 64, 8 waves (2 x 4), each wave 128x64 = 4x2 MFMA tiles of v_mfma_f32_32x32x16_bf16,
// fp32 accumulate.  Both operand tiles are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4, no
// registers, no ds_write), double buffered (2 x 64 KB), ONE barrier per K-tile: the DMA of tile t+1 is in
// flight while tile t is multiplied.  An LDS-DMA image is lane-linear, so bank conflicts are removed by
// permuting the per-lane SOURCE address and applying the same permutation in the fragment read:
//   RC operand (reduction index contiguous), image [256 x][8 chunks of 16 B]: chunk c of row x sits at
//       position c ^ ((x >> 1) & 7)  -> ds_read_b128 conflict-free for every 16-lane service group;
//   XC operand (output index contiguous), image [64 k][8 granules of 64 B]: granule g of k-row k sits at
//       position g ^ (k & 3) -> the 4 k-rows of one ds_read_b64_tr_b16 (gfx950 transposing read) fall in
//       the 4 different quarters of the 256-B bank row.
// Workgroup ids are remapped so that the blocks of one XCD (one L2) work on a compact panel of tiles.
// A reduction length that is not a multiple of 64 ends with one tile staged through registers (zero filled).
#include "gemm.h"

namespace c2 {
namespace wide {

constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
constexpr int TILE_BYTES = 256 * BK * 2;      // one operand tile: 32 KB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // 128 KB
constexpr int GROUP_M = 8;                    // tiles along M that share a panel of B

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void g_void;

// Per-lane source offsets (elements, relative to the tile's first row / column at reduction index 0) of the 4
// LDS-DMA instructions a wave issues for one operand tile.  Instruction i = 4*wave + j writes LDS bytes
// [1024*i, 1024*i + 1024), lane l the 16 B at 1024*i + 16*l.
template <bool RC>
__device__ __forceinline__ int stage_offset(int j, int64_t ld, int x0, int xtot, int wave, int lane) {
    const int i = 4 * wave + j;
    if (RC) {
        const int row = 8 * i + (lane >> 3), p = lane & 7;
        const int c = p ^ ((row >> 1) & 7);
        const int x = min(row, xtot - 1 - x0);  // rows past the edge re-read the last row (outputs discarded)
        return x * (int)ld + c * 8;
    } else {
        const int k = 2 * i + (lane >> 5), p = lane & 31;
        const int ch = (((p >> 2) ^ (k & 3)) << 2) | (p & 3);
        int x = ch * 8;
        if (x0 + x >= xtot) x = 0;  // chunk entirely past the edge: any valid address (outputs discarded)
        return k * (int)ld + x;
    }
}

// First reduction index (inside the K-tile) of the chunk that lane `lane` moves in instruction j.
template <bool RC>
__device__ __forceinline__ int stage_k(int j, int wave, int lane) {
    const int i = 4 * wave + j;
    if (RC) {
        const int row = 8 * i + (lane >> 3), p = lane & 7;
        return (p ^ ((row >> 1) & 7)) * 8;
    }
    return 2 * i + (lane >> 5);
}

__device__ __forceinline__ void stage_dma(const bf16_t* base, const int (&voff)[4], char* tile, int wave) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((g_void*)(base + voff[j]), (lds_void*)(tile + (4 * wave + j) * 1024), 16, 0, 0);
}

// Last, partial K-tile: same image through registers, reduction indices >= kvalid are zero.
template <bool RC>
__device__ __forceinline__ void stage_tail(const bf16_t* base, const int (&voff)[4], char* tile, int wave, int lane,
                                           int kvalid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (stage_k<RC>(j, wave, lane) < kvalid) v = *reinterpret_cast<const u32x4*>(base + voff[j]);
        *reinterpret_cast<u32x4*>(tile + (4 * wave + j) * 1024 + lane * 16) = v;
    }
}

// Fragment of v_mfma_f32_32x32x16_bf16 for rows/cols [xt, xt+32) and reduction step ks (16 wide):
// lane l holds x = xt + (l & 31), k = 16*ks + 8*(l >> 5) + 0..7.
template <bool RC>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int xt, int ks, int lane) {
    if (RC) {
        const int row = xt + (lane & 31);
        const int c = (2 * ks + (lane >> 5)) ^ ((row >> 1) & 7);
        return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (c << 4));
    } else {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p4 = i & 3, h = lane >> 5;
        const int k = 16 * ks + 8 * h + q;
        const int xb = (xt + 16 * (g & 1) + 4 * p4) * 2;
        const int off = k * 512 + ((((xb >> 6) ^ q) & 7) << 6) + (xb & 63);
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off + 4 * 512));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return __builtin_bit_cast(bf16x8, r);
    }
}

template <bool A_RC, bool B_RC>
__device__ __forceinline__ void compute_tile(const char* stage, int wm, int wn, int lane, f32x16 (&acc)[4][2]) {
    const char* tA = stage;
    const char* tB = stage + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 fa[4], fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = read_frag<B_RC>(tB, wn * 64 + j * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag<A_RC>(tA, wm * 128 + i * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
}

template <bool A_RC, bool B_RC>
__global__ __launch_bounds__(NT) void gemm_wide_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    // ---- workgroup -> (split, tile) with the blocks of one XCD on a compact panel
    int wg;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tiles = tiles_m * tiles_n;
    const int split = wg / tiles;
    int tm, tn;
    {
        const int t = wg - split * tiles;
        const int per = GROUP_M * tiles_n;
        const int gid = t / per, first = gid * GROUP_M;
        const int gsz = min(tiles_m - first, GROUP_M);
        const int u = t - gid * per;
        tm = first + u % gsz;
        tn = u / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int rbeg = split * a.r_per_split;
    const int rend = min(a.R, rbeg + a.r_per_split);
    const int rlen = max(rend - rbeg, 0);
    const int nfull = rlen / BK;
    const int ktail = rlen - nfull * BK;
    const int nt = nfull + (ktail ? 1 : 0);

    // uniform tile bases (scalar registers) + 32-bit per-lane offsets
    const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A) + (A_RC ? (int64_t)m0 * a.lda + rbeg : (int64_t)rbeg * a.lda + m0);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(a.B) + (B_RC ? (int64_t)n0 * a.ldb + rbeg : (int64_t)rbeg * a.ldb + n0);
    int voa[4], vob[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        voa[j] = stage_offset<A_RC>(j, a.lda, m0, a.M, wave, lane);
        vob[j] = stage_offset<B_RC>(j, a.ldb, n0, a.N, wave, lane);
    }
    const int64_t stepA = A_RC ? BK : (int64_t)BK * a.lda;
    const int64_t stepB = B_RC ? BK : (int64_t)BK * a.ldb;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto stage_in = [&](int t, char* stage) {
        if (t < nfull) {
            stage_dma(A + t * stepA, voa, stage, wave);
            stage_dma(B + t * stepB, vob, stage + TILE_BYTES, wave);
        } else if (t < nt) {
            stage_tail<A_RC>(A + t * stepA, voa, stage, wave, lane, ktail);
            stage_tail<B_RC>(B + t * stepB, vob, stage + TILE_BYTES, wave, lane, ktail);
        }
    };

    stage_in(0, smem);
    for (int t = 0; t < nt; t += 2) {
        __syncthreads();  // tile t has landed (vmcnt(0) + barrier); stage 1 is free
        stage_in(t + 1, smem + STAGE_BYTES);
        compute_tile<A_RC, B_RC>(smem, wm, wn, lane, acc);
        if (t + 1 < nt) {
            __syncthreads();
            stage_in(t + 2, smem);
            compute_tile<A_RC, B_RC>(smem + STAGE_BYTES, wm, wn, lane, acc);
        }
    }

    // ---- epilogue: C = alpha * acc + bias
    const float alpha = a.alpha_dev ? *a.alpha_dev : 1.f;
    float* C = a.C + (int64_t)split * a.slab_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        const float bv = (a.bias && col < a.N) ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < a.M && col < a.N) C[(int64_t)row * a.ldc + col] = alpha * acc[i][j][e] + bv;
            }
        }
    }
}

}  // namespace wide

bool wide_eligible(int layout, int dtype, const GemmArgs& a) {
    if (dtype != C2_BF16) return false;
    if (!aligned16(a.A) || !aligned16(a.B) || (a.lda & 7) || (a.ldb & 7) || (a.R & 7)) return false;
    // worth a 256x256 tile: a gene-width problem
    if ((int64_t)a.M * a.N < 256ll * 256 * 32 || a.R < 256 || a.M < 192 || a.N < 192) return false;
    return true;
}

int launch_gemm_wide(int layout, const GemmArgs& a, int nslab, hipStream_t stream) {
    using namespace wide;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const dim3 grid(tiles_m * tiles_n * nslab);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    if (layout == 0)
        hipLaunchKernelGGL((gemm_wide_kernel<true, true>), grid, dim3(NT), LDS_BYTES, stream, a, tiles_m, tiles_n);
    else if (layout == 1)
        hipLaunchKernelGGL((gemm_wide_kernel<true, false>), grid, dim3(NT), LDS_BYTES, stream, a, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL((gemm_wide_kernel<false, false>), grid, dim3(NT), LDS_BYTES, stream, a, tiles_m, tiles_n);
    return 0;
}

}  // namespace c2
