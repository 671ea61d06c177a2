// Gene-width bf16 GEMMs (encoder L0 / decoder output layer: forward, dgrad, wgrad) for gfx950.
//
// 256x256 block tile, BK = 64, 8 waves (2 x 4), each wave 128x64 = 4x2 MFMA tiles of v_mfma_f32_32x32x16_bf16,
// fp32 accumulate.  Operand tiles (32 KB each) are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4: no
// registers, no ds_write) into a RING OF FIVE 32 KB slots (all 160 KB of LDS): loads alternate A_t, B_t, A_t+1, ...
// and load n lives in slot n % 5.  One raw s_barrier per K-tile; the wait in front of it is a COUNTED vmcnt that
// leaves the newest load in flight, so the DMA stream never drains: while tile t is multiplied, B_t+1 and A_t+2
// are being fetched and A_t+1 has already landed.
// An LDS-DMA image is lane-linear, so bank conflicts are removed by permuting the per-lane SOURCE address and
// applying the same permutation in the fragment read (SQ_LDS_BANK_CONFLICT = 0 measured):
//   RC operand (reduction index contiguous), image [256 x][8 chunks of 16 B]: chunk c of row x sits at
//       position c ^ ((x >> 1) & 7)  -> ds_read_b128 conflict-free for every 16-lane service group;
//   XC operand (output index contiguous), image [64 k][8 granules of 64 B]: granule g of k-row k sits at
//       position g ^ (k & 3) -> the 4 k-rows of one ds_read_b64_tr_b16 (gfx950 transposing read) fall in
//       the 4 different quarters of the 256-B bank row.
// Workgroup ids are remapped so that the blocks of one XCD (one L2) work on a compact panel of tiles.
// A reduction length that is not a multiple of 64 ends with one tile staged through registers (zero filled).
// Epilogue: accumulators are transposed through LDS (each wave its own 16 KB) so that every store is a
// 16-byte-per-lane, 256-B-per-row global_store_dwordx4.
#include <stdlib.h>

#include "gemm.h"

namespace c2 {
namespace wide {

constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
constexpr int TILE_BYTES = 256 * BK * 2;      // one operand tile = one ring slot: 32 KB
constexpr int NSLOT = 5;
constexpr int LDS_BYTES = NSLOT * TILE_BYTES; // 160 KB
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

template <int J0, int J1>
__device__ __forceinline__ void stage_dma(const bf16_t* base, const int (&voff)[4], char* tile, int wave) {
#pragma unroll
    for (int j = J0; j < J1; ++j)
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
__device__ __forceinline__ void load_frags(const char* tA, const char* tB, int wm, int wn, int ks, int lane,
                                           bf16x8 (&fa)[4], bf16x8 (&fb)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = read_frag<B_RC>(tB, wn * 64 + j * 32, ks, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = read_frag<A_RC>(tA, wm * 128 + i * 32, ks, lane);
}

template <typename F>
__device__ __forceinline__ void mma_step(const bf16x8 (&fa)[4], const bf16x8 (&fb)[2], f32x16 (&acc)[4][2], F&& piece,
                                         int p0) {
    piece(p0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    piece(p0 + 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 2; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

// One K-tile: the fragments of step ks+1 are read while step ks is multiplied (two register sets).
// `piece(p)`, p = 0..7, issues one eighth of the iteration's LDS-DMA in front of each group of 4 MFMAs, so that a
// wave never sits in a long run of VMEM issue while the memory pipeline is full.
template <bool A_RC, bool B_RC, typename F>
__device__ __forceinline__ void compute_tile(const char* tA, const char* tB, int wm, int wn, int lane,
                                             f32x16 (&acc)[4][2], F&& piece) {
    bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
    load_frags<A_RC, B_RC>(tA, tB, wm, wn, 0, lane, fa0, fb0);
    load_frags<A_RC, B_RC>(tA, tB, wm, wn, 1, lane, fa1, fb1);
    mma_step(fa0, fb0, acc, piece, 0);
    load_frags<A_RC, B_RC>(tA, tB, wm, wn, 2, lane, fa0, fb0);
    mma_step(fa1, fb1, acc, piece, 2);
    load_frags<A_RC, B_RC>(tA, tB, wm, wn, 3, lane, fa1, fb1);
    mma_step(fa0, fb0, acc, piece, 4);
    mma_step(fa1, fb1, acc, piece, 6);
}

template <bool A_RC, bool B_RC>
__global__ __launch_bounds__(NT) void gemm_wide_kernel(GemmArgs a, int tiles_m, int tiles_n, int dbg) {
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

    // load n: operand tile (n & 1 ? B : A) of K-tile n >> 1, into ring slot n % 5; `q` = which of the 4 DMA
    // instructions (-1: all four).  A tail tile goes through registers in one piece.
    auto issue = [&](int n, int slot, int q) {
        if (dbg & 1) return;
        const int t = n >> 1;
        char* dst = smem + slot * TILE_BYTES;
        const bf16_t* src = (n & 1) ? B + t * stepB : A + t * stepA;
        if (t < nfull) {
            if (n & 1) {
                if (q < 0 || q == 0) stage_dma<0, 1>(src, vob, dst, wave);
                if (q < 0 || q == 1) stage_dma<1, 2>(src, vob, dst, wave);
                if (q < 0 || q == 2) stage_dma<2, 3>(src, vob, dst, wave);
                if (q < 0 || q == 3) stage_dma<3, 4>(src, vob, dst, wave);
            } else {
                if (q < 0 || q == 0) stage_dma<0, 1>(src, voa, dst, wave);
                if (q < 0 || q == 1) stage_dma<1, 2>(src, voa, dst, wave);
                if (q < 0 || q == 2) stage_dma<2, 3>(src, voa, dst, wave);
                if (q < 0 || q == 3) stage_dma<3, 4>(src, voa, dst, wave);
            }
        } else if (t < nt && q <= 0) {
            if (n & 1) stage_tail<B_RC>(src, vob, dst, wave, lane, ktail);
            else stage_tail<A_RC>(src, voa, dst, wave, lane, ktail);
        }
    };

    issue(0, 0, -1);
    issue(1, 1, -1);
    issue(2, 2, -1);
    int sa = 0;  // slot of A_t; B_t is in the next one
    for (int t = 0; t < nt; ++t) {
        // loads 2t and 2t+1 must have landed; load 2t+2 (4 DMA instructions per wave) may stay in flight
        if (t + 1 < nfull) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int sb = sa + 1 >= NSLOT ? sa + 1 - NSLOT : sa + 1;
        const int s3 = sa + 3 >= NSLOT ? sa + 3 - NSLOT : sa + 3;
        const int s4 = sa + 4 >= NSLOT ? sa + 4 - NSLOT : sa + 4;
        // loads 2t+3 and 2t+4 go to the slots tile t-1 occupied: every wave is past its reads of them
        auto piece = [&](int p) {
            if (dbg & 4) {
                if (p == 0) { issue(2 * t + 3, s3, -1); issue(2 * t + 4, s4, -1); }
                return;
            }
            // two DMA instructions in front of every group of 8 MFMAs (measured: 980 TF; one per 4 MFMAs 920,
            // all eight up front 840)
            if (p & 1) return;
            if (p < 4) { issue(2 * t + 3, s3, p); issue(2 * t + 3, s3, p + 1); }
            else { issue(2 * t + 4, s4, p - 4); issue(2 * t + 4, s4, p - 3); }
        };
        if (!(dbg & 2)) {
            compute_tile<A_RC, B_RC>(smem + sa * TILE_BYTES, smem + sb * TILE_BYTES, wm, wn, lane, acc, piece);
        } else {
            piece(0); piece(1); piece(2); piece(3); piece(4); piece(5); piece(6); piece(7);
        }
        sa = sa + 2 >= NSLOT ? sa + 2 - NSLOT : sa + 2;
    }

    // ---- epilogue: C = alpha * acc + bias, transposed through LDS for 16-byte-per-lane stores
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done reading the ring
    asm volatile("" ::: "memory");
    const float alpha = a.alpha_dev ? *a.alpha_dev : 1.f;
    float* C = a.C + (int64_t)split * a.slab_stride;
    float* wl = reinterpret_cast<float*>(smem + wave * 16384);  // [64 rows][64 cols] fp32, private to the wave
    const int col4 = n0 + wn * 64 + (lane & 15) * 4;
    const bool vec_ok = ((a.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (col4 + 4 <= a.N);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (col4 + c < a.N) bv[c] = a.bias[col4 + c];
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    wl[r * 64 + j * 32 + (lane & 31)] = acc[half * 2 + ii][j][e];
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int r = rr * 4 + (lane >> 4);
            f32x4 v = *reinterpret_cast<const f32x4*>(wl + r * 64 + (lane & 15) * 4);
            const int row = m0 + wm * 128 + half * 64 + r;
            if (row < a.M) {
                float* dst = C + (int64_t)row * a.ldc + col4;
                if (vec_ok) {
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = alpha * v[c] + bv[c];
                    *reinterpret_cast<f32x4*>(dst) = o;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (col4 + c < a.N) dst[c] = alpha * v[c] + bv[c];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next half overwrites
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
    static const int dbg = getenv("C2_GEMM_DBG") ? atoi(getenv("C2_GEMM_DBG")) : 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    if (layout == 0)
        hipLaunchKernelGGL((gemm_wide_kernel<true, true>), grid, dim3(NT), LDS_BYTES, stream, a, tiles_m, tiles_n, dbg);
    else if (layout == 1)
        hipLaunchKernelGGL((gemm_wide_kernel<true, false>), grid, dim3(NT), LDS_BYTES, stream, a, tiles_m, tiles_n, dbg);
    else
        hipLaunchKernelGGL((gemm_wide_kernel<false, false>), grid, dim3(NT), LDS_BYTES, stream, a, tiles_m, tiles_n, dbg);
    return 0;
}

}  // namespace c2
