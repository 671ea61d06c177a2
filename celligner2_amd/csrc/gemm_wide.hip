// LDS-DMA bf16 GEMMs (forward, dgrad, wgrad of the Linear layers) for gfx950.
//
// Two instances of one template:
//   NW = 8: 256x256 block tile, 8 waves (2 x 4), each wave 128x64 = 4x2 MFMA tiles -- the gene-width layers;
//   NW = 4: 128x128 block tile, 4 waves (2 x 2), each wave  64x64 = 2x2 MFMA tiles, two blocks per CU -- the narrow
//           hidden layers, whose fp32 operands arrive as two bf16 planes (hi, lo = x - hi) and are multiplied as
//           three products hi*hi + hi*lo + lo*hi over a three-segment virtual reduction ("bf16x3").
// BK = 64, v_mfma_f32_16x16x32_bf16 (256x256 forward / dgrad) or v_mfma_f32_32x32x16_bf16, fp32 accumulate.  Operand tiles (32 KB each) are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4: no
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

// This file is compiled twice: as it stands (bf16 operands), and through gemm_wide_f16.hip with C2_WIDE_F16 defined --
// the same kernels on v_mfma_f32_*_f16 (IEEE half operands: 11 significant bits instead of 8 at the same matrix rate;
// the encoder's first layer reads its input and weight that way in bf16 mode).  Nothing but the MFMA instruction depends
// on the 2-byte format: LDS-DMA staging, transposing reads and the epilogue move 16-bit words.  The twin lives in
// namespace c2::wide_f16 and exports wide_tile_f16 / launch_gemm_wide_f16 / launch_gemm_wide_rows_f16.
#ifdef C2_WIDE_F16
#define C2_WIDE_NS wide_f16
#define C2_WIDE_DT C2_F16
#define wide_tile wide_tile_f16
#define launch_gemm_wide launch_gemm_wide_f16
#define launch_gemm_wide_rows launch_gemm_wide_rows_f16
#define c2_dbg_clk c2_dbg_clk_f16
typedef __attribute__((ext_vector_type(8))) _Float16 c2_f16x8;
#define C2_MFMA_32x32x16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c2_f16x8, a), __builtin_bit_cast(c2_f16x8, b), c, 0, 0, 0)
#define C2_MFMA_16x16x32(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(c2_f16x8, a), __builtin_bit_cast(c2_f16x8, b), c, 0, 0, 0)
#else
#define C2_WIDE_NS wide
#define C2_WIDE_DT C2_BF16
#define C2_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define C2_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif

namespace c2 {
namespace C2_WIDE_NS {

// Ablation builds only (-DC2_GEMM_DBG=1: no DMA, 2: no MFMA, 4 / 8: no A / B DMA in the steady loop); 0 in the product.
#ifndef C2_GEMM_DBG
#define C2_GEMM_DBG 0
#endif
constexpr int dbg = C2_GEMM_DBG;
// 0: ablation build -- every epilogue on its read - scale - store loop (A/B of store_rows16)
#ifndef C2_EPI_BATCH
#define C2_EPI_BATCH 1
#endif

#if C2_GEMM_DBG & 16
__device__ unsigned long long g_clk[4];  // shader-clock cycles and 100 MHz ticks of block 0 (clock probe)
#endif

constexpr int BK = 64;
constexpr int GROUP_M = 8;  // tiles along M that share a panel of B

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void g_void;

// Per-lane source offset (elements, relative to the tile's first row / column at reduction index 0) of LDS-DMA
// instruction i = 4*wave + j of one operand tile of BX rows (RC) / BX columns (XC).  Instruction i writes LDS bytes
// [1024*i, 1024*i + 1024), lane l the 16 B at 1024*i + 16*l.
// XC image, k-row k: the 64-byte granule index is XORed with k & 3 (the 4 k-rows of one transposing read fall in 4
// different granules = 4 different quarters of the 256-byte bank row), and for BX = 256 the two 32-byte halves of every
// granule are swapped when (k >> 3) & 1: a 32-lane half of the 16x16x32 operand reads, in the same columns, four
// k-rows and the four k-rows 8 further on -- 4 KB apart, i.e. on the same banks (measured: SQ_LDS_BANK_CONFLICT = 26 % of
// the LDS cycles of the dgrad kernel); with the swap they take the other half of each quarter.  The 32x32x16 read's
// 32-lane halves stay inside one value of (k >> 3) & 1 and remain conflict-free.
template <int BX>
__device__ __forceinline__ int xc_swz(int k) {
    return k & 3;
}
template <int BX>
__device__ __forceinline__ int xc_half(int k) {
    return BX == 256 ? ((k >> 3) & 1) : 0;
}

template <bool RC, int BX, int PPW = 4>  // PPW: LDS-DMA pieces (1 KB) of an operand tile that one wave moves
__device__ __forceinline__ int stage_offset(int j, int64_t ld, int x0, int xtot, int wave, int lane) {
    const int i = PPW * wave + j;
    if (RC) {
        const int row = 8 * i + (lane >> 3), p = lane & 7;
        const int c = p ^ ((row >> 1) & 7);
        const int x = min(row, xtot - 1 - x0);  // rows past the edge re-read the last row (outputs discarded)
        return x * (int)ld + c * 8;
    } else {
        constexpr int LPR = BX / 8;  // lanes (16-byte chunks) per k-row
        const int k = (64 / LPR) * i + lane / LPR, p = lane % LPR;
        const int ch = (((p >> 2) ^ xc_swz<BX>(k)) << 2) | ((p & 3) ^ (xc_half<BX>(k) << 1));
        int x = ch * 8;
        if (x0 + x >= xtot) x = 0;  // chunk entirely past the edge: any valid address (outputs discarded)
        return k * (int)ld + x;
    }
}

// First reduction index (inside the K-tile) of the chunk that lane `lane` moves in instruction j.
template <bool RC, int BX, int PPW = 4>
__device__ __forceinline__ int stage_k(int j, int wave, int lane) {
    const int i = PPW * wave + j;
    if (RC) {
        const int row = 8 * i + (lane >> 3), p = lane & 7;
        return (p ^ ((row >> 1) & 7)) * 8;
    }
    constexpr int LPR = BX / 8;
    return (64 / LPR) * i + lane / LPR;
}

// `wave` must be wave-uniform in the compiler's eyes (v_readfirstlane once per kernel): the LDS destination then
// reaches m0 through scalar arithmetic instead of a v_readfirstlane per instruction.  The source address is formed
// as (uniform base) + zext(32-bit per-lane BYTE offset), the shape the backend turns into the scalar-base form of the
// instruction (global_load_lds_dwordx4 v_off, s[base:base+1]) -- no 64-bit vector add per instruction.
template <int J0, int J1, int PPW = 4>
__device__ __forceinline__ void stage_dma(const bf16_t* base, const int (&voff)[4], char* tile, int wave) {
    const char* b = reinterpret_cast<const char*>(base);
#pragma unroll
    for (int j = J0; j < J1; ++j)
        __builtin_amdgcn_global_load_lds((g_void*)(b + (uint64_t)((uint32_t)voff[j] * 2u)),
                                         (lds_void*)(tile + (PPW * wave + j) * 1024), 16, 0, 0);
}

// Last, partial K-tile: same image through registers, reduction indices >= kvalid are zero.
template <bool RC, int BX, int PPW = 4>
__device__ __forceinline__ void stage_tail(const bf16_t* base, const int (&voff)[4], char* tile, int wave, int lane,
                                           int kvalid) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (stage_k<RC, BX, PPW>(j, wave, lane) < kvalid) v = *reinterpret_cast<const u32x4*>(base + voff[j]);
        *reinterpret_cast<u32x4*>(tile + (PPW * wave + j) * 1024 + lane * 16) = v;
    }
}

// Fragment of v_mfma_f32_32x32x16_bf16 for rows/cols [xt, xt+32) and reduction step ks (16 wide):
// lane l holds x = xt + (l & 31), k = 16*ks + 8*(l >> 5) + 0..7.
template <bool RC, int BX>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int xt, int ks, int lane) {
    if (RC) {
        const int row = xt + (lane & 31);
        const int c = (2 * ks + (lane >> 5)) ^ ((row >> 1) & 7);
        return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (c << 4));
    } else {
        constexpr int RB = BX * 2;  // bytes per k-row
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p4 = i & 3, h = lane >> 5;
        const int k = 16 * ks + 8 * h + q;
        const int xb = (xt + 16 * (g & 1) + 4 * p4) * 2;
        const int off = k * RB + ((((xb >> 6) ^ xc_swz<BX>(k)) & (RB / 64 - 1)) << 6) + ((xb & 63) ^ (xc_half<BX>(k) << 5));
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off + 4 * RB));  // (k + 4: same swizzle)
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return __builtin_bit_cast(bf16x8, r);
    }
}

template <bool A_RC, bool B_RC, int BX, int MI>
__device__ __forceinline__ void load_frags(const char* tA, const char* tB, int wm, int wn, int ks, int lane,
                                           bf16x8 (&fa)[MI], bf16x8 (&fb)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = read_frag<B_RC, BX>(tB, wn * 64 + j * 32, ks, lane);
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = read_frag<A_RC, BX>(tA, wm * (MI * 32) + i * 32, ks, lane);
}

template <int MI, typename F>
__device__ __forceinline__ void mma_step(const bf16x8 (&fa)[MI], const bf16x8 (&fb)[2], f32x16 (&acc)[MI][2], F&& piece,
                                         int p0) {
    piece(p0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[i][j] = C2_MFMA_32x32x16(fa[i], fb[j], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
}

// One K-tile: the fragments of step ks+1 are read while step ks is multiplied (two register sets).
// `piece(p)`, p = 0..3, issues a quarter of the iteration's LDS-DMA (two instructions) in front of each MFMA group,
// so that a wave never sits in a long run of VMEM issue while the memory pipeline is full (measured at NW = 8:
// 980 TF; one instruction per 4 MFMAs 920; all eight up front 840; all eight after the last MFMA group, where the
// wave would otherwise idle at the barrier, 830; in the second half of the tile 860).
template <bool A_RC, bool B_RC, int BX, int MI, typename F>
__device__ __forceinline__ void compute_tile(const char* tA, const char* tB, int wm, int wn, int lane,
                                             f32x16 (&acc)[MI][2], F&& piece) {
    bf16x8 fa0[MI], fb0[2], fa1[MI], fb1[2];
    load_frags<A_RC, B_RC, BX, MI>(tA, tB, wm, wn, 0, lane, fa0, fb0);
    load_frags<A_RC, B_RC, BX, MI>(tA, tB, wm, wn, 1, lane, fa1, fb1);
    mma_step<MI>(fa0, fb0, acc, piece, 0);
    load_frags<A_RC, B_RC, BX, MI>(tA, tB, wm, wn, 2, lane, fa0, fb0);
    mma_step<MI>(fa1, fb1, acc, piece, 1);
    load_frags<A_RC, B_RC, BX, MI>(tA, tB, wm, wn, 3, lane, fa1, fb1);
    mma_step<MI>(fa0, fb0, acc, piece, 2);
    mma_step<MI>(fa1, fb1, acc, piece, 3);
}

// ---- v_mfma_f32_16x16x32_bf16 (the 256x256 tile, forward and dgrad): same operand bytes per FLOP as 32x32x16 for a 128x64 wave tile,
// same cycles per FLOP, but the chip holds a higher clock on it under load (MI355X_MICROARCH.md, DVFS give-back (7)).
// Fragment for rows/cols [xt, xt+16) and reduction step ks (32 wide): lane l holds x = xt + (l & 15),
// k = 32*ks + 8*(l >> 4) + 0..7.
template <bool RC, int BX>
__device__ __forceinline__ bf16x8 read_frag16(const char* tile, int xt, int ks, int lane) {
    if (RC) {
        const int row = xt + (lane & 15);
        const int c = (4 * ks + (lane >> 4)) ^ ((row >> 1) & 7);
        return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (c << 4));
    } else {
        // each 16-lane group g transposes two blocks of 4 k-rows x 16 columns: k = 32 ks + 8 g + {0..3}, then + 4;
        // lane 4q + p4 of the group supplies the address of k-row q, columns 4 p4 .. 4 p4 + 3
        constexpr int RB = BX * 2;  // bytes per k-row
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p4 = i & 3;
        const int k = 32 * ks + 8 * g + q;
        const int xb = (xt + 4 * p4) * 2;
        const int off = k * RB + ((((xb >> 6) ^ xc_swz<BX>(k)) & (RB / 64 - 1)) << 6) + ((xb & 63) ^ (xc_half<BX>(k) << 5));
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off + 4 * RB));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return __builtin_bit_cast(bf16x8, r);
    }
}

template <typename F>
__device__ __forceinline__ void mma16_group(const bf16x8 (&fa)[4], const bf16x8 (&fb)[4], f32x4 (&acc)[8][4], int i0,
                                            F&& piece, int p0) {
    piece(p0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i0 + i][j] = C2_MFMA_16x16x32(fa[i], fb[j], acc[i0 + i][j]);
    __builtin_amdgcn_s_setprio(0);
}

// One K-tile of the 128x64 wave tile as 8 x 4 MFMA tiles of 16x16, two reduction steps of 32: four groups of 16 MFMAs
// (256 cycles each); the fragments a group needs are read while the group before it runs (B of a step and the lower
// four A tiles together, the upper four A tiles one group later: 64 fragment registers instead of 96).
template <bool A_RC, bool B_RC, int BX, typename F>
__device__ __forceinline__ void compute_tile16(const char* tA, const char* tB, int wm, int wn, int lane,
                                               f32x4 (&acc)[8][4], F&& piece) {
    bf16x8 b0[4], b1[4], alo[4], ahi[4];
    auto ldB = [&](int ks, bf16x8 (&f)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = read_frag16<B_RC, BX>(tB, wn * 64 + j * 16, ks, lane);
    };
    auto ldA = [&](int ks, int half, bf16x8 (&f)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = read_frag16<A_RC, BX>(tA, wm * 128 + half * 64 + i * 16, ks, lane);
    };
    ldB(0, b0);
    ldA(0, 0, alo);
    ldA(0, 1, ahi);
    mma16_group(alo, b0, acc, 0, piece, 0);
    ldB(1, b1);
    ldA(1, 0, alo);
    mma16_group(ahi, b0, acc, 4, piece, 1);
    ldA(1, 1, ahi);
    mma16_group(alo, b1, acc, 0, piece, 2);
    mma16_group(ahi, b1, acc, 4, piece, 3);
}

// The same K-tile rotated by one MFMA group (C2_WIDE_ROT, steady-state loop only): right after a tile's barrier all
// eight waves ask the LDS for their first fragments at once, and nothing they could multiply has arrived -- so the LAST
// group of the previous tile (its operands `ahi`, `b1` are in registers since before the barrier) is multiplied there,
// under that burst of reads.  Same accumulation order per accumulator, so bit-identical results; the caller zero-fills
// `ahi` / `b1` before the first tile and multiplies the group left over after the last one (flush_tile16).
template <bool A_RC, bool B_RC, int BX, typename F>
__device__ __forceinline__ void compute_tile16_rot(const char* tA, const char* tB, int wm, int wn, int lane,
                                                   f32x4 (&acc)[8][4], F&& piece, bf16x8 (&ahi)[4], bf16x8 (&b1)[4]) {
    bf16x8 b0[4], alo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b0[j] = read_frag16<B_RC, BX>(tB, wn * 64 + j * 16, 0, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) alo[i] = read_frag16<A_RC, BX>(tA, wm * 128 + i * 16, 0, lane);
    __builtin_amdgcn_sched_barrier(0);  // the reads are on their way before the held group starts
    piece(0);  // (behind the group's first eight MFMAs instead: measured 0.3 % slower per step)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // a row of the held group, then the read that refills its A fragment: three of the four
                                   // reads the next group waits for (every LDS wait here is a full drain) are issued early
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 + i][j] = C2_MFMA_16x16x32(ahi[i], b1[j], acc[4 + i][j]);
        ahi[i] = read_frag16<A_RC, BX>(tA, wm * 128 + 64 + i * 16, 0, lane);
    }
    __builtin_amdgcn_s_setprio(0);
    mma16_group(alo, b0, acc, 0, piece, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) b1[j] = read_frag16<B_RC, BX>(tB, wn * 64 + j * 16, 1, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) alo[i] = read_frag16<A_RC, BX>(tA, wm * 128 + i * 16, 1, lane);
    mma16_group(ahi, b0, acc, 4, piece, 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) ahi[i] = read_frag16<A_RC, BX>(tA, wm * 128 + 64 + i * 16, 1, lane);
    mma16_group(alo, b1, acc, 0, piece, 3);
}
__device__ __forceinline__ void flush_tile16(f32x4 (&acc)[8][4], const bf16x8 (&ahi)[4], const bf16x8 (&b1)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 + i][j] = C2_MFMA_16x16x32(ahi[i], b1[j], acc[4 + i][j]);
}
#ifndef C2_WIDE_ROT
#define C2_WIDE_ROT 1
#endif

// s_waitcnt vmcnt(4 * c): all but the c newest operand-tile loads (4 DMA instructions per wave each) have landed
__device__ __forceinline__ void wait_loads(int c) {
    switch (c) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    }
}

// Ring of NSLOT = 2*D + 1 operand-tile slots: while K-tile T is multiplied, the loads of tiles T+1 .. T+D-1 and A of
// T+D are in flight or landed.  D = 2 for the 256x256 tile (5 x 32 KB), D = 4 for the 128x128 tile (9 x 16 KB): its
// K-step is short (16 MFMAs per wave), so the ~1 us LDS-DMA round trip needs more tiles ahead of it.
// IDX: 0 plain operands; 1 the rows of the (RC) A operand are read through a.rows (forward of a batch that lives in a
// resident matrix); 2 the k-rows of the (XC) B operand are (its weight gradient).  The per-lane DMA source address
// carries the row: for IDX 1 once per block, for IDX 2 per K-tile from eight wave-uniform (scalar) index loads.
// DD (NW = 4 only): ring depth 2 instead of 4 -- 5 x 16 KB = 80 KB, TWO blocks per CU -- for launches whose blocks reduce
// over a few K-tiles only (the gene-width layers at the reference's batch of 128: a weight gradient is 298 tiles of two
// K-tiles each, which at one block per CU ran as a full round and a sixth of a second one; a deep ring has nothing to
// prefetch there).
template <bool A_RC, bool B_RC, int NW, int IDX = 0, int DD = 0>
__global__ __launch_bounds__(64 * NW) void gemm_wide_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    static_assert(IDX == 0 || (IDX == 1 && A_RC) || (IDX == 2 && !B_RC), "row index on an RC A or an XC B operand");
    constexpr int BX = 32 * NW;             // block tile edge: 256 or 128
    constexpr int D = DD ? DD : (NW == 8 ? 2 : 4);
    constexpr int NSLOT = 2 * D + 1;
    constexpr int MI = NW / 2;              // MFMA tiles per wave along M
    constexpr int WN = NW / 2;              // waves along N
    constexpr int TILE_BYTES = BX * BK * 2; // one operand tile = one ring slot
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // the same number, known to be wave-uniform
    const int wm = wave / WN, wn = wave % WN;
#if C2_GEMM_DBG & 16
    const unsigned long long clk0 = clock64(), rt0 = wall_clock64();
#endif

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
    const int m0 = tm * BX, n0 = tn * BX;
    const int rbeg = split * a.r_per_split;
    const int rend = min(a.R, rbeg + a.r_per_split);
    const int rlen = max(rend - rbeg, 0);
    const int nfull = rlen / BK;
    const int ktail = rlen - nfull * BK;
    const int nt1 = nfull + (ktail ? 1 : 0);        // K-tiles of one segment
    const int nt = a.A_lo ? 3 * nt1 : nt1;          // bf16x3: segments (hi,hi) (hi,lo) (lo,hi)

    // uniform tile bases (scalar registers) + 32-bit per-lane offsets
    const int64_t offA = IDX == 1 ? (int64_t)rbeg : (A_RC ? (int64_t)m0 * a.lda + rbeg : (int64_t)rbeg * a.lda + m0);
    const int64_t offB = IDX == 2 ? (int64_t)n0 : (B_RC ? (int64_t)n0 * a.ldb + rbeg : (int64_t)rbeg * a.ldb + n0);
    const bf16_t* Ah = reinterpret_cast<const bf16_t*>(a.A) + offA;
    const bf16_t* Bh = reinterpret_cast<const bf16_t*>(a.B) + offB;
    const bf16_t* Al = reinterpret_cast<const bf16_t*>(a.A_lo) + offA;
    const bf16_t* Bl = reinterpret_cast<const bf16_t*>(a.B_lo) + offB;
    int voa[4], vob[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        voa[j] = stage_offset<A_RC, BX>(j, a.lda, m0, a.M, wave, lane);
        vob[j] = stage_offset<B_RC, BX>(j, a.ldb, n0, a.N, wave, lane);
    }
    const int64_t stepA = A_RC ? BK : (int64_t)BK * a.lda;
    const int64_t stepB = IDX == 2 ? 0 : (B_RC ? BK : (int64_t)BK * a.ldb);
    if (IDX == 1) {  // row part of A's offsets: the indexed row instead of the tile row (edge rows re-read the last row)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 8 * (4 * wave + j) + (lane >> 3);
            const int x = min(row, a.M - 1 - m0);
            voa[j] += (a.rows[m0 + x] - x) * (int)a.lda;
        }
    }
    // IDX 2: vob[j] = rows[k-row] * ldb + xob[j]; the DMA instruction j of this wave covers k-rows 8*wave + 2j + {0,1}
    int xob[4];
    int kr[8];  // the wave's eight row indices of the B tile that is issued next
    auto load_krows = [&](int t) {
        const int32_t* rp = a.rows + rbeg + t * BK + 8 * wave_u;  // wave-uniform address: scalar loads
#pragma unroll
        for (int q = 0; q < 8; ++q) kr[q] = rp[q];
    };
    auto set_vob = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) vob[j] = ((lane >> 5) ? kr[2 * j + 1] : kr[2 * j]) * (int)a.ldb + xob[j];
    };
    if (IDX == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 8 * wave + 2 * j + (lane >> 5);
            xob[j] = vob[j] - k * (int)a.ldb;  // the column part stage_offset<XC> computed
        }
    }

    // The 256x256 tile multiplies with v_mfma_f32_16x16x32_bf16 where an operand is read row-wise (forward, dgrad:
    // measured +7 ... +19 % over 32x32x16 on the 19k-gene shapes, same box, same data); the weight gradient, whose two
    // operands both come through the transposing read, measured 0.5-2.6 % slower on it and keeps 32x32x16.
    constexpr bool M16 = NW == 8 && (A_RC || B_RC);
    f32x16 acc[M16 ? 1 : MI][M16 ? 1 : 2];
    f32x4 acc16[M16 ? 8 : 1][4];
    if (M16) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }

    constexpr bool ROT = M16 && C2_WIDE_ROT;
    bf16x8 held_a[ROT ? 4 : 1], held_b[ROT ? 4 : 1];  // (compute_tile16_rot: the previous tile's last MFMA group)
    if (ROT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            held_a[i] = __builtin_bit_cast(bf16x8, s16x8{0, 0, 0, 0, 0, 0, 0, 0});
            held_b[i] = __builtin_bit_cast(bf16x8, s16x8{0, 0, 0, 0, 0, 0, 0, 0});
        }
    }

    // ---- issue state: the next A tile and the next B tile to stage (segment, K-tile inside it, source pointer)
    const int nseg = a.A_lo ? 3 : 1;
    int segA = 0, tA = 0, segB = 0, tB = 0;
    const bf16_t* pA = Ah;
    const bf16_t* pB = Bh;
    auto advA = [&]() {
        ++tA;
        pA += stepA;
        if (tA == nt1) { tA = 0; ++segA; pA = (segA == 2) ? Al : Ah; }
    };
    auto advB = [&]() {
        ++tB;
        pB += stepB;
        if (tB == nt1) { tB = 0; ++segB; pB = (segB == 1) ? Bl : Bh; }
    };
    // generic issue of the next A / B tile into `slot`; q = which half of its 4 DMA instructions (-1: all four).
    // A partial (tail) tile goes through registers in one piece; past the last segment nothing is issued.
    auto issueA = [&](int slot, int q) {
        if ((dbg & 1) || segA >= nseg) return;
        char* dst = smem + slot * TILE_BYTES;
        if (tA < nfull) {
            if (q != 1) stage_dma<0, 2>(pA, voa, dst, wave_u);
            if (q != 0) stage_dma<2, 4>(pA, voa, dst, wave_u);
        } else if (q <= 0) {
            stage_tail<A_RC, BX>(pA, voa, dst, wave, lane, ktail);
        }
    };
    auto issueB = [&](int slot, int q) {
        if ((dbg & 1) || segB >= nseg) return;
        if (IDX == 2 && q <= 0) {  // (generic path: index loads not prefetched)
            load_krows(tB);
            set_vob();
        }
        char* dst = smem + slot * TILE_BYTES;
        if (tB < nfull) {
            if (q != 1) stage_dma<0, 2>(pB, vob, dst, wave_u);
            if (q != 0) stage_dma<2, 4>(pB, vob, dst, wave_u);
        } else if (q <= 0) {
            stage_tail<B_RC, BX>(pB, vob, dst, wave, lane, ktail);
        }
    };
    // is virtual K-tile T staged by DMA (a full tile)?
    auto is_dma = [&](int T) {
        if (T >= nt) return false;
        const int t = T - (T >= nt1 ? nt1 : 0) - (T >= 2 * nt1 ? nt1 : 0);
        return t < nfull;
    };

    // prologue: loads 0 .. 2D-2 = tiles 0 .. D-2 (A and B) and A of tile D-1, into slots 0 .. 2D-2
#pragma unroll
    for (int i = 0; i < D - 1; ++i) {
        issueA(2 * i, -1); advA();
        issueB(2 * i + 1, -1); advB();
    }
    issueA(2 * D - 2, -1); advA();

    // K-tiles [0, nfast): every tile up to D ahead exists and is a full tile -> branch-free steady state
    const int nfast = max(0, ((ktail == 0) ? nt : (nseg == 1 ? nfull : 0)) - D);
    int sa = 0;  // slot of A_T; B_T is in the next one
    auto slot_of = [&](int d) { return sa + d >= NSLOT ? sa + d - NSLOT : sa + d; };
    int T = 0;
#if C2_GEMM_DBG & 16
    unsigned long long wait_clk = 0, bar_clk = 0;
#endif
    if (IDX == 2 && nfast > 0) load_krows(tB);
    for (; T < nfast; ++T) {
        if (IDX == 2) {
            set_vob();                                // offsets of the B tile issued in this iteration (tile tB) ...
            load_krows(min(tB + 1, nt1 - 1));         // ... and the scalar index loads of the next one
        }
        // loads 2T and 2T+1 must have landed; the 2D-3 newer ones stay in flight
#if C2_GEMM_DBG & 16
        const unsigned long long w0 = clock64();
#endif
        wait_loads(2 * D - 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if C2_GEMM_DBG & 16
        const unsigned long long w1 = clock64();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#if C2_GEMM_DBG & 16
        wait_clk += w1 - w0;
        bar_clk += clock64() - w1;
#endif
        char* d3 = smem + slot_of(2 * D - 1) * TILE_BYTES;  // the slots tile T-1 occupied: every wave is past
        char* d4 = smem + slot_of(2 * D) * TILE_BYTES;      // its reads of them
        auto piece = [&](int p) {
            if (dbg & 1) return;
            if (p == 0 && !(dbg & 8)) stage_dma<0, 2>(pB, vob, d3, wave_u);
            if (p == 1 && !(dbg & 8)) stage_dma<2, 4>(pB, vob, d3, wave_u);
            if (p == 2 && !(dbg & 4)) stage_dma<0, 2>(pA, voa, d4, wave_u);
            if (p == 3 && !(dbg & 4)) stage_dma<2, 4>(pA, voa, d4, wave_u);
        };
        if (!(dbg & 2)) {
            if constexpr (ROT)
                compute_tile16_rot<A_RC, B_RC, BX>(smem + sa * TILE_BYTES, smem + slot_of(1) * TILE_BYTES, wm, wn, lane,
                                                   acc16, piece, held_a, held_b);
            else if constexpr (M16)
                compute_tile16<A_RC, B_RC, BX>(smem + sa * TILE_BYTES, smem + slot_of(1) * TILE_BYTES, wm, wn, lane, acc16,
                                               piece);
            else
                compute_tile<A_RC, B_RC, BX, MI>(smem + sa * TILE_BYTES, smem + slot_of(1) * TILE_BYTES, wm, wn, lane, acc,
                                                 piece);
        } else {
            piece(0); piece(1); piece(2); piece(3);
        }
        advB();
        advA();
        sa = slot_of(2);
    }
    if constexpr (ROT) {
        if (!(dbg & 2)) flush_tile16(acc16, held_a, held_b);  // the last steady-state tile's held group (zeros if there was none)
    }
    // the rest: tails, segment ends, drain
    for (; T < nt; ++T) {
        int inflight = 0;
#pragma unroll
        for (int n = 2; n <= 2 * D - 2; ++n) inflight += is_dma(T + (n >> 1)) ? 1 : 0;
        wait_loads(inflight);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int s3 = slot_of(2 * D - 1), s4 = slot_of(2 * D);
        auto piece = [&](int p) {
            if (p < 2) issueB(s3, p);
            else issueA(s4, p - 2);
        };
        if (!(dbg & 2)) {
            if constexpr (M16)
                compute_tile16<A_RC, B_RC, BX>(smem + sa * TILE_BYTES, smem + slot_of(1) * TILE_BYTES, wm, wn, lane, acc16,
                                               piece);
            else
                compute_tile<A_RC, B_RC, BX, MI>(smem + sa * TILE_BYTES, smem + slot_of(1) * TILE_BYTES, wm, wn, lane, acc,
                                                 piece);
        } else {
            piece(0); piece(1); piece(2); piece(3);
        }
        advB();
        advA();
        sa = slot_of(2);
    }

#if C2_GEMM_DBG & 16
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_clk[0] = clock64() - clk0;
        g_clk[1] = wall_clock64() - rt0;
        g_clk[2] = wait_clk;
        g_clk[3] = bar_clk;
    }
#endif
    // ---- epilogue: C = alpha * acc + bias, transposed through LDS for 16-byte-per-lane stores
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done reading the ring
    asm volatile("" ::: "memory");
    if (dbg & 32) {  // (ablation: no epilogue -- one store keeps the accumulators alive)
        if (M16) { if (acc16[0][0][0] == 123.456f) a.C[0] = 1.f; } else if (acc[0][0][0] == 123.456f) a.C[0] = 1.f;
        return;
    }
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
    // (half-precision output without a rank-one term, rows that start on 16 bytes: the eight-column walk below)
    const bool half16 = a.c_f16 && !a.rowscale && ((a.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    const int col8 = n0 + wn * 64 + (lane & 7) * 8;
    float bv8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (half16 && a.bias) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (col8 + c < a.N) bv8[c] = a.bias[col8 + c];
    }
#pragma unroll
    for (int half = 0; half < MI / 2; ++half) {
        if constexpr (M16) {  // 16x16 tiles: lane l holds column l & 15, rows 4 (l >> 4) + 0..3
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        wl[(ii * 16 + 4 * (lane >> 4) + e) * 64 + j * 16 + (lane & 15)] = acc16[half * 4 + ii][j][e];
        } else {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int r = ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        wl[r * 64 + j * 32 + (lane & 31)] = acc[half * 2 + ii][j][e];
                    }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (half16) {
            // output as IEEE halves: eight columns (16 bytes) per lane, eight rows per pass -- 128-byte row segments by
            // 16-byte stores instead of the 8-byte stores the four-column walk would make of them
            const int rowh = m0 + wm * (MI * 32) + half * 64 + (lane >> 3);
            if (C2_EPI_BATCH && __all(col8 + 8 <= a.N && rowh + 56 < a.M)) {
                // an interior wave tile: the eight packed pieces and their addresses first, then eight stores back to back
                // (a store's registers are busy until vmcnt has seen it complete: see store_rows16)
                u32x4 pk[8];
                f16_t* p8[8];
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = rr * 8 + (lane >> 3);
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(wl + r * 64 + (lane & 7) * 8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(wl + r * 64 + (lane & 7) * 8 + 4);
                    pk[rr][0] = pack2<f16_t>(alpha * v0[0] + bv8[0], alpha * v0[1] + bv8[1]);
                    pk[rr][1] = pack2<f16_t>(alpha * v0[2] + bv8[2], alpha * v0[3] + bv8[3]);
                    pk[rr][2] = pack2<f16_t>(alpha * v1[0] + bv8[4], alpha * v1[1] + bv8[5]);
                    pk[rr][3] = pack2<f16_t>(alpha * v1[2] + bv8[6], alpha * v1[3] + bv8[7]);
                    p8[rr] = reinterpret_cast<f16_t*>(C) + (int64_t)(rowh + rr * 8) * a.ldc + col8;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) *reinterpret_cast<u32x4*>(p8[rr]) = pk[rr];
                __builtin_amdgcn_sched_barrier(0);
            } else
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = rr * 8 + (lane >> 3);
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(wl + r * 64 + (lane & 7) * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(wl + r * 64 + (lane & 7) * 8 + 4);
                const int row = m0 + wm * (MI * 32) + half * 64 + r;
                if (row < a.M) {
                    if (col8 + 8 <= a.N) {
                        u32x4 pk;
                        pk[0] = pack2<f16_t>(alpha * v0[0] + bv8[0], alpha * v0[1] + bv8[1]);
                        pk[1] = pack2<f16_t>(alpha * v0[2] + bv8[2], alpha * v0[3] + bv8[3]);
                        pk[2] = pack2<f16_t>(alpha * v1[0] + bv8[4], alpha * v1[1] + bv8[5]);
                        pk[3] = pack2<f16_t>(alpha * v1[2] + bv8[6], alpha * v1[3] + bv8[7]);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<f16_t*>(C) + (int64_t)row * a.ldc + col8) = pk;
                    } else {
                        const float bl[4] = {bv8[0], bv8[1], bv8[2], bv8[3]}, bh[4] = {bv8[4], bv8[5], bv8[6], bv8[7]};
                        store_out4t<false>(a, C, row, col8, v0, alpha, bl, false, 0.f);  // (half16: no rank-one term)
                        store_out4t<false>(a, C, row, col8 + 4, v1, alpha, bh, false, 0.f);
                    }
                }
            }
        } else {
            // (two instances of the walk, chosen by a uniform branch OUTSIDE it; the rank-one factors of the one are fetched
            // before its first store: no load inside either loop -- see store_out4t)
            const int row0 = m0 + wm * (MI * 32) + half * 64 + (lane >> 4);
            auto piece = [&](int rr) { return *reinterpret_cast<const f32x4*>(wl + (rr * 4 + (lane >> 4)) * 64 + (lane & 15) * 4); };
            // an interior wave tile (every piece exists, aligned, fp32): all sixteen stores in flight together
            const bool fast = C2_EPI_BATCH && !a.c_f16 && !(dbg & 64) && __all(vec_ok && row0 + 60 < a.M);
            if (a.rowscale) {
                float rs[16];
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) rs[rr] = (split == 0 && row0 + rr * 4 < a.M) ? a.rowscale[row0 + rr * 4] : 0.f;
                if (fast) store_rows16<16, true>(C + (int64_t)row0 * a.ldc + col4, 4 * a.ldc, alpha, bv, rs, piece);
                else
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(wl + (rr * 4 + (lane >> 4)) * 64 + (lane & 15) * 4);
                    if (row0 + rr * 4 < a.M) store_out4t<true>(a, C, row0 + rr * 4, col4, v, alpha, bv, vec_ok, rs[rr]);
                }
            } else if (fast) {
                store_rows16<16, false>(C + (int64_t)row0 * a.ldc + col4, 4 * a.ldc, alpha, bv, nullptr, piece);
            } else {
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(wl + (rr * 4 + (lane >> 4)) * 64 + (lane & 15) * 4);
                    if (row0 + rr * 4 < a.M && (!(dbg & 64) || v[0] == 123.456f))  // (dbg 64: the LDS trip without the stores)
                        store_out4t<false>(a, C, row0 + rr * 4, col4, v, alpha, bv, vec_ok, 0.f);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next half overwrites
    }
}

// ---- sixteen waves (4 x 4) of 64x64 on the same 256x256 tile and ring: an LDS-DMA instruction holds the issuing wave for
// ~100 cycles; eight waves issue eight per K-tile each, sixteen issue four and leave three partner waves per SIMD to cover
// them.  128 registers per wave: 64 accumulators, fragments single-buffered.  Layouts with a row-wise operand only.
template <int PPW>
__device__ __forceinline__ void wait_pieces(int c) {  // all but the c newest operand-tile loads (PPW pieces each) have landed
    static_assert(PPW == 2, "counted waits are written out for two pieces per tile");
    switch (c) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}
template <bool A_RC, bool B_RC, int BX, typename F>
__device__ __forceinline__ void compute_tile16_64(const char* tA, const char* tB, int wm, int wn, int lane,
                                                  f32x4 (&acc)[4][4], F&& piece) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag16<B_RC, BX>(tB, wn * 64 + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag16<A_RC, BX>(tA, wm * 64 + i * 16, ks, lane);
        piece(2 * ks);
        piece(2 * ks + 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = C2_MFMA_16x16x32(fa[i], fb[j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    }
}

template <bool A_RC, bool B_RC, int IDX = 0>
__global__ __launch_bounds__(1024) void gemm_wide16_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    static_assert(IDX == 0 || (IDX == 1 && A_RC) || (IDX == 2 && !B_RC), "row index on an RC A or an XC B operand");
    static_assert(A_RC || B_RC, "the sixteen-wave form multiplies on 16x16x32: one operand read row-wise");
    constexpr int BX = 256, PPW = 2;        // block tile edge; LDS-DMA pieces of an operand tile per wave
    constexpr int D = 2;
    constexpr int NSLOT = 2 * D + 1;
    constexpr int WN = 4;                   // waves along N (4 x 4 waves of 64x64)
    constexpr int TILE_BYTES = BX * BK * 2; // one operand tile = one ring slot
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // the same number, known to be wave-uniform
    const int wm = wave / WN, wn = wave % WN;
#if C2_GEMM_DBG & 16
    const unsigned long long clk0 = clock64(), rt0 = wall_clock64();
#endif

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
    const int m0 = tm * BX, n0 = tn * BX;
    const int rbeg = split * a.r_per_split;
    const int rend = min(a.R, rbeg + a.r_per_split);
    const int rlen = max(rend - rbeg, 0);
    const int nfull = rlen / BK;
    const int ktail = rlen - nfull * BK;
    const int nt1 = nfull + (ktail ? 1 : 0);        // K-tiles of one segment
    const int nt = a.A_lo ? 3 * nt1 : nt1;          // bf16x3: segments (hi,hi) (hi,lo) (lo,hi)

    // uniform tile bases (scalar registers) + 32-bit per-lane offsets
    const int64_t offA = IDX == 1 ? (int64_t)rbeg : (A_RC ? (int64_t)m0 * a.lda + rbeg : (int64_t)rbeg * a.lda + m0);
    const int64_t offB = IDX == 2 ? (int64_t)n0 : (B_RC ? (int64_t)n0 * a.ldb + rbeg : (int64_t)rbeg * a.ldb + n0);
    const bf16_t* Ah = reinterpret_cast<const bf16_t*>(a.A) + offA;
    const bf16_t* Bh = reinterpret_cast<const bf16_t*>(a.B) + offB;
    const bf16_t* Al = reinterpret_cast<const bf16_t*>(a.A_lo) + offA;
    const bf16_t* Bl = reinterpret_cast<const bf16_t*>(a.B_lo) + offB;
    int voa[4], vob[4];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        voa[j] = stage_offset<A_RC, BX, PPW>(j, a.lda, m0, a.M, wave, lane);
        vob[j] = stage_offset<B_RC, BX, PPW>(j, a.ldb, n0, a.N, wave, lane);
    }
    const int64_t stepA = A_RC ? BK : (int64_t)BK * a.lda;
    const int64_t stepB = IDX == 2 ? 0 : (B_RC ? BK : (int64_t)BK * a.ldb);
    if (IDX == 1) {  // row part of A's offsets: the indexed row instead of the tile row (edge rows re-read the last row)
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int row = 8 * (PPW * wave + j) + (lane >> 3);
            const int x = min(row, a.M - 1 - m0);
            voa[j] += (a.rows[m0 + x] - x) * (int)a.lda;
        }
    }
    // IDX 2: vob[j] = rows[k-row] * ldb + xob[j]; the DMA instruction j of this wave covers k-rows 4*wave + 2j + {0,1}
    int xob[4];
    int kr[2 * PPW];  // the wave's four row indices of the B tile that is issued next
    auto load_krows = [&](int t) {
        const int32_t* rp = a.rows + rbeg + t * BK + 2 * PPW * wave_u;  // wave-uniform address: scalar loads
#pragma unroll
        for (int q = 0; q < 2 * PPW; ++q) kr[q] = rp[q];
    };
    auto set_vob = [&]() {
#pragma unroll
        for (int j = 0; j < PPW; ++j) vob[j] = ((lane >> 5) ? kr[2 * j + 1] : kr[2 * j]) * (int)a.ldb + xob[j];
    };
    if (IDX == 2) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int k = 2 * PPW * wave + 2 * j + (lane >> 5);
            xob[j] = vob[j] - k * (int)a.ldb;  // the column part stage_offset<XC> computed
        }
    }

    f32x4 acc16[4][4];  // 64x64 wave tile as 4 x 4 tiles of 16x16
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- issue state: the next A tile and the next B tile to stage (segment, K-tile inside it, source pointer)
    const int nseg = a.A_lo ? 3 : 1;
    int segA = 0, tA = 0, segB = 0, tB = 0;
    const bf16_t* pA = Ah;
    const bf16_t* pB = Bh;
    auto advA = [&]() {
        ++tA;
        pA += stepA;
        if (tA == nt1) { tA = 0; ++segA; pA = (segA == 2) ? Al : Ah; }
    };
    auto advB = [&]() {
        ++tB;
        pB += stepB;
        if (tB == nt1) { tB = 0; ++segB; pB = (segB == 1) ? Bl : Bh; }
    };
    // generic issue of the next A / B tile into `slot`; q = which half of its 4 DMA instructions (-1: all four).
    // A partial (tail) tile goes through registers in one piece; past the last segment nothing is issued.
    auto issueA = [&](int slot, int q) {
        if ((dbg & 1) || segA >= nseg) return;
        char* dst = smem + slot * TILE_BYTES;
        if (tA < nfull) {
            if (q != 1) stage_dma<0, 1, PPW>(pA, voa, dst, wave_u);
            if (q != 0) stage_dma<1, 2, PPW>(pA, voa, dst, wave_u);
        } else if (q <= 0) {
            stage_tail<A_RC, BX, PPW>(pA, voa, dst, wave, lane, ktail);
        }
    };
    auto issueB = [&](int slot, int q) {
        if ((dbg & 1) || segB >= nseg) return;
        if (IDX == 2 && q <= 0) {  // (generic path: index loads not prefetched)
            load_krows(tB);
            set_vob();
        }
        char* dst = smem + slot * TILE_BYTES;
        if (tB < nfull) {
            if (q != 1) stage_dma<0, 1, PPW>(pB, vob, dst, wave_u);
            if (q != 0) stage_dma<1, 2, PPW>(pB, vob, dst, wave_u);
        } else if (q <= 0) {
            stage_tail<B_RC, BX, PPW>(pB, vob, dst, wave, lane, ktail);
        }
    };
    // is virtual K-tile T staged by DMA (a full tile)?
    auto is_dma = [&](int T) {
        if (T >= nt) return false;
        const int t = T - (T >= nt1 ? nt1 : 0) - (T >= 2 * nt1 ? nt1 : 0);
        return t < nfull;
    };

    // prologue: loads 0 .. 2D-2 = tiles 0 .. D-2 (A and B) and A of tile D-1, into slots 0 .. 2D-2
#pragma unroll
    for (int i = 0; i < D - 1; ++i) {
        issueA(2 * i, -1); advA();
        issueB(2 * i + 1, -1); advB();
    }
    issueA(2 * D - 2, -1); advA();

    // K-tiles [0, nfast): every tile up to D ahead exists and is a full tile -> branch-free steady state
    const int nfast = max(0, ((ktail == 0) ? nt : (nseg == 1 ? nfull : 0)) - D);
    int sa = 0;  // slot of A_T; B_T is in the next one
    auto slot_of = [&](int d) { return sa + d >= NSLOT ? sa + d - NSLOT : sa + d; };
    int T = 0;
#if C2_GEMM_DBG & 16
    unsigned long long wait_clk = 0, bar_clk = 0;
#endif
    if (IDX == 2 && nfast > 0) load_krows(tB);
    for (; T < nfast; ++T) {
        if (IDX == 2) {
            set_vob();                                // offsets of the B tile issued in this iteration (tile tB) ...
            load_krows(min(tB + 1, nt1 - 1));         // ... and the scalar index loads of the next one
        }
        // loads 2T and 2T+1 must have landed; the 2D-3 newer ones stay in flight
#if C2_GEMM_DBG & 16
        const unsigned long long w0 = clock64();
#endif
        wait_pieces<PPW>(2 * D - 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if C2_GEMM_DBG & 16
        const unsigned long long w1 = clock64();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#if C2_GEMM_DBG & 16
        wait_clk += w1 - w0;
        bar_clk += clock64() - w1;
#endif
        char* d3 = smem + slot_of(2 * D - 1) * TILE_BYTES;  // the slots tile T-1 occupied: every wave is past
        char* d4 = smem + slot_of(2 * D) * TILE_BYTES;      // its reads of them
        auto piece = [&](int p) {
            if (dbg & 1) return;
            if (p == 0 && !(dbg & 8)) stage_dma<0, 1, PPW>(pB, vob, d3, wave_u);
            if (p == 1 && !(dbg & 8)) stage_dma<1, 2, PPW>(pB, vob, d3, wave_u);
            if (p == 2 && !(dbg & 4)) stage_dma<0, 1, PPW>(pA, voa, d4, wave_u);
            if (p == 3 && !(dbg & 4)) stage_dma<1, 2, PPW>(pA, voa, d4, wave_u);
        };
        if (!(dbg & 2)) {
            compute_tile16_64<A_RC, B_RC, BX>(smem + sa * TILE_BYTES, smem + slot_of(1) * TILE_BYTES, wm, wn, lane, acc16,
                                              piece);
        } else {
            piece(0); piece(1); piece(2); piece(3);
        }
        advB();
        advA();
        sa = slot_of(2);
    }
    // the rest: tails, segment ends, drain
    for (; T < nt; ++T) {
        int inflight = 0;
#pragma unroll
        for (int n = 2; n <= 2 * D - 2; ++n) inflight += is_dma(T + (n >> 1)) ? 1 : 0;
        wait_pieces<PPW>(inflight);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int s3 = slot_of(2 * D - 1), s4 = slot_of(2 * D);
        auto piece = [&](int p) {
            if (p < 2) issueB(s3, p);
            else issueA(s4, p - 2);
        };
        if (!(dbg & 2)) {
            compute_tile16_64<A_RC, B_RC, BX>(smem + sa * TILE_BYTES, smem + slot_of(1) * TILE_BYTES, wm, wn, lane, acc16,
                                              piece);
        } else {
            piece(0); piece(1); piece(2); piece(3);
        }
        advB();
        advA();
        sa = slot_of(2);
    }

#if C2_GEMM_DBG & 16
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_clk[0] = clock64() - clk0;
        g_clk[1] = wall_clock64() - rt0;
        g_clk[2] = wait_clk;
        g_clk[3] = bar_clk;
    }
#endif
    // ---- epilogue: C = alpha * acc + bias, transposed through LDS for 16-byte-per-lane stores
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done reading the ring
    asm volatile("" ::: "memory");
    const float alpha = a.alpha_dev ? *a.alpha_dev : 1.f;
    float* C = a.C + (int64_t)split * a.slab_stride;
    float* wl = reinterpret_cast<float*>(smem + wave * 8192);  // [32 rows][64 cols] fp32, private to the wave
    const int col4 = n0 + wn * 64 + (lane & 15) * 4;
    const bool vec_ok = ((a.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (col4 + 4 <= a.N);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (col4 + c < a.N) bv[c] = a.bias[col4 + c];
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {  // the wave's 64 rows in two quarters of 32: sixteen waves x 8 KB fit the LDS
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    wl[(ii * 16 + 4 * (lane >> 4) + e) * 64 + j * 16 + (lane & 15)] = acc16[qt * 2 + ii][j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int row0 = m0 + wm * 64 + qt * 32 + (lane >> 4);
        auto piece = [&](int rr) { return *reinterpret_cast<const f32x4*>(wl + (rr * 4 + (lane >> 4)) * 64 + (lane & 15) * 4); };
        const bool fast = C2_EPI_BATCH && !a.c_f16 && __all(vec_ok && row0 + 28 < a.M);  // (see store_rows16)
        if (a.rowscale) {  // (no load inside either store loop: see store_out4t)
            float rs[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) rs[rr] = (split == 0 && row0 + rr * 4 < a.M) ? a.rowscale[row0 + rr * 4] : 0.f;
            if (fast) store_rows16<8, true>(C + (int64_t)row0 * a.ldc + col4, 4 * a.ldc, alpha, bv, rs, piece);
            else
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                f32x4 v = *reinterpret_cast<const f32x4*>(wl + (rr * 4 + (lane >> 4)) * 64 + (lane & 15) * 4);
                if (row0 + rr * 4 < a.M) store_out4t<true>(a, C, row0 + rr * 4, col4, v, alpha, bv, vec_ok, rs[rr]);
            }
        } else if (fast) {
            store_rows16<8, false>(C + (int64_t)row0 * a.ldc + col4, 4 * a.ldc, alpha, bv, nullptr, piece);
        } else {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                f32x4 v = *reinterpret_cast<const f32x4*>(wl + (rr * 4 + (lane >> 4)) * 64 + (lane & 15) * 4);
                if (row0 + rr * 4 < a.M) store_out4t<false>(a, C, row0 + rr * 4, col4, v, alpha, bv, vec_ok, 0.f);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next quarter overwrites
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16x3 on planes, fused: ONE pass over the reduction in which every K-tile brings its four operand tiles
// (A_hi, B_hi, A_lo, B_lo; 16 KB each) and is multiplied three times (hi*hi + hi*lo + lo*hi) from the same
// fragments -- 64 KB staged per 48 MFMAs of a wave instead of 96 KB per 48 (three passes over a virtual reduction,
// each with its own pair of tiles), and a third as many barrier / LDS-DMA round trips, which is what bounds the
// narrow layers (one 128x128 block per CU, 24 K-tiles of 16 MFMAs each before).  128x128 tile, 4 waves (2 x 2), each
// wave 64x64; two stages of four slots (128 KB): while K-tile T is multiplied, T+1 is in flight.
// NWX = 16 (default; C2_X3_WAVES=4 / 8 for A/B): an LDS-DMA instruction holds the issuing wave for ~100 cycles, and with
// four waves of 64x64 each one issues sixteen of them per K-tile, longer than its 48 MFMAs take.  Eight waves of 32x64
// issue eight each, two waves per SIMD: ~10 % faster on every narrow shape (22.3 -> 20.1 us at 4096 x 512 x 1024, C4 step
// 1.840 -> 1.810 ms); sixteen waves of 32x32 issue four each, four waves per SIMD: a further 1-9 % per launch (19.8 us;
// 29.6 -> 27.0 us on the 1024-wide input gradient; C4 step 1.836 -> 1.829 ms on the box that compared them).
// (Measured and not kept: the next tile's DMA issued one operand tile per reduction step in front of that step's MFMAs
// instead of all up front: 20.1 -> 22.3 us, 29.6 -> 37.8 us on the 1024-wide input gradient.)
template <bool A_RC, bool B_RC, int NWX>
__global__ __launch_bounds__(64 * NWX) void gemm_x3_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    // waves: 4 = 2 x 2 of 64x64, 8 = 4 x 2 of 32x64, 16 = 4 x 4 of 32x32
    constexpr int BX = 128, MI = NWX == 4 ? 2 : 1, NJ = NWX == 16 ? 1 : 2, WN = NWX == 16 ? 4 : 2, PPW = 16 / NWX;
    constexpr int WCOLS = NJ * 32;
    constexpr int TILE_BYTES = BX * BK * 2;  // 16 KB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WN, wn = wave % WN;
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
    const int m0 = tm * BX, n0 = tn * BX;
    const int rbeg = split * a.r_per_split;
    const int rend = min(a.R, rbeg + a.r_per_split);
    const int rlen = max(rend - rbeg, 0);
    const int nfull = rlen / BK;
    const int ktail = rlen - nfull * BK;
    const int nt = nfull + (ktail ? 1 : 0);

    const int64_t offA = A_RC ? (int64_t)m0 * a.lda + rbeg : (int64_t)rbeg * a.lda + m0;
    const int64_t offB = B_RC ? (int64_t)n0 * a.ldb + rbeg : (int64_t)rbeg * a.ldb + n0;
    const bf16_t* pAh = reinterpret_cast<const bf16_t*>(a.A) + offA;
    const bf16_t* pBh = reinterpret_cast<const bf16_t*>(a.B) + offB;
    const bf16_t* pAl = reinterpret_cast<const bf16_t*>(a.A_lo) + offA;
    const bf16_t* pBl = reinterpret_cast<const bf16_t*>(a.B_lo) + offB;
    int voa[4], vob[4];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        voa[j] = stage_offset<A_RC, BX, PPW>(j, a.lda, m0, a.M, wave, lane);
        vob[j] = stage_offset<B_RC, BX, PPW>(j, a.ldb, n0, a.N, wave, lane);
    }
    const int64_t stepA = A_RC ? BK : (int64_t)BK * a.lda;
    const int64_t stepB = B_RC ? BK : (int64_t)BK * a.ldb;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // stage s (0/1) = slots 4s .. 4s+3 = (A_hi, B_hi, A_lo, B_lo) of one K-tile
    auto issue = [&](int t, int s) {
        char* d = smem + 4 * s * TILE_BYTES;
        if (t < nfull) {
            stage_dma<0, PPW, PPW>(pAh, voa, d, wave_u);
            stage_dma<0, PPW, PPW>(pBh, vob, d + TILE_BYTES, wave_u);
            stage_dma<0, PPW, PPW>(pAl, voa, d + 2 * TILE_BYTES, wave_u);
            stage_dma<0, PPW, PPW>(pBl, vob, d + 3 * TILE_BYTES, wave_u);
        } else {  // the partial last tile: through registers, zero filled
            stage_tail<A_RC, BX, PPW>(pAh, voa, d, wave, lane, ktail);
            stage_tail<B_RC, BX, PPW>(pBh, vob, d + TILE_BYTES, wave, lane, ktail);
            stage_tail<A_RC, BX, PPW>(pAl, voa, d + 2 * TILE_BYTES, wave, lane, ktail);
            stage_tail<B_RC, BX, PPW>(pBl, vob, d + 3 * TILE_BYTES, wave, lane, ktail);
        }
        pAh += stepA; pAl += stepA; pBh += stepB; pBl += stepB;
    };
    if (nt > 0) issue(0, 0);
    for (int T = 0; T < nt; ++T) {
        // tile T (issued one iteration ago) has landed; every wave is past its reads of tile T-1's stage
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (T + 1 < nt) issue(T + 1, (T + 1) & 1);
        const char* d = smem + 4 * (T & 1) * TILE_BYTES;
        const char *tAh = d, *tBh = d + TILE_BYTES, *tAl = d + 2 * TILE_BYTES, *tBl = d + 3 * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 ah[MI], al[MI], bh[NJ], bl[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                bh[j] = read_frag<B_RC, BX>(tBh, wn * WCOLS + j * 32, ks, lane);
                bl[j] = read_frag<B_RC, BX>(tBl, wn * WCOLS + j * 32, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = read_frag<A_RC, BX>(tAh, wm * (MI * 32) + i * 32, ks, lane);
                al[i] = read_frag<A_RC, BX>(tAl, wm * (MI * 32) + i * 32, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = C2_MFMA_32x32x16(ah[i], bh[j], acc[i][j]);
                    acc[i][j] = C2_MFMA_32x32x16(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = C2_MFMA_32x32x16(al[i], bh[j], acc[i][j]);
                }
        }
    }

    // ---- epilogue (as gemm_wide_kernel): accumulators transposed through LDS for 16-byte-per-lane stores
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float alpha = a.alpha_dev ? *a.alpha_dev : 1.f;
    float* C = a.C + (int64_t)split * a.slab_stride;
    constexpr int LPR = WCOLS / 4, RPP = 64 / LPR;  // lanes per row of the wave tile, rows per pass of 64 lanes
    float* wl = reinterpret_cast<float*>(smem + wave * (MI * 32 * WCOLS * 4));  // [MI * 32 rows][WCOLS] fp32, wave-private
    const int col4 = n0 + wn * WCOLS + (lane % LPR) * 4;
    const bool vec_ok = ((a.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (col4 + 4 <= a.N);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (col4 + c < a.N) bv[c] = a.bias[col4 + c];
    }
#pragma unroll
    for (int ii = 0; ii < MI; ++ii)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                wl[r * WCOLS + j * 32 + (lane & 31)] = acc[ii][j][e];
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int NRR = MI * 32 / RPP;
    const int row0 = m0 + wm * (MI * 32) + lane / LPR;
    auto piece = [&](int rr) { return *reinterpret_cast<const f32x4*>(wl + (rr * RPP + lane / LPR) * WCOLS + (lane % LPR) * 4); };
    const bool fast = C2_EPI_BATCH && !a.c_f16 && __all(vec_ok && row0 + (NRR - 1) * RPP < a.M);  // (see store_rows16)
    if (a.rowscale) {  // (no load inside either store loop: see store_out4t)
        float rs[NRR];
#pragma unroll
        for (int rr = 0; rr < NRR; ++rr) rs[rr] = (split == 0 && row0 + rr * RPP < a.M) ? a.rowscale[row0 + rr * RPP] : 0.f;
        if (fast) store_rows16<NRR, true>(C + (int64_t)row0 * a.ldc + col4, (int64_t)RPP * a.ldc, alpha, bv, rs, piece);
        else
#pragma unroll
        for (int rr = 0; rr < NRR; ++rr) {
            f32x4 v = *reinterpret_cast<const f32x4*>(wl + (rr * RPP + lane / LPR) * WCOLS + (lane % LPR) * 4);
            if (row0 + rr * RPP < a.M) store_out4t<true>(a, C, row0 + rr * RPP, col4, v, alpha, bv, vec_ok, rs[rr]);
        }
    } else if (fast) {
        store_rows16<NRR, false>(C + (int64_t)row0 * a.ldc + col4, (int64_t)RPP * a.ldc, alpha, bv, nullptr, piece);
    } else {
#pragma unroll
        for (int rr = 0; rr < NRR; ++rr) {
            f32x4 v = *reinterpret_cast<const f32x4*>(wl + (rr * RPP + lane / LPR) * WCOLS + (lane % LPR) * 4);
            if (row0 + rr * RPP < a.M) store_out4t<false>(a, C, row0 + rr * RPP, col4, v, alpha, bv, vec_ok, 0.f);
        }
    }
}

}  // namespace C2_WIDE_NS

#if C2_GEMM_DBG & 16
extern "C" int c2_dbg_clk(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(C2_WIDE_NS::g_clk), 32);
}
#endif

// 0 = not taken by this file; 256 / 128 = block-tile edge.
int wide_tile(int dtype, const GemmArgs& a) {
#ifdef C2_WIDE_F16
    if (dtype != C2_F16) return 0;
#else
    if (dtype != C2_BF16 && dtype != C2_BF16X3P) return 0;
#endif
    if (!aligned16(a.A) || !aligned16(a.B) || (a.lda & 7) || (a.ldb & 7) || (a.R & 7)) return 0;
    if (dtype == C2_BF16X3P) {
        if (!aligned16(a.A_lo) || !aligned16(a.B_lo)) return 0;
        return (a.R >= 64 && a.M >= 1 && a.N >= 1) ? 128 : 0;
    }
    // worth a 256x256 tile: a gene-width problem
    if ((int64_t)a.M * a.N >= 256ll * 256 * 32 && a.R >= 256 && a.M >= 192 && a.N >= 192) return 256;
    // short-and-wide bf16 problems (the gene-width layers at the reference's default batch of 128): the 128x128
    // LDS-DMA instance instead of the register-staged kernel (C2_GEMM_BF16_128=0: A/B)
    static const bool small_on = []() { const char* e = getenv("C2_GEMM_BF16_128"); return !(e && e[0] == '0'); }();
    if (small_on && a.R >= 128 && (int64_t)a.M * a.N >= 128ll * 128 * 2) return 128;
    return 0;
}

// Which layouts run the sixteen-wave form of the 256x256 kernel.  Measured (tools/bench_gemm.py, same box): (XC, RC) -- the
// decoder's weight gradient with its narrow operand transposed -- 207.6 -> 193.9 us; forward 216.4 -> 214.8 and 194.9 ->
// 200.9, input gradient 216.3 -> 216.1, row-indexed (RC, XC) 214.5 -> 217.5: a wash.  So layout 3 only by default;
// C2_WIDE_WAVES=16 / 8 puts every eligible layout / none on it (A/B).
static bool wide16_on(int layout) {
    static const int forced = []() { const char* e = getenv("C2_WIDE_WAVES"); return e ? atoi(e) : 0; }();
    if (forced == 16) return true;
    if (forced == 8) return false;
    return layout == 3;
}
template <bool A_RC, bool B_RC, int IDX>
static void launch_w16(const GemmArgs& a, int nslab, hipStream_t stream) {
    using namespace C2_WIDE_NS;
    constexpr int LDS = 5 * 256 * BK * 2;
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_wide16_kernel<A_RC, B_RC, IDX>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_wide16_kernel<A_RC, B_RC, IDX>), dim3(tiles_m * tiles_n * nslab), dim3(1024), LDS, stream, a,
                       tiles_m, tiles_n);
}

template <int NW>
static void launch_nw(int layout, const GemmArgs& a, int nslab, hipStream_t stream) {
    using namespace C2_WIDE_NS;
    if constexpr (NW == 8) {
        if (layout != 2 && wide16_on(layout)) {
            if (layout == 0) launch_w16<true, true, 0>(a, nslab, stream);
            else if (layout == 1) launch_w16<true, false, 0>(a, nslab, stream);
            else launch_w16<false, true, 0>(a, nslab, stream);
            return;
        }
    }
    constexpr int BX = 32 * NW;
    constexpr int LDS = (NW == 8 ? 5 : 9) * BX * BK * 2;
    const int tiles_m = (a.M + BX - 1) / BX, tiles_n = (a.N + BX - 1) / BX;
    const dim3 grid(tiles_m * tiles_n * nslab);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<true, true, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<true, false, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<false, false, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    if constexpr (NW == 4) {
        // short reductions on more blocks than CUs: the shallow ring, two blocks per CU (C2_WIDE_SHALLOW=0: A/B)
        static const bool shallow_on = []() { const char* e = getenv("C2_WIDE_SHALLOW"); return !(e && e[0] == '0'); }();
        static const int shallow_kt = []() { const char* e = getenv("C2_WIDE_SHALLOW_KT"); return e ? atoi(e) : 8; }();
        const int kt = ((a.R < a.r_per_split ? a.R : a.r_per_split) + BK - 1) / BK * (a.A_lo ? 3 : 1);
        if (shallow_on && layout != 3 && grid.x > 256 && kt <= shallow_kt) {
            constexpr int LDS2 = 5 * BX * BK * 2;
            static bool attr2 = false;
            if (!attr2) {
                (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<true, true, 4, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
                (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<true, false, 4, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
                (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<false, false, 4, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
                attr2 = true;
            }
            if (layout == 0)
                hipLaunchKernelGGL((gemm_wide_kernel<true, true, 4, 0, 2>), grid, dim3(256), LDS2, stream, a, tiles_m, tiles_n);
            else if (layout == 1)
                hipLaunchKernelGGL((gemm_wide_kernel<true, false, 4, 0, 2>), grid, dim3(256), LDS2, stream, a, tiles_m, tiles_n);
            else
                hipLaunchKernelGGL((gemm_wide_kernel<false, false, 4, 0, 2>), grid, dim3(256), LDS2, stream, a, tiles_m, tiles_n);
            return;
        }
    }
    if (layout == 0)
        hipLaunchKernelGGL((gemm_wide_kernel<true, true, NW>), grid, dim3(64 * NW), LDS, stream, a, tiles_m, tiles_n);
    else if (layout == 1)
        hipLaunchKernelGGL((gemm_wide_kernel<true, false, NW>), grid, dim3(64 * NW), LDS, stream, a, tiles_m, tiles_n);
    else if (layout == 3) {  // (XC, RC): a weight gradient whose second operand was handed over transposed
        if constexpr (NW == 8) {
            static bool attr3 = false;
            if (!attr3) {
                (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<false, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                attr3 = true;
            }
            hipLaunchKernelGGL((gemm_wide_kernel<false, true, 8>), grid, dim3(64 * NW), LDS, stream, a, tiles_m, tiles_n);
        }
    } else
        hipLaunchKernelGGL((gemm_wide_kernel<false, false, NW>), grid, dim3(64 * NW), LDS, stream, a, tiles_m, tiles_n);
}

int launch_gemm_wide_rows(int layout, const GemmArgs& a, int nslab, hipStream_t stream) {
    using namespace C2_WIDE_NS;
    if (!a.rows || a.A_lo || (layout != 0 && layout != 2 && layout != 4) || wide_tile(C2_WIDE_DT, a) != 256) return 3;
    if (layout != 2 && wide16_on(layout)) {
        if (layout == 0) launch_w16<true, true, 1>(a, nslab, stream);
        else launch_w16<true, false, 2>(a, nslab, stream);
        return 0;
    }
    constexpr int NW = 8, BX = 256;
    constexpr int LDS = 5 * BX * BK * 2;
    const int tiles_m = (a.M + BX - 1) / BX, tiles_n = (a.N + BX - 1) / BX;
    const dim3 grid(tiles_m * tiles_n * nslab);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<true, true, NW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<false, false, NW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<true, false, NW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    if (layout == 0)
        hipLaunchKernelGGL((gemm_wide_kernel<true, true, NW, 1>), grid, dim3(64 * NW), LDS, stream, a, tiles_m, tiles_n);
    else if (layout == 4)  // (RC, XC with indexed reduction rows): the weight gradient with dY handed over transposed
        hipLaunchKernelGGL((gemm_wide_kernel<true, false, NW, 2>), grid, dim3(64 * NW), LDS, stream, a, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL((gemm_wide_kernel<false, false, NW, 2>), grid, dim3(64 * NW), LDS, stream, a, tiles_m, tiles_n);
    return 0;
}

template <int NWX>
static void launch_x3_nw(int layout, const GemmArgs& a, int nslab, hipStream_t stream) {
    using namespace C2_WIDE_NS;
    constexpr int BX = 128;
    constexpr int LDS = 8 * BX * BK * 2;  // two stages of four 16 KB slots
    const int tiles_m = (a.M + BX - 1) / BX, tiles_n = (a.N + BX - 1) / BX;
    const dim3 grid(tiles_m * tiles_n * nslab);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_x3_kernel<true, true, NWX>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_x3_kernel<true, false, NWX>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_x3_kernel<false, false, NWX>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    if (layout == 0)
        hipLaunchKernelGGL((gemm_x3_kernel<true, true, NWX>), grid, dim3(64 * NWX), LDS, stream, a, tiles_m, tiles_n);
    else if (layout == 1)
        hipLaunchKernelGGL((gemm_x3_kernel<true, false, NWX>), grid, dim3(64 * NWX), LDS, stream, a, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL((gemm_x3_kernel<false, false, NWX>), grid, dim3(64 * NWX), LDS, stream, a, tiles_m, tiles_n);
}
static void launch_x3(int layout, const GemmArgs& a, int nslab, hipStream_t stream) {
    static const int waves = []() { const char* e = getenv("C2_X3_WAVES"); return e ? atoi(e) : 16; }();
    if (waves == 4) launch_x3_nw<4>(layout, a, nslab, stream);
    else if (waves == 8) launch_x3_nw<8>(layout, a, nslab, stream);
    else launch_x3_nw<16>(layout, a, nslab, stream);
}

int launch_gemm_wide(int layout, int tile, const GemmArgs& a, int nslab, hipStream_t stream) {
    static const bool x3_passes = getenv("C2_X3_PASSES") != nullptr;  // A/B: the three-pass instance
    if (tile == 256) launch_nw<8>(layout, a, nslab, stream);
    else if (a.A_lo && !x3_passes) launch_x3(layout, a, nslab, stream);
    else launch_nw<4>(layout, a, nslab, stream);
    return 0;
}

}  // namespace c2
