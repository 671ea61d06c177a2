// MFMA GEMMs of the Linear layers (forward, dgrad, wgrad) for gfx950.
//
// One kernel template serves the three layouts.  C[M,N] = sum_r Aop[i][r] * Bop[j][r], where each
// operand is either "RC" (reduction index contiguous in memory: element (x,r) at p[x*ld + r]) or
// "XC" (output index contiguous: element (x,r) at p[r*ld + x]).  forward = (RC,RC), dgrad = (RC,XC),
// wgrad = (XC,XC).  Tiles keep their GLOBAL orientation in LDS, so global loads and LDS writes are
// 16-byte and coalesced for every layout; the transposition an XC operand needs is done by the LDS read:
//   bf16: ds_read_b64_tr_b16 (gfx950 transposing read) feeds v_mfma_f32_32x32x16_bf16;
//   fp32: v_mfma_f32_32x32x2_f32 takes one float per lane, any orientation reads conflict-free.
// 128x128 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32, fp32 accumulate.
// Register-staged pipeline: the global loads of K-tile t+1 are in flight while tile t is multiplied.
// Split reductions write fp32 slabs (deterministic; the consumer kernel sums them).
#include <stdlib.h>

#include <type_traits>

#include "gemm.h"

namespace c2 {

struct x3_t {};  // tag: fp32 operands in memory, multiplied as three bf16 MFMA products (hi*hi + hi*lo + lo*hi)

template <typename T>
struct Cfg;
template <>
struct Cfg<bf16_t> {
    typedef bf16_t GT;  // element type in global memory
    typedef bf16_t LT;  // element type in LDS
    static constexpr int BK = 64, VEC = 8, NTILE = 1;
    static constexpr int S_RC = 72;   // [128][72] bf16: 144-B rows, ds_read_b128 conflict-free
    static constexpr int S_XC = 160;  // [64][160] bf16: 320-B rows, 4 k-rows hit 4 distinct 64-B bank groups
};
struct f16_tag {};  // tag: IEEE half operands (raw 16-bit words, staged like bf16; v_mfma_f32_32x32x16_f16)
template <>
struct Cfg<f16_tag> : Cfg<bf16_t> {};
template <>
struct Cfg<float> {
    typedef float GT;
    typedef float LT;
    static constexpr int BK = 32, VEC = 4, NTILE = 1;
    static constexpr int S_RC = 33;   // [128][33] f32
    static constexpr int S_XC = 132;  // [32][132] f32
};
template <>
struct Cfg<x3_t> {
    typedef float GT;
    typedef bf16_t LT;
    static constexpr int BK = 32, VEC = 4, NTILE = 2;  // hi and lo bf16 tiles per operand
    static constexpr int S_RC = 40;   // [128][40] bf16: 80-B rows
    static constexpr int S_XC = 160;  // [32][160] bf16
};

constexpr int BM = 128, BN = 128, NT = 256;

template <typename T, bool RC>
__host__ __device__ constexpr int tile_elems() {
    return ((RC ? BM * Cfg<T>::S_RC : Cfg<T>::BK * Cfg<T>::S_XC) + 7) & ~7;
}

// 16 bytes of operand data held in registers between the global load and the LDS write
struct Vec16 {
    u32x4 v;
};

template <typename T, bool RC, bool ALIGNED>
__device__ __forceinline__ void load_tile(const typename Cfg<T>::GT* __restrict__ base, int64_t ld, int x0, int xtot,
                                          int r0, int rend, Vec16 (&regs)[4]) {
    typedef typename Cfg<T>::GT GT;
    constexpr int VEC = Cfg<T>::VEC, BK = Cfg<T>::BK;
    constexpr int VPR = RC ? (BK / VEC) : (BM / VEC);  // vectors per LDS row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = threadIdx.x + NT * i;
        const int row = v / VPR, cv = v % VPR;
        int x, r;
        int64_t off;
        bool full, any;
        if (RC) {
            x = x0 + row;
            r = r0 + cv * VEC;
            off = (int64_t)x * ld + r;
            any = (x < xtot) && (r < rend);
            full = any && (r + VEC <= rend);
        } else {
            r = r0 + row;
            x = x0 + cv * VEC;
            off = (int64_t)r * ld + x;
            any = (r < rend) && (x < xtot);
            full = any && (x + VEC <= xtot);
        }
        u32x4 val = {0u, 0u, 0u, 0u};
        if (ALIGNED && full) {
            val = *reinterpret_cast<const u32x4*>(base + off);
        } else if (any) {
            const int lim = RC ? (rend - r) : (xtot - x);
            GT tmp[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) tmp[e] = (e < lim) ? base[off + e] : GT(0);
            val = *reinterpret_cast<u32x4*>(tmp);
        }
        regs[i].v = val;
    }
}

template <typename T, bool RC>
__device__ __forceinline__ void store_tile(typename Cfg<T>::LT* __restrict__ lds, const Vec16 (&regs)[4]) {
    constexpr int VEC = Cfg<T>::VEC, BK = Cfg<T>::BK;
    constexpr int VPR = RC ? (BK / VEC) : (BM / VEC);
    constexpr int S = RC ? Cfg<T>::S_RC : Cfg<T>::S_XC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = threadIdx.x + NT * i;
        const int row = v / VPR, cv = v % VPR;
        if constexpr (Cfg<T>::NTILE == 2) {  // split each fp32 into hi + lo bf16, two tiles
            const float* f = reinterpret_cast<const float*>(&regs[i].v);
            bf16_t hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = f2bf(f[e]);
                lo[e] = f2bf(f[e] - bf2f(hi[e]));
            }
            bf16_t* dst = lds + row * S + cv * VEC;
            u32x2 ph, pl;
            ph[0] = (unsigned)hi[0] | ((unsigned)hi[1] << 16);
            ph[1] = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
            pl[0] = (unsigned)lo[0] | ((unsigned)lo[1] << 16);
            pl[1] = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
            *reinterpret_cast<u32x2*>(dst) = ph;
            *reinterpret_cast<u32x2*>(dst + tile_elems<T, RC>()) = pl;
        } else {
            typename Cfg<T>::LT* dst = lds + row * S + cv * VEC;
            if (sizeof(typename Cfg<T>::LT) == 4 && RC) {  // stride 33: scalar stores
                const float* sf = reinterpret_cast<const float*>(&regs[i].v);
                float* d = reinterpret_cast<float*>(dst);
                d[0] = sf[0];
                d[1] = sf[1];
                d[2] = sf[2];
                d[3] = sf[3];
            } else {
                *reinterpret_cast<u32x4*>(dst) = regs[i].v;
            }
        }
    }
}

// bf16 fragment (8 values along the reduction for row/col x = xt + (lane&31), k = 8*(lane>>5)+j)
template <bool RC, int S>
__device__ __forceinline__ bf16x8 read_frag_bf16(const bf16_t* lds, int xt, int r0, int lane) {
    if (RC) {
        const bf16_t* p = lds + (xt + (lane & 31)) * S + r0 + 8 * (lane >> 5);
        return *reinterpret_cast<const bf16x8*>(p);
    } else {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p4 = i & 3, h = lane >> 5;
        const bf16_t* p = lds + (r0 + 8 * h + q) * S + xt + 16 * (g & 1) + 4 * p4;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * S));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return __builtin_bit_cast(bf16x8, r);
    }
}

template <bool RC>
__device__ __forceinline__ float read_frag_f32(const float* lds, int xt, int r, int lane) {
    if (RC) return lds[(xt + (lane & 31)) * Cfg<float>::S_RC + r];
    return lds[r * Cfg<float>::S_XC + xt + (lane & 31)];
}

template <typename T, bool A_RC, bool B_RC, bool ALIGNED>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs a) {
    typedef typename Cfg<T>::GT GT;
    typedef typename Cfg<T>::LT LT;
    constexpr int BK = Cfg<T>::BK;
    constexpr int SA = A_RC ? Cfg<T>::S_RC : Cfg<T>::S_XC;
    constexpr int SB = B_RC ? Cfg<T>::S_RC : Cfg<T>::S_XC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LT* ldsA = reinterpret_cast<LT*>(smem);
    LT* ldsB = ldsA + Cfg<T>::NTILE * tile_elems<T, A_RC>();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int split = blockIdx.z;
    const int rbeg = split * a.r_per_split;
    const int rend = min(a.R, rbeg + a.r_per_split);
    const GT* A = reinterpret_cast<const GT*>(a.A);
    const GT* B = reinterpret_cast<const GT*>(a.B);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    Vec16 ra[4], rb[4];
    const int nt = (rend > rbeg) ? (rend - rbeg + BK - 1) / BK : 0;
    if (nt > 0) {
        load_tile<T, A_RC, ALIGNED>(A, a.lda, m0, a.M, rbeg, rend, ra);
        load_tile<T, B_RC, ALIGNED>(B, a.ldb, n0, a.N, rbeg, rend, rb);
        store_tile<T, A_RC>(ldsA, ra);
        store_tile<T, B_RC>(ldsB, rb);
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) {
            const int r0 = rbeg + (t + 1) * BK;
            load_tile<T, A_RC, ALIGNED>(A, a.lda, m0, a.M, r0, rend, ra);
            load_tile<T, B_RC, ALIGNED>(B, a.ldb, n0, a.N, r0, rend, rb);
        }
        if constexpr (sizeof(LT) == 2 && Cfg<T>::NTILE == 1) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = read_frag_bf16<A_RC, SA>(ldsA, wm * 64 + i * 32, ks * 16, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = read_frag_bf16<B_RC, SB>(ldsB, wn * 64 + j * 32, ks * 16, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if constexpr (std::is_same<T, f16_tag>::value) {
                            typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                __builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fb[j]), acc[i][j], 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                        }
            }
        } else if constexpr (Cfg<T>::NTILE == 2) {
            const bf16_t* aLo = ldsA + tile_elems<T, A_RC>();
            const bf16_t* bLo = ldsB + tile_elems<T, B_RC>();
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fah[i] = read_frag_bf16<A_RC, SA>(ldsA, wm * 64 + i * 32, ks * 16, lane);
                    fal[i] = read_frag_bf16<A_RC, SA>(aLo, wm * 64 + i * 32, ks * 16, lane);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fbh[j] = read_frag_bf16<B_RC, SB>(ldsB, wn * 64 + j * 32, ks * 16, lane);
                    fbl[j] = read_frag_bf16<B_RC, SB>(bLo, wn * 64 + j * 32, ks * 16, lane);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll 4
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int r = 2 * ks + (lane >> 5);
                float fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fa[i] = read_frag_f32<A_RC>(reinterpret_cast<const float*>(ldsA), wm * 64 + i * 32, r, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fb[j] = read_frag_f32<B_RC>(reinterpret_cast<const float*>(ldsB), wn * 64 + j * 32, r, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (t + 1 < nt) {
            store_tile<T, A_RC>(ldsA, ra);
            store_tile<T, B_RC>(ldsB, rb);
        }
        __syncthreads();
    }

    // Epilogue: every load (alpha, the two bias columns, the rank-one factors of the lane's 32 rows) before the first
    // store, and an s_waitcnt the compiler's counter sees: a load inside the store loop -- even behind a branch never taken
    // -- made each of a lane's 64 stores wait for the one before it (stores count in vmcnt on gfx950).
    const float alpha = a.alpha_dev ? *a.alpha_dev : 1.f;
    float* C = a.C + (int64_t)split * a.slab_stride;
    float bvj[2], rs[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        bvj[j] = (a.bias && col < a.N) ? a.bias[col] : 0.f;
    }
    const bool has_rs = a.rowscale != nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            rs[i][e] = (has_rs && split == 0 && row < a.M) ? a.rowscale[row] : 0.f;
        }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // s_waitcnt vmcnt(0)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < a.M && col < a.N) {
                    const float add = has_rs ? bvj[j] * rs[i][e] : bvj[j];
                    if (a.c_f16) reinterpret_cast<f16_t*>(C)[(int64_t)row * a.ldc + col] = (f16_t)(alpha * acc[i][j][e] + add);
                    else C[(int64_t)row * a.ldc + col] = alpha * acc[i][j][e] + add;
                }
            }
        }
}

// ---- tiny problems (the classifier's 16 -> 32 -> 64 -> 30 layers, the latent heads and the decoder's first layer at the
// reference's default batch of 128): one thread per output element, plain fp32 FMAs.  The tiled kernels spend 12-16 us on
// such a launch (a 128x128 tile's prologue and epilogue around one or two K-steps); this one is bound by the launch itself.
// Exact fp32 products, so it also serves the bf16x3 dtype (an approximation of the same thing).
static int64_t tiny_macs() {  // C2_GEMM_TINY_MACS: the threshold, for measurements (default 2^22 multiply-adds: 128 x 128 x 256)
    static const int64_t v = [] {
        const char* e = getenv("C2_GEMM_TINY_MACS");
        return e ? (int64_t)atoll(e) : (int64_t)(1 << 22);
    }();
    return v;
}
#define kTinyMacs tiny_macs()
// Eight lanes share one output element: each walks every eighth reduction index (so 16 384 outputs are 2048 waves, not 256,
// and the serial chain of a 128-long reduction is 16 loads deep, not 128), then three shuffle steps add the eight partial sums.
template <bool A_RC, bool B_RC>
__device__ __forceinline__ void gemm_tiny_body(const GemmArgs& a, int64_t block, int split) {
    const int64_t idx = (block * 256ll + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    const bool live = idx < (int64_t)a.M * a.N;
    const int m = live ? (int)(idx / a.N) : 0, n = live ? (int)(idx - (int64_t)m * a.N) : 0;
    const int rbeg = split * a.r_per_split, rend = min(a.R, rbeg + a.r_per_split);
    const float* A = reinterpret_cast<const float*>(a.A);
    const float* B = reinterpret_cast<const float*>(a.B);
    float acc0 = 0.f, acc1 = 0.f;
    if (live) {
        int r = rbeg + sub;
        for (; r + 8 < rend; r += 16) {  // two independent chains
            const float a0 = A_RC ? A[(int64_t)m * a.lda + r] : A[(int64_t)r * a.lda + m];
            const float b0 = B_RC ? B[(int64_t)n * a.ldb + r] : B[(int64_t)r * a.ldb + n];
            const float a1 = A_RC ? A[(int64_t)m * a.lda + r + 8] : A[(int64_t)(r + 8) * a.lda + m];
            const float b1 = B_RC ? B[(int64_t)n * a.ldb + r + 8] : B[(int64_t)(r + 8) * a.ldb + n];
            acc0 += a0 * b0;
            acc1 += a1 * b1;
        }
        if (r < rend) {
            const float a0 = A_RC ? A[(int64_t)m * a.lda + r] : A[(int64_t)r * a.lda + m];
            const float b0 = B_RC ? B[(int64_t)n * a.ldb + r] : B[(int64_t)r * a.ldb + n];
            acc0 += a0 * b0;
        }
    }
    float acc = acc0 + acc1;
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (!live || sub != 0) return;
    const float alpha = a.alpha_dev ? *a.alpha_dev : 1.f;
    float add = (a.bias ? a.bias[n] : 0.f);
    if (a.rowscale) add = split == 0 ? add * a.rowscale[m] : 0.f;
    float* C = a.C + (int64_t)split * a.slab_stride;
    const float v = alpha * acc + add;
    if (a.c_f16) reinterpret_cast<f16_t*>(C)[(int64_t)m * a.ldc + n] = (f16_t)v;
    else C[(int64_t)m * a.ldc + n] = v;
}
template <bool A_RC, bool B_RC>
__global__ __launch_bounds__(256) void gemm_tiny_kernel(GemmArgs a) {
    gemm_tiny_body<A_RC, B_RC>(a, blockIdx.x, blockIdx.y);
}
// The two GEMMs of a tiny layer's backward pass -- input gradient (dY W) and weight gradient (dY^T A) -- in one launch: the
// first nb0 blocks work on the one, the rest on the other (each a launch-bound kernel of its own before).
__global__ __launch_bounds__(256) void gemm_tiny_pair_kernel(GemmArgs dgrad, GemmArgs wgrad, int nb0) {
    if ((int)blockIdx.x < nb0) gemm_tiny_body<true, false>(dgrad, blockIdx.x, 0);
    else gemm_tiny_body<false, false>(wgrad, blockIdx.x - nb0, 0);
}
static bool tiny_on() {
    static const bool on = [] { const char* e = getenv("C2_GEMM_TINY"); return !(e && e[0] == '0'); }();
    return on;
}
static bool is_tiny(int dtype, const GemmArgs& a) {
    return tiny_on() && (dtype == C2_F32 || dtype == C2_F32X3) && (int64_t)a.M * a.N * a.R <= kTinyMacs;
}
static int launch_tiny(int layout, const GemmArgs& a, int nslab, hipStream_t stream) {
    const dim3 grid((unsigned)(((int64_t)a.M * a.N * 8 + 255) / 256), nslab);  // eight lanes per output element
    if (layout == 0) hipLaunchKernelGGL((gemm_tiny_kernel<true, true>), grid, dim3(256), 0, stream, a);
    else if (layout == 1) hipLaunchKernelGGL((gemm_tiny_kernel<true, false>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gemm_tiny_kernel<false, false>), grid, dim3(256), 0, stream, a);
    return 0;
}

template <typename T, bool A_RC, bool B_RC>
static int launch_gemm(const GemmArgs& a, int nslab, hipStream_t stream) {
    constexpr int VEC = Cfg<T>::VEC;
    const bool aligned = aligned16(a.A) && aligned16(a.B) && (a.lda % VEC == 0) && (a.ldb % VEC == 0);
    const size_t lds = (size_t)Cfg<T>::NTILE * (tile_elems<T, A_RC>() + tile_elems<T, B_RC>()) * sizeof(typename Cfg<T>::LT);
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nslab);
    if (aligned)
        hipLaunchKernelGGL((gemm_kernel<T, A_RC, B_RC, true>), grid, dim3(NT), lds, stream, a);
    else
        hipLaunchKernelGGL((gemm_kernel<T, A_RC, B_RC, false>), grid, dim3(NT), lds, stream, a);
    return 0;
}

template <typename T>
static int gemm_layout(int layout, const GemmArgs& a, int nslab, hipStream_t stream) {
    if (layout == 0) return launch_gemm<T, true, true>(a, nslab, stream);
    if (layout == 1) return launch_gemm<T, true, false>(a, nslab, stream);
    return launch_gemm<T, false, false>(a, nslab, stream);
}

// C2_GEMM_WIDE=0 keeps every GEMM on the 128x128 kernel (A/B runs of the two kernels).
static bool wide_enabled() {
    static const bool on = [] {
        const char* e = getenv("C2_GEMM_WIDE");
        return !(e && e[0] == '0');
    }();
    return on;
}

static int gemm_dispatch(int layout, int dtype, GemmArgs a, int nslab, hipStream_t stream) {
    const int BK = (dtype == C2_BF16 || dtype == C2_BF16X3P || dtype == C2_F16) ? Cfg<bf16_t>::BK : Cfg<float>::BK;  // x3 shares the fp32 K tile
    int64_t per = (a.R + nslab - 1) / nslab;
    per = (per + BK - 1) / BK * BK;
    a.r_per_split = (int)per;
    if (layout <= 2 && is_tiny(dtype, a)) return launch_tiny(layout, a, nslab, stream);
    if (dtype == C2_F16) {  // the half-precision twin of the LDS-DMA kernels, else the register-staged kernel
        if (layout == 3) return 3;
        const int wt16 = wide_enabled() ? wide_tile_f16(dtype, a) : 0;
        if (wt16) return launch_gemm_wide_f16(layout, wt16, a, nslab, stream);
        return gemm_layout<f16_tag>(layout, a, nslab, stream);
    }
    const int wt = (wide_enabled() || dtype == C2_BF16X3P) ? wide_tile(dtype, a) : 0;
    if (layout == 3 && (wt != 256 || dtype != C2_BF16)) return 3;  // (XC, RC) exists on the 256x256 kernel only
    if (wt) return launch_gemm_wide(layout, wt, a, nslab, stream);
    if (dtype == C2_BF16X3P) return 3;  // planes only run on the LDS-DMA kernels
    if (dtype == C2_BF16) return gemm_layout<bf16_t>(layout, a, nslab, stream);
    if (dtype == C2_F32X3) return gemm_layout<x3_t>(layout, a, nslab, stream);
    return gemm_layout<float>(layout, a, nslab, stream);
}

__global__ void sum_slabs_kernel(const float* __restrict__ slabs, int nslab, int64_t stride, float* __restrict__ out,
                                 int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nslab; ++k) s += slabs[k * stride + i];
        out[i] = s;
    }
}

// several split reductions closed by ONE launch (the narrow layers' weight gradients at the end of a backward pass): a table
// of jobs by value, each block owns 1024 consecutive elements of one job
constexpr int kMaxSlabJobs = 16;
struct SlabJobs {
    c2_slab_sum job[kMaxSlabJobs];
    int first_block[kMaxSlabJobs + 1];
    int n;
};
__global__ __launch_bounds__(256) void sum_slabs_multi_kernel(SlabJobs t) {
    int j = 0;
    while (j + 1 < t.n && (int)blockIdx.x >= t.first_block[j + 1]) ++j;
    const c2_slab_sum d = t.job[j];
    const int64_t i0 = ((int64_t)blockIdx.x - t.first_block[j]) * 1024 + threadIdx.x * 4;
    if (i0 >= d.n) return;
    const bool vec = ((d.slab_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.slabs) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(d.out) & 15) == 0) && i0 + 4 <= d.n;
    if (vec) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < d.nslab; ++k) s += *reinterpret_cast<const f32x4*>(d.slabs + k * d.slab_stride + i0);
        *reinterpret_cast<f32x4*>(d.out + i0) = s;
    } else {
        for (int64_t i = i0; i < i0 + 4 && i < d.n; ++i) {
            float s = 0.f;
            for (int k = 0; k < d.nslab; ++k) s += d.slabs[k * d.slab_stride + i];
            d.out[i] = s;
        }
    }
}

__global__ void sum_slabs_2d_kernel(const float* __restrict__ slabs, int nslab, int64_t stride, int64_t rows, int64_t cols,
                                    int64_t lds, float* __restrict__ out, int64_t ldo) {
    const int64_t n = rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i - r * cols;
        float s = 0.f;
        for (int k = 0; k < nslab; ++k) s += slabs[k * stride + r * lds + c];
        out[r * ldo + c] = s;
    }
}

}  // namespace c2

using namespace c2;

static bool gemm_args_ok(int dtype, int64_t M, int64_t N, int64_t K, int nslab) {
    return (dtype == C2_F32 || dtype == C2_BF16 || dtype == C2_F32X3 || dtype == C2_BF16X3P || dtype == C2_F16) && M >= 0 && N >= 0 && K >= 0 && nslab >= 1 && nslab <= 65535 &&
           M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31);
}

extern "C" int c2_gemm_is_tiny(int64_t M, int64_t N, int64_t R) {
    return (tiny_on() && M >= 0 && N >= 0 && R >= 0 && M * N * R <= kTinyMacs) ? 1 : 0;
}

extern "C" int c2_gemm_tile(int dtype, int64_t M, int64_t N, int64_t R) {
    GemmArgs a{};
    a.M = (int)std::min<int64_t>(M, 1 << 30);
    a.N = (int)std::min<int64_t>(N, 1 << 30);
    a.R = (int)std::min<int64_t>(R, 1 << 30);
    if (dtype == C2_BF16X3P) return wide_tile(dtype, a);
    const int wt = !wide_enabled() ? 0 : (dtype == C2_F16 ? wide_tile_f16(dtype, a) : wide_tile(dtype, a));
    return wt ? wt : 128;
}

__global__ void split_bf16_kernel(const float* __restrict__ src, int64_t rows, int64_t cols, int64_t ld,
                                  bf16_t* __restrict__ dst, int64_t ldd, bool vec, int64_t plane) {
    if (vec) {
        const int64_t c4 = cols >> 2, n = rows * c4;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t r = i / c4, c = (i - r * c4) << 2;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * ld + c);
            bf16_t h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = f2bf(v[e]);
                l[e] = f2bf(v[e] - bf2f(h[e]));
            }
            u32x2 ph, pl;
            ph[0] = (unsigned)h[0] | ((unsigned)h[1] << 16);
            ph[1] = (unsigned)h[2] | ((unsigned)h[3] << 16);
            pl[0] = (unsigned)l[0] | ((unsigned)l[1] << 16);
            pl[1] = (unsigned)l[2] | ((unsigned)l[3] << 16);
            *reinterpret_cast<u32x2*>(dst + r * ldd + c) = ph;
            *reinterpret_cast<u32x2*>(dst + plane + r * ldd + c) = pl;
        }
    } else {
        const int64_t n = rows * cols;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t r = i / cols, c = i - r * cols;
            const float v = src[r * ld + c];
            const bf16_t h = f2bf(v);
            dst[r * ldd + c] = h;
            dst[plane + r * ldd + c] = f2bf(v - bf2f(h));
        }
    }
}

extern "C" int c2_split_bf16_planes(const float* src, int64_t rows, int64_t cols, int64_t ld, void* dst_hi, void* dst_lo,
                                    int64_t ldd, void* stream) {
    C2_CHECK(src && dst_hi && dst_lo, "null pointer");
    C2_CHECK(rows >= 0 && cols >= 0 && ld >= cols && ldd >= cols, "bad shape");
    if (rows == 0 || cols == 0) return 0;
    const int64_t plane = (const bf16_t*)dst_lo - (const bf16_t*)dst_hi;  // (elements; either sign)
    const bool vec = (cols % 4 == 0) && (ld % 4 == 0) && (ldd % 4 == 0) && aligned16(src) &&
                     ((reinterpret_cast<uintptr_t>(dst_hi) & 7) == 0) && ((reinterpret_cast<uintptr_t>(dst_lo) & 7) == 0);
    const int64_t n = vec ? rows * (cols / 4) : rows * cols;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(split_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, rows, cols, ld,
                       (bf16_t*)dst_hi, ldd, vec, plane);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_split_bf16(const float* src, int64_t rows, int64_t cols, int64_t ld, void* dst, int64_t ldd,
                             void* stream) {
    C2_CHECK(src && dst, "null pointer");
    return c2_split_bf16_planes(src, rows, cols, ld, dst, (bf16_t*)dst + rows * ldd, ldd, stream);
}

// (A_lo / W_lo: the low planes of C2_BF16X3P operands; NULL = directly behind the high plane, rows x ld elements on)
static int linear_fwd_impl(const char* fn, const void* A, const void* A_lo, int64_t lda, const void* W, const void* W_lo,
                           int64_t ldw, int dtype, int64_t M, int64_t N, int64_t K, float* Y, int64_t ldy, int nslab,
                           int64_t slab_stride, const float* bias, void* stream) {
#define C2_CHECK_FN(cond, msg) do { if (!(cond)) { c2::set_error(std::string(fn) + ": " + (msg)); return 1; } } while (0)
    C2_CHECK_FN(gemm_args_ok(dtype, M, N, K, nslab), "bad dtype/shape");
    C2_CHECK_FN(A && W && Y, "null pointer");
    C2_CHECK_FN(lda >= K && ldw >= K && ldy >= N, "leading dimension too small");
    C2_CHECK_FN(!(bias && nslab != 1), "bias needs nslab == 1");
    C2_CHECK_FN(nslab == 1 || slab_stride >= M * ldy, "slab_stride too small");
    if (M == 0 || N == 0) return 0;
    GemmArgs a{A, W, Y, bias, nullptr, lda, ldw, ldy, slab_stride, (int)M, (int)N, (int)K, 0, nullptr, nullptr, nullptr};
    if (dtype == C2_BF16X3P) {
        a.A_lo = A_lo ? A_lo : (const void*)((const bf16_t*)A + M * lda);
        a.B_lo = W_lo ? W_lo : (const void*)((const bf16_t*)W + N * ldw);
    }
    C2_CHECK_FN(gemm_dispatch(0, dtype, a, nslab, (hipStream_t)stream) == 0, "C2_BF16X3P: shape/alignment not supported");
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_linear_fwd(const void* A, int64_t lda, const void* W, int64_t ldw, int dtype, int64_t M, int64_t N,
                             int64_t K, float* Y, int64_t ldy, int nslab, int64_t slab_stride, const float* bias,
                             void* stream) {
    return linear_fwd_impl(__func__, A, nullptr, lda, W, nullptr, ldw, dtype, M, N, K, Y, ldy, nslab, slab_stride, bias, stream);
}

extern "C" int c2_linear_fwd_planes(const void* A_hi, const void* A_lo, int64_t lda, const void* W_hi, const void* W_lo,
                                    int64_t ldw, int64_t M, int64_t N, int64_t K, float* Y, int64_t ldy, int nslab,
                                    int64_t slab_stride, const float* bias, void* stream) {
    C2_CHECK(A_lo && W_lo, "null pointer");
    return linear_fwd_impl(__func__, A_hi, A_lo, lda, W_hi, W_lo, ldw, C2_BF16X3P, M, N, K, Y, ldy, nslab, slab_stride, bias,
                           stream);
}

extern "C" int c2_linear_fwd_f16out(const void* A, int64_t lda, const void* W, int64_t ldw, int dtype, int64_t M, int64_t N,
                                    int64_t K, void* Y16, int64_t ldy, const float* bias, void* stream) {
    C2_CHECK(gemm_args_ok(dtype, M, N, K, 1), "bad dtype/shape");
    C2_CHECK(A && W && Y16, "null pointer");
    C2_CHECK(lda >= K && ldw >= K && ldy >= N, "leading dimension too small");
    if (M == 0 || N == 0) return 0;
    GemmArgs a{A, W, reinterpret_cast<float*>(Y16), bias, nullptr, lda, ldw, ldy, 0, (int)M, (int)N, (int)K, 0, nullptr, nullptr,
               nullptr, 1};
    if (dtype == C2_BF16X3P) {
        a.A_lo = (const bf16_t*)A + M * lda;
        a.B_lo = (const bf16_t*)W + N * ldw;
    }
    C2_CHECK(gemm_dispatch(0, dtype, a, 1, (hipStream_t)stream) == 0, "C2_BF16X3P: shape/alignment not supported");
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_linear_fwd_rows(const void* A_full, int64_t lda, int64_t a_rows, const int32_t* row_index, const void* W,
                                  int64_t ldw, int dtype, int64_t M, int64_t N, int64_t K, float* Y, int64_t ldy, int nslab,
                                  int64_t slab_stride, const float* bias, void* stream) {
    C2_CHECK(dtype == C2_BF16 || dtype == C2_F16, "row-indexed operands are bf16 or f16");
    C2_CHECK(gemm_args_ok(dtype, M, N, K, nslab), "bad shape");
    C2_CHECK(A_full && row_index && W && Y, "null pointer");
    C2_CHECK(lda >= K && ldw >= K && ldy >= N, "leading dimension too small");
    C2_CHECK(a_rows >= 1 && a_rows * lda < (1ll << 31), "resident matrix too large for 32-bit element offsets");
    C2_CHECK(!(bias && nslab != 1), "bias needs nslab == 1");
    C2_CHECK(nslab == 1 || slab_stride >= M * ldy, "slab_stride too small");
    if (M == 0 || N == 0) return 0;
    GemmArgs a{A_full, W, Y, bias, nullptr, lda, ldw, ldy, slab_stride, (int)M, (int)N, (int)K, 0, nullptr, nullptr, row_index};
    int64_t per = (a.R + nslab - 1) / nslab;
    a.r_per_split = (int)((per + 63) / 64 * 64);
    C2_CHECK((dtype == C2_F16 ? launch_gemm_wide_rows_f16 : launch_gemm_wide_rows)(0, a, nslab, (hipStream_t)stream) == 0,
             "shape/alignment not taken by the 256x256 LDS-DMA kernel (see c2_gemm_tile): gather the rows instead");
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_linear_wgrad_rows(const void* dY, int64_t lddy, const void* A_full, int64_t lda, int64_t a_rows,
                                    const int32_t* row_index, int64_t M, int64_t N, int64_t K, float* dW, int64_t lddw,
                                    int nslab, int64_t slab_stride, const float* alpha_dev, const float* rank1_row, const float* rank1_col, void* stream) {
    C2_CHECK(gemm_args_ok(C2_BF16, M, N, K, nslab), "bad shape");
    C2_CHECK(dY && A_full && row_index && dW, "null pointer");
    C2_CHECK(lddy >= N && lda >= K && lddw >= K, "leading dimension too small");
    C2_CHECK(a_rows >= 1 && a_rows * lda < (1ll << 31), "resident matrix too large for 32-bit element offsets");
    C2_CHECK(nslab == 1 || slab_stride >= N * lddw, "slab_stride too small");
    if (N == 0 || K == 0) return 0;
    // C[n][k] = sum_m dY[m][n] A[rows[m]][k]: A-op = dY (XC over n), B-op = A_full (XC over k, reduction rows indexed)
    GemmArgs a{dY, A_full, dW, nullptr, alpha_dev, lddy, lda, lddw, slab_stride, (int)N, (int)K, (int)M, 0, nullptr, nullptr,
               row_index};
    C2_CHECK(!rank1_row == !rank1_col, "rank1_row and rank1_col come together");
    if (rank1_row) { a.rowscale = rank1_row; a.bias = rank1_col; }
    int64_t per = (a.R + nslab - 1) / nslab;
    a.r_per_split = (int)((per + 63) / 64 * 64);
    C2_CHECK(launch_gemm_wide_rows(2, a, nslab, (hipStream_t)stream) == 0,
             "shape/alignment not taken by the 256x256 LDS-DMA kernel (see c2_gemm_tile): gather the rows instead");
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_linear_wgrad_rows_at(const void* dYt, int64_t lddyt, const void* A_full, int64_t lda, int64_t a_rows,
                                       const int32_t* row_index, int64_t M, int64_t N, int64_t K, float* dW, int64_t lddw,
                                       int nslab, int64_t slab_stride, const float* alpha_dev, const float* rank1_row, const float* rank1_col, void* stream) {
    C2_CHECK(gemm_args_ok(C2_BF16, M, N, K, nslab), "bad shape");
    C2_CHECK(dYt && A_full && row_index && dW, "null pointer");
    C2_CHECK(lddyt >= M && lda >= K && lddw >= K, "leading dimension too small");
    C2_CHECK(a_rows >= 1 && a_rows * lda < (1ll << 31), "resident matrix too large for 32-bit element offsets");
    C2_CHECK(nslab == 1 || slab_stride >= N * lddw, "slab_stride too small");
    if (N == 0 || K == 0) return 0;
    // C[n][k] = sum_m dYt[n][m] A[rows[m]][k]: A-op = dYt (RC over m), B-op = A_full (XC over k, reduction rows indexed)
    GemmArgs a{dYt, A_full, dW, nullptr, alpha_dev, lddyt, lda, lddw, slab_stride, (int)N, (int)K, (int)M, 0, nullptr, nullptr,
               row_index};
    C2_CHECK(!rank1_row == !rank1_col, "rank1_row and rank1_col come together");
    if (rank1_row) { a.rowscale = rank1_row; a.bias = rank1_col; }
    int64_t per = (a.R + nslab - 1) / nslab;
    a.r_per_split = (int)((per + 63) / 64 * 64);
    C2_CHECK(launch_gemm_wide_rows(4, a, nslab, (hipStream_t)stream) == 0,
             "shape/alignment not taken by the 256x256 LDS-DMA kernel (see c2_gemm_tile): use c2_linear_wgrad_rows");
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_linear_wgrad_bt(const void* dY, int64_t lddy, const void* At, int64_t ldat, int64_t M, int64_t N,
                                  int64_t K, float* dW, int64_t lddw, int nslab, int64_t slab_stride,
                                  const float* alpha_dev, const float* rank1_row, const float* rank1_col, void* stream) {
    C2_CHECK(gemm_args_ok(C2_BF16, M, N, K, nslab), "bad shape");
    C2_CHECK(dY && At && dW, "null pointer");
    C2_CHECK(lddy >= N && ldat >= M && lddw >= K, "leading dimension too small");
    C2_CHECK(nslab == 1 || slab_stride >= N * lddw, "slab_stride too small");
    if (N == 0 || K == 0) return 0;
    // C[n][k] = sum_m dY[m][n] At[k][m]: A-op = dY (XC over n), B-op = At (RC over m)
    GemmArgs a{dY, At, dW, nullptr, alpha_dev, lddy, ldat, lddw, slab_stride, (int)N, (int)K, (int)M, 0, nullptr, nullptr, nullptr};
    C2_CHECK(!rank1_row == !rank1_col, "rank1_row and rank1_col come together");
    if (rank1_row) { a.rowscale = rank1_row; a.bias = rank1_col; }
    C2_CHECK(gemm_dispatch(3, C2_BF16, a, nslab, (hipStream_t)stream) == 0,
             "shape/alignment not taken by the 256x256 LDS-DMA kernel (see c2_gemm_tile): use c2_linear_wgrad");
    C2_LAUNCH_CHECK();
    return 0;
}

static int linear_dgrad_impl(const char* fn, const void* dY, const void* dY_lo, int64_t lddy, const void* W, const void* W_lo,
                             int64_t ldw, int dtype, int64_t M, int64_t N, int64_t K, float* dA, int64_t ldda, int nslab,
                             int64_t slab_stride, const float* alpha_dev, void* stream) {
    C2_CHECK_FN(gemm_args_ok(dtype, M, N, K, nslab), "bad dtype/shape");
    C2_CHECK_FN(dY && W && dA, "null pointer");
    C2_CHECK_FN(lddy >= N && ldw >= K && ldda >= K, "leading dimension too small");
    C2_CHECK_FN(nslab == 1 || slab_stride >= M * ldda, "slab_stride too small");
    if (M == 0 || K == 0) return 0;
    // C[m][k] = sum_n dY[m][n] W[n][k]:  A-op = dY (RC over n), B-op = W ([n][k]: XC, output index k contiguous)
    GemmArgs a{dY, W, dA, nullptr, alpha_dev, lddy, ldw, ldda, slab_stride, (int)M, (int)K, (int)N, 0, nullptr, nullptr, nullptr};
    if (dtype == C2_BF16X3P) {
        a.A_lo = dY_lo ? dY_lo : (const void*)((const bf16_t*)dY + M * lddy);
        a.B_lo = W_lo ? W_lo : (const void*)((const bf16_t*)W + N * ldw);
    }
    C2_CHECK_FN(gemm_dispatch(1, dtype, a, nslab, (hipStream_t)stream) == 0, "C2_BF16X3P: shape/alignment not supported");
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_linear_dgrad(const void* dY, int64_t lddy, const void* W, int64_t ldw, int dtype, int64_t M,
                               int64_t N, int64_t K, float* dA, int64_t ldda, int nslab, int64_t slab_stride,
                               const float* alpha_dev, void* stream) {
    return linear_dgrad_impl(__func__, dY, nullptr, lddy, W, nullptr, ldw, dtype, M, N, K, dA, ldda, nslab, slab_stride,
                             alpha_dev, stream);
}

extern "C" int c2_linear_dgrad_planes(const void* dY_hi, const void* dY_lo, int64_t lddy, const void* W_hi, const void* W_lo,
                                      int64_t ldw, int64_t M, int64_t N, int64_t K, float* dA, int64_t ldda, int nslab,
                                      int64_t slab_stride, const float* alpha_dev, void* stream) {
    C2_CHECK(dY_lo && W_lo, "null pointer");
    return linear_dgrad_impl(__func__, dY_hi, dY_lo, lddy, W_hi, W_lo, ldw, C2_BF16X3P, M, N, K, dA, ldda, nslab, slab_stride,
                             alpha_dev, stream);
}

static int linear_wgrad_impl(const char* fn, const void* dY, const void* dY_lo, int64_t lddy, const void* A, const void* A_lo,
                             int64_t lda, int dtype, int64_t M, int64_t N, int64_t K, float* dW, int64_t lddw, int nslab,
                             int64_t slab_stride, const float* alpha_dev, const float* rank1_row, const float* rank1_col,
                             void* stream) {
    C2_CHECK_FN(gemm_args_ok(dtype, M, N, K, nslab), "bad dtype/shape");
    C2_CHECK_FN(dY && A && dW, "null pointer");
    C2_CHECK_FN(lddy >= N && lda >= K && lddw >= K, "leading dimension too small");
    C2_CHECK_FN(nslab == 1 || slab_stride >= N * lddw, "slab_stride too small");
    if (N == 0 || K == 0) return 0;
    // C[n][k] = sum_m dY[m][n] A[m][k]:  A-op = dY ([m][n]: XC over n), B-op = A ([m][k]: XC over k)
    GemmArgs a{dY, A, dW, nullptr, alpha_dev, lddy, lda, lddw, slab_stride, (int)N, (int)K, (int)M, 0, nullptr, nullptr, nullptr};
    C2_CHECK_FN(!rank1_row == !rank1_col, "rank1_row and rank1_col come together");
    if (rank1_row) { a.rowscale = rank1_row; a.bias = rank1_col; }
    if (dtype == C2_BF16X3P) {
        a.A_lo = dY_lo ? dY_lo : (const void*)((const bf16_t*)dY + M * lddy);
        a.B_lo = A_lo ? A_lo : (const void*)((const bf16_t*)A + M * lda);
    }
    C2_CHECK_FN(gemm_dispatch(2, dtype, a, nslab, (hipStream_t)stream) == 0, "C2_BF16X3P: shape/alignment not supported");
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_linear_wgrad(const void* dY, int64_t lddy, const void* A, int64_t lda, int dtype, int64_t M,
                               int64_t N, int64_t K, float* dW, int64_t lddw, int nslab, int64_t slab_stride,
                               const float* alpha_dev, const float* rank1_row, const float* rank1_col, void* stream) {
    return linear_wgrad_impl(__func__, dY, nullptr, lddy, A, nullptr, lda, dtype, M, N, K, dW, lddw, nslab, slab_stride,
                             alpha_dev, rank1_row, rank1_col, stream);
}

extern "C" int c2_linear_wgrad_planes(const void* dY_hi, const void* dY_lo, int64_t lddy, const void* A_hi, const void* A_lo,
                                      int64_t lda, int64_t M, int64_t N, int64_t K, float* dW, int64_t lddw, int nslab,
                                      int64_t slab_stride, const float* alpha_dev, const float* rank1_row,
                                      const float* rank1_col, void* stream) {
    C2_CHECK(dY_lo && A_lo, "null pointer");
    return linear_wgrad_impl(__func__, dY_hi, dY_lo, lddy, A_hi, A_lo, lda, C2_BF16X3P, M, N, K, dW, lddw, nslab, slab_stride,
                             alpha_dev, rank1_row, rank1_col, stream);
}

extern "C" int c2_linear_bwd_tiny(const float* dY, int64_t lddy, const float* W, int64_t ldw, const float* A, int64_t lda,
                                  int64_t M, int64_t N, int64_t K, float* dA, int64_t ldda, float* dW, int64_t lddw,
                                  const float* alpha_dev, void* stream) {
    C2_CHECK(dY && W && A && dA && dW, "null pointer");
    C2_CHECK(M > 0 && N > 0 && K > 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "bad shape");
    C2_CHECK(lddy >= N && ldw >= K && lda >= K && ldda >= K && lddw >= K, "leading dimension too small");
    C2_CHECK(c2_gemm_is_tiny(M, K, N) && c2_gemm_is_tiny(N, K, M), "not a tiny problem (c2_gemm_is_tiny)");
    // dA[m][k] = sum_n dY[m][n] W[n][k]  (A-op dY row-wise, B-op W with the output index contiguous)
    GemmArgs dg{dY, W, dA, nullptr, alpha_dev, lddy, ldw, ldda, 0, (int)M, (int)K, (int)N, 0, nullptr, nullptr, nullptr};
    // dW[n][k] = sum_m dY[m][n] A[m][k]  (both operands with the output index contiguous)
    GemmArgs wg{dY, A, dW, nullptr, alpha_dev, lddy, lda, lddw, 0, (int)N, (int)K, (int)M, 0, nullptr, nullptr, nullptr};
    dg.r_per_split = (int)N;
    wg.r_per_split = (int)M;
    const int nb0 = (int)((M * K * 8 + 255) / 256), nb1 = (int)((N * K * 8 + 255) / 256);
    hipLaunchKernelGGL(gemm_tiny_pair_kernel, dim3(nb0 + nb1), dim3(256), 0, (hipStream_t)stream, dg, wg, nb0);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_sum_slabs_2d(const float* slabs, int nslab, int64_t slab_stride, int64_t rows, int64_t cols, int64_t lds,
                               float* out, int64_t ldo, void* stream) {
    C2_CHECK(slabs && out && nslab >= 1 && rows >= 0 && cols >= 0 && lds >= cols && ldo >= cols, "bad arguments");
    if (rows == 0 || cols == 0) return 0;
    int blocks = (int)std::min<int64_t>((rows * cols + 255) / 256, 2048);
    hipLaunchKernelGGL(sum_slabs_2d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, nslab, slab_stride, rows,
                       cols, lds, out, ldo);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_sum_slabs_multi(const c2_slab_sum* jobs_host, int n_jobs, void* stream) {
    C2_CHECK(n_jobs >= 0 && (n_jobs == 0 || jobs_host), "bad arguments");
    for (int i = 0; i < n_jobs;) {
        SlabJobs t;
        t.n = 0;
        int blocks = 0;
        for (; i < n_jobs && t.n < kMaxSlabJobs; ++i) {
            const c2_slab_sum& d = jobs_host[i];
            if (d.n == 0) continue;
            C2_CHECK(d.slabs && d.out && d.nslab >= 1 && d.n > 0 && d.slab_stride >= d.n, "bad job");
            t.job[t.n] = d;
            t.first_block[t.n] = blocks;
            blocks += (int)((d.n + 1023) / 1024);
            ++t.n;
        }
        t.first_block[t.n] = blocks;
        if (t.n == 0) break;
        hipLaunchKernelGGL(sum_slabs_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
    }
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_sum_slabs(const float* slabs, int nslab, int64_t slab_stride, float* out, int64_t n,
                            void* stream) {
    C2_CHECK(slabs && out && nslab >= 1 && n >= 0, "bad arguments");
    if (n == 0) return 0;
    int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, nslab, slab_stride,
                       out, n);
    C2_LAUNCH_CHECK();
    return 0;
}
