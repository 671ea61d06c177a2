// Multi-dataset MMD (celligner2/model/losses.py:127-249, _utils.py:27-36) for gfx950.
//
// The reference partitions the latent rows by unique condition row and sums, over pairs of partitions,
//   mean K(X,X) + mean K(Y,Y) - 2 mean K(X,Y),   K = sum over 19 bandwidths of exp(-|a-b|^2 / (2 alpha)).
// Here that is ONE weighted double sum over all row pairs,  sum_{a,b} wt[part(a)][part(b)] * K(f_a,f_b):
//   c2_partition      -> part id per row (torch.unique(dim=0) order) + the pair-weight table wt
//   c2_mmd_fwd_bwd    -> 32x32 pair tiles; the Gram block and the gradient contraction sum_j gw_ij f_j both run
//                        on v_mfma_f32_32x32x2_f32 (exact fp32 products), the 19-scale exp sum on the VALU;
//                        every self block is computed once however many pairs use it.
// Rows may be a local shard [row0,row0+Bl) against the all-gathered Bg rows (data-parallel MMD): by symmetry of K
// the local rows' full gradient needs only their own row block, so no backward collective exists.
#include "common.h"
#include <mutex>
#include <stdlib.h>

namespace c2 {

constexpr int MAXP = C2_MMD_MAX_PARTS;

__device__ const float c_mmd_alphas[19] = {1e-6f, 1e-5f, 1e-4f, 1e-3f, 1e-2f, 1e-1f, 1.f,  5.f,  10.f, 15.f,
                                       20.f,  25.f,  30.f,  35.f,  100.f, 1e3f,  1e4f, 1e5f, 1e6f};

// ---------------------------------------------------------------------------------------------------------
// partition: single block
__global__ __launch_bounds__(1024) void partition_kernel(const int64_t* __restrict__ cond, int64_t B, int C,
                                                         int n_conditions, int main_col, int boundary,
                                                         int32_t* __restrict__ part_id, float* __restrict__ wt,
                                                         unsigned long long* __restrict__ keys,
                                                         int32_t* __restrict__ status) {
    __shared__ unsigned long long s_red[16];
    __shared__ unsigned long long s_uniq[MAXP];
    __shared__ int s_map[MAXP];
    __shared__ int s_cnt[MAXP];
    __shared__ int s_n, s_bad, s_nparts;
    __shared__ float s_wt[MAXP * MAXP];
    const int tid = threadIdx.x;
    if (tid == 0) { s_n = 0; s_bad = 0; s_nparts = 0; }
    for (int i = tid; i < MAXP; i += 1024) s_cnt[i] = 0;
    __syncthreads();
    // keys: column 0 is the most significant bit -> numeric order == lexicographic row order
    // (the first KREG keys of a thread also stay in registers: batches up to 32 768 rows never re-read global memory)
    constexpr int KREG = 32;
    unsigned long long kreg[KREG];
    int bad = 0;
#pragma unroll
    for (int i = 0; i < KREG; ++i) kreg[i] = ~0ull;
    auto make_key = [&](int64_t r) {
        unsigned long long k = 0;
        for (int j = 0; j < C; ++j) {
            const int64_t v = cond[r * C + j];
            bad |= (v != 0 && v != 1);
            k = (k << 1) | (unsigned long long)(v == 1);
        }
        return k;
    };
#pragma unroll
    for (int q = 0; q < KREG; ++q) {  // independent iterations: the loads of all of a thread's rows are in flight together
        const int64_t r = tid + 1024ll * q;
        if (r < B) kreg[q] = make_key(r);
    }
    for (int64_t r = tid + 1024ll * KREG; r < B; r += 1024) keys[r] = make_key(r);
    if (bad) s_bad = 1;
    __syncthreads();
    // unique keys ascending by repeated min-extraction
    const unsigned long long NONE = ~0ull;
    unsigned long long cur = 0;
    bool have_cur = false;
    for (int it = 0; it <= MAXP; ++it) {
        unsigned long long m = NONE;
#pragma unroll
        for (int q = 0; q < KREG; ++q) {  // slots past the batch hold NONE, which never wins
            const unsigned long long k = kreg[q];
            if ((!have_cur || k > cur) && k < m) m = k;
        }
        for (int64_t r = tid + 1024ll * KREG; r < B; r += 1024) {
            const unsigned long long k = keys[r];
            if ((!have_cur || k > cur) && k < m) m = k;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(m, o, 64);
            m = other < m ? other : m;
        }
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = m;
        __syncthreads();
        m = s_red[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = s_red[i] < m ? s_red[i] : m;
        if (m == NONE) break;
        if (it == MAXP) {  // more than MAXP distinct rows
            if (tid == 0) s_bad = 1;
            break;
        }
        if (tid == 0) { s_uniq[it] = m; s_n = it + 1; }
        cur = m;
        have_cur = true;
    }
    __syncthreads();
    const int n = s_n;
    if (tid == 0) {
        // partition order: [cat(main rows)] + the rest ascending (_utils.py:31-36)
        int np = 0;
        if (main_col >= 0) {
            bool any_main = false;
            for (int u = 0; u < n; ++u)
                if ((s_uniq[u] >> (C - 1 - main_col)) & 1ull) any_main = true;
            if (!any_main) s_bad = 1;  // torch.cat([]) raises in the reference
            np = 1;
            for (int u = 0; u < n; ++u) {
                if ((s_uniq[u] >> (C - 1 - main_col)) & 1ull) s_map[u] = 0;
                else s_map[u] = np++;
            }
        } else {
            for (int u = 0; u < n; ++u) s_map[u] = np++;
        }
        s_nparts = np;
    }
    __syncthreads();
    auto assign = [&](int64_t r, unsigned long long k) {
        int p = -1;
        if (r < B) {
            int u = 0;
            for (; u < n; ++u)
                if (s_uniq[u] == k) break;
            p = (u < n) ? s_map[u] : 0;
            part_id[r] = p;
        }
        // one LDS atomic per wave and partition (a per-row atomic serialises 4096 adds on a handful of counters);
        // every lane takes part in the ballots (uniform trip counts below)
        for (int q = 0; q < s_nparts; ++q) {
            const unsigned long long bal = __ballot(p == q);
            if ((tid & 63) == 0 && bal) atomicAdd(&s_cnt[q], __popcll(bal));
        }
    };
#pragma unroll
    for (int q = 0; q < KREG; ++q)
        if (1024ll * q < B) assign(tid + 1024ll * q, kreg[q]);
    for (int64_t base = 1024ll * KREG; base < B; base += 1024) {
        const int64_t r = base + tid;
        assign(r, r < B ? keys[r] : 0ull);
    }
    for (int i = tid; i < MAXP * MAXP; i += 1024) s_wt[i] = 0.f;
    __syncthreads();
    if (tid == 0) {
        const int np = s_nparts;
        auto add_pair = [&](int i, int j) {  // accumulated in LDS: a chain of global read-modify-writes costs ~1 us each
            const float ni = (float)s_cnt[i], nj = (float)s_cnt[j];
            s_wt[i * MAXP + i] += 1.f / (ni * ni);
            s_wt[j * MAXP + j] += 1.f / (nj * nj);
            s_wt[i * MAXP + j] -= 1.f / (ni * nj);
            s_wt[j * MAXP + i] -= 1.f / (ni * nj);
        };
        if (boundary >= 0) {  // losses.py:232-238
            for (int i = 0; i < boundary; ++i)
                for (int j = boundary; j < n_conditions; ++j) {
                    if (i >= np || j >= np) { s_bad = 1; continue; }  // IndexError in the reference
                    if (s_cnt[i] < 2 || s_cnt[j] < 2) continue;
                    add_pair(i, j);
                }
        } else {  // losses.py:240-248
            for (int i = 0; i < np; ++i) {
                if (s_cnt[i] < 1) continue;
                for (int j = 0; j < i; ++j) {
                    if (s_cnt[j] < 1) continue;
                    if (main_col >= 0 && j != 0) continue;
                    add_pair(i, j);
                }
            }
        }
        if (s_bad) *status = 2;
    }
    __syncthreads();
    for (int i = tid; i < MAXP * MAXP; i += 1024) wt[i] = s_wt[i];
}

// The same partition for C <= 6 condition columns (at most 64 distinct rows = MAXP, every model of the reference's
// examples): a row IS its key, so the unique rows, their order and the partition sizes fall out of a 64-entry
// presence / count table and two wave ballots -- no repeated min-extraction over the batch, no per-row search.
// Same outputs and status as partition_kernel (tests/test_ops_gpu.py compares both with the oracle).
__global__ __launch_bounds__(1024) void partition_small_kernel(const int64_t* __restrict__ cond, int64_t B, int C,
                                                               int n_conditions, int main_col, int boundary,
                                                               int32_t* __restrict__ part_id, float* __restrict__ wt,
                                                               int32_t* __restrict__ status) {
    __shared__ int s_keycnt[MAXP];   // rows per key
    __shared__ int s_keypart[MAXP];  // key -> partition id
    __shared__ int s_cnt[MAXP];      // rows per partition
    __shared__ int s_bad, s_nparts;
    constexpr int KCACHE = 32768;  // keys of the rows after the first 4096 (one byte each), so that `cond` is walked once
    __shared__ unsigned char s_key[KCACHE];
    const int tid = threadIdx.x;
    if (tid < MAXP) { s_keycnt[tid] = 0; s_cnt[tid] = 0; s_keypart[tid] = 0; }
    if (tid == 0) { s_bad = 0; s_nparts = 0; }
    __syncthreads();
    int bad = 0;
    auto count_key = [&](int k) {  // one LDS atomic per wave and key present in it (leader = lowest lane holding the key)
        unsigned long long todo = __ballot(k >= 0);
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const int kl = __shfl(k, lead, 64);
            const unsigned long long same = __ballot(k == kl);
            if ((tid & 63) == lead) atomicAdd(&s_keycnt[kl], __popcll(same));
            todo &= ~same;
        }
    };
    // The kernel is one block and a chain of latencies (it runs ahead of the MMD in every step): the keys of the first
    // 4096 rows stay in registers -- their loads issued together, rows past the end re-reading the last row, and no second
    // walk over `cond` where part_id is written --, and the weight table is written by 4096 entries in parallel below.
    constexpr int KR = 4;
    auto keys_of = [&](int64_t r0, int (&k)[KR]) {  // rows r0 + q * 1024 + tid, q < KR
        // (six columns whatever C is -- the spare ones re-read the last -- so that the walk over the columns has no trip
        // count the compiler does not know: with one it waits for each load where it is issued)
        int64_t v[KR][6];
#pragma unroll
        for (int q = 0; q < KR; ++q) {
            const int64_t r = min(r0 + (int64_t)q * 1024 + tid, B - 1);
#pragma unroll
            for (int j = 0; j < 6; ++j) v[q][j] = cond[r * C + min(j, C - 1)];
        }
#pragma unroll
        for (int q = 0; q < KR; ++q) {
            int key = 0;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < C) {
                    bad |= (v[q][j] != 0 && v[q][j] != 1);
                    key = (key << 1) | (int)(v[q][j] == 1);
                }
            k[q] = (r0 + (int64_t)q * 1024 + tid < B) ? key : -1;
        }
    };
    int kreg[KR];
    keys_of(0, kreg);
#pragma unroll
    for (int q = 0; q < KR; ++q)
        if ((int64_t)q * 1024 < B) count_key(kreg[q]);
    for (int64_t r0 = (int64_t)KR * 1024; r0 < B; r0 += (int64_t)KR * 1024) {  // (larger batches: the same, a batch at a time)
        int k[KR];
        keys_of(r0, k);
#pragma unroll
        for (int q = 0; q < KR; ++q) {
            const int64_t c = r0 + (int64_t)q * 1024 + tid - (int64_t)KR * 1024;
            if (k[q] >= 0 && c < KCACHE) s_key[c] = (unsigned char)k[q];  // (read back by the thread that wrote it)
            if (r0 + (int64_t)q * 1024 < B) count_key(k[q]);
        }
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (tid < 64) {  // wave 0: lane = key
        const int cnt = s_keycnt[tid];
        const bool present = cnt > 0;
        const unsigned long long below = (1ull << tid) - 1ull;
        int p = 0;
        int np;
        if (main_col >= 0) {  // partition order: [cat(main rows)] + the rest ascending (_utils.py:31-36)
            const bool is_main = present && ((tid >> (C - 1 - main_col)) & 1);
            const unsigned long long bal_main = __ballot(is_main), bal_rest = __ballot(present && !is_main);
            if (!bal_main && tid == 0) s_bad = 1;  // torch.cat([]) raises in the reference
            p = is_main ? 0 : 1 + __popcll(bal_rest & below);
            np = 1 + __popcll(bal_rest);
        } else {
            const unsigned long long bal = __ballot(present);
            p = __popcll(bal & below);
            np = __popcll(bal);
        }
        if (present) {
            s_keypart[tid] = p;
            atomicAdd(&s_cnt[p], cnt);
        }
        if (tid == 0) s_nparts = np;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < KR; ++q)
        if (kreg[q] >= 0) part_id[(int64_t)q * 1024 + tid] = s_keypart[kreg[q]];
    for (int64_t r0 = (int64_t)KR * 1024; r0 < B; r0 += (int64_t)KR * 1024) {
        int k[KR], pid[KR];
        if (r0 <= KCACHE) {  // rows r0 .. r0 + 4095 are entries r0 - 4096 .. r0 - 1 of the key cache (block-uniform)
#pragma unroll
            for (int q = 0; q < KR; ++q) {
                const int64_t r = r0 + (int64_t)q * 1024 + tid;
                k[q] = r < B ? (int)s_key[r - (int64_t)KR * 1024] : -1;
            }
        } else {
            keys_of(r0, k);
        }
#pragma unroll
        for (int q = 0; q < KR; ++q) pid[q] = s_keypart[max(k[q], 0)];
#pragma unroll
        for (int q = 0; q < KR; ++q)
            if (k[q] >= 0) part_id[r0 + (int64_t)q * 1024 + tid] = pid[q];
    }
    // Pair weights.  The reference walks the partition pairs once (losses.py:232-238 with a boundary, :240-248 without)
    // and adds 1/ni^2, 1/nj^2 to the two diagonal entries and -1/(ni nj) to the two off-diagonal ones; here entry (x, y)
    // decides for itself whether that walk visits the pair {x, y}, and a diagonal entry adds its 1/nx^2 once per visited
    // partner, in sequence like the walk does (equal terms: their order does not matter, their count does).
    const int np = s_nparts;
    const bool bnd = boundary >= 0;
    auto visited = [&](int x, int y) {  // x != y
        if (bnd) {
            const int lo = min(x, y), hi = max(x, y);
            return lo < boundary && hi >= boundary && hi < n_conditions && hi < np && s_cnt[lo] >= 2 && s_cnt[hi] >= 2;
        }
        return x < np && y < np && (main_col < 0 || min(x, y) == 0);
    };
#pragma unroll
    for (int q = 0; q < MAXP * MAXP / 1024; ++q) {
        const int e = tid + 1024 * q, x = e / MAXP, y = e % MAXP;
        float w = 0.f;
        if (x < np && y < np) {
            const float nx = (float)s_cnt[x];
            if (x != y) {
                if (visited(x, y)) {
                    // (the walk computes 1/(ni*nj) with i the loop's outer index: the smaller one with a boundary, the
                    // larger one without -- the product commutes, the bits are the same)
                    w = 0.f - 1.f / (nx * (float)s_cnt[y]);
                }
            } else {
                const float t = 1.f / (nx * nx);
                for (int o = 0; o < np; ++o)
                    if (o != x && visited(x, o)) w += t;
            }
        }
        wt[e] = w;
    }
    // with a boundary every (i < boundary, boundary <= j < n_conditions) must name a partition: IndexError in the reference
    if (tid == 0 && (s_bad || (bnd && boundary > 0 && n_conditions > boundary && n_conditions > np))) *status = 2;
}

// ---------------------------------------------------------------------------------------------------------
template <typename TF>
__global__ void row_norms_kernel(const TF* __restrict__ f, int64_t ldf, int64_t Bg, int D, float* __restrict__ norms) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= Bg) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float v = ld_as_f32<TF>(f + row * ldf + d);
        s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) norms[row] = s;
}

constexpr int TS = 32;      // pair tile edge
constexpr int DC = 64;      // feature chunk staged in LDS
constexpr int LDT = DC + 1; // LDS row stride of a feature tile

// one wave per block; block (it, js, dchunk)
// FAST: fp32 features, 16-byte aligned rows, D <= 64 (the latent z): one feature chunk, vector loads only, the next
// column tile prefetched into registers.  The general instance keeps every path (bf16 / ragged / chunked D).
template <typename TF, bool FAST>
__global__ __launch_bounds__(64, FAST ? 3 : 1) void mmd_tile_kernel(const TF* __restrict__ f, int64_t ldf,
                                                      const int32_t* __restrict__ part_id,
                                                      const float* __restrict__ wt, const float* __restrict__ norms,
                                                      int64_t row0, int64_t Bl, int64_t Bg, int D, int jt_per,
                                                      float beta, float* __restrict__ loss_partials,
                                                      float* __restrict__ gslabs /*[jsplit][Bl][D]*/) {
    __shared__ float ldsI[TS * LDT];
    __shared__ float ldsJ[TS * LDT];
    __shared__ float ldsW[TS * (TS + 1)];
    const int lane = threadIdx.x;
    const int it = blockIdx.x, js = blockIdx.y, dchunk = blockIdx.z;
    const int64_t i0 = (int64_t)it * TS;  // local row index of the tile
    const int hi = lane >> 5, lo = lane & 31;
    const int nchunks = FAST ? 1 : (D + DC - 1) / DC;
    const int dgrad0 = FAST ? 0 : dchunk * DC;  // gradient columns owned by this block

    // per-lane row metadata (16 rows) and its norms/parts
    float ni[16];
    int pi[16];
    const int64_t gi0 = row0 + i0 + 4 * hi;  // global row of accumulator element e: gi0 + (e & 3) + 8 * (e >> 2)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * hi;
        const int64_t li = i0 + r;
        const bool ok = li < Bl;
        ni[e] = ok ? norms[row0 + li] : 0.f;
        pi[e] = ok ? part_id[row0 + li] : -1;
    }
    const bool vec_ok = FAST || ((sizeof(TF) == 4) && ((ldf & 3) == 0) && ((D & 3) == 0) &&
                                 ((reinterpret_cast<uintptr_t>(f) & 15) == 0));
    // 32 rows x 64 cols: 16 lanes per row, 4 consecutive features each (one 16-byte load when aligned)
    auto stage_load = [&](f32x4 (&v)[8], int64_t grow0, int64_t nrows_total, int d0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = lane + 64 * k;  // 0..511
            const int r = idx >> 4, c4 = (idx & 15) * 4;
            const int64_t gr = grow0 + r;
            const int d = d0 + c4;
            v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gr < nrows_total && d < D) {
                if (FAST || vec_ok) {
                    v[k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(f) + gr * ldf + d);
                } else if (!FAST) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (d + q < D) v[k][q] = ld_as_f32<TF>(f + gr * ldf + d + q);
                }
            }
        }
    };
    auto stage_store = [&](float* dst, const f32x4 (&v)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = lane + 64 * k;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[r * LDT + c4 + q] = v[k][q];
        }
    };
    auto stage = [&](float* dst, int64_t grow0, int64_t nrows_total, int d0) {
        f32x4 v[8];
        stage_load(v, grow0, nrows_total, d0);
        stage_store(dst, v);
    };

    f32x16 gacc[2];
    float racc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { gacc[0][e] = 0.f; gacc[1][e] = 0.f; racc[e] = 0.f; }
    float lacc = 0.f;

    const int64_t njt = (Bg + TS - 1) / TS;
    const int64_t jt_beg = (int64_t)js * jt_per, jt_end = min(njt, jt_beg + jt_per);
    // single feature chunk (D <= 64): the next column tile is fetched into registers while this one is evaluated
    f32x4 vnext[8];
    if (nchunks == 1) {
        stage(ldsI, row0 + i0, row0 + Bl, 0);
        if (jt_beg < jt_end) stage_load(vnext, jt_beg * TS, Bg, 0);
    }
    for (int64_t jt = jt_beg; jt < jt_end; ++jt) {
        const int64_t j0 = jt * TS;
        // ---- Gram block via fp32 MFMA
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int ch = 0; ch < nchunks; ++ch) {
            __syncthreads();
            if (nchunks > 1) {
                stage(ldsI, row0 + i0, row0 + Bl, ch * DC);
                stage(ldsJ, j0, Bg, ch * DC);
            } else {
                stage_store(ldsJ, vnext);
                if (jt + 1 < jt_end) stage_load(vnext, (jt + 1) * TS, Bg, 0);
            }
            __syncthreads();
#pragma unroll 8
            for (int s = 0; s < DC / 2; ++s) {
                const float a = ldsI[lo * LDT + 2 * s + hi];
                const float b = ldsJ[lo * LDT + 2 * s + hi];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
        // ---- kernel sums for this lane's 16 (i, j) pairs; column j = j0 + lo
        const int64_t gj = j0 + lo;
        const bool jok = gj < Bg;
        const float nj = jok ? norms[gj] : 0.f;
        const int pj = jok ? part_id[gj] : -1;
        float gw[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float w = 0.f;
            if (jok && pi[e] >= 0) w = wt[(unsigned)(pi[e] * MAXP + pj)];
            float d = fmaxf(ni[e] + nj - 2.f * acc[e], 0.f);
            const bool self = (gi0 + ((e & 3) + 8 * (e >> 2))) == gj;
            if (self) d = 0.f;  // exact self distance
            float S = 0.f, Wp = 0.f;
            if (w != 0.f) {
                // beta_k = 1/(2 alpha_k) folded at compile time in fp32, as the reference forms it (losses.py:156);
                // exp(-beta d) = exp2(-beta log2(e) d) on v_exp_f32
#define C2_MMD_TERM(ALPHA)                                                      \
    {                                                                           \
        constexpr float bk = 1.0f / (2.0f * (ALPHA));                           \
        constexpr float ck = -bk * 1.4426950408889634f;                         \
        const float ex = __builtin_amdgcn_exp2f(ck * d);                        \
        S += ex;                                                                \
        Wp = fmaf(2.f * bk, ex, Wp);                                            \
    }
                C2_MMD_TERM(1e-6f) C2_MMD_TERM(1e-5f) C2_MMD_TERM(1e-4f) C2_MMD_TERM(1e-3f) C2_MMD_TERM(1e-2f)
                C2_MMD_TERM(1e-1f) C2_MMD_TERM(1.f) C2_MMD_TERM(5.f) C2_MMD_TERM(10.f) C2_MMD_TERM(15.f)
                C2_MMD_TERM(20.f) C2_MMD_TERM(25.f) C2_MMD_TERM(30.f) C2_MMD_TERM(35.f) C2_MMD_TERM(100.f)
                C2_MMD_TERM(1e3f) C2_MMD_TERM(1e4f) C2_MMD_TERM(1e5f) C2_MMD_TERM(1e6f)
#undef C2_MMD_TERM
            }
            lacc += w * S;
            // a self pair adds K = 19 to the loss but no gradient (f_i - f_i = 0): drop its 1e6-scale weight,
            // which would otherwise only cancel to fp32 rounding in G_i - r_i f_i
            gw[e] = self ? 0.f : w * Wp;
            racc[e] += gw[e];
            // keep the scheduler from interleaving all 16 x 19 exponentials (it did: 437 registers, one wave per SIMD)
            if ((e & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- gradient contraction  G_i += sum_j gw_ij f_j  (columns dgrad0 .. dgrad0+63)
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * hi;
            ldsW[r * (TS + 1) + lo] = gw[e];
        }
        if (nchunks > 1) stage(ldsJ, j0, Bg, dgrad0);
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < TS / 2; ++s) {
            const float a = ldsW[lo * (TS + 1) + 2 * s + hi];
            const float b0 = ldsJ[(2 * s + hi) * LDT + lo];
            const float b1 = ldsJ[(2 * s + hi) * LDT + 32 + lo];
            gacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, gacc[0], 0, 0, 0);
            gacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, gacc[1], 0, 0, 0);
        }
    }
    // ---- finish: r_i over the 32 column lanes, dfeat = 2*beta*(G_i - r_i f_i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) racc[e] += __shfl_xor(racc[e], o, 64);
    }
    __syncthreads();
    if (nchunks > 1) stage(ldsI, row0 + i0, row0 + Bl, dgrad0);
    __syncthreads();
    float* slab = gslabs + (int64_t)js * Bl * D;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * hi;
            const int64_t li = i0 + r;
            const int dcol = dgrad0 + h * 32 + lo;
            if (li < Bl && dcol < D) {
                const float fi = ldsI[r * LDT + h * 32 + lo];
                slab[li * D + dcol] = 2.f * beta * (gacc[h][e] - racc[e] * fi);
            }
        }
    lacc = wave_sum(lacc);
    if (lane == 0 && dchunk == 0) loss_partials[(int64_t)js * gridDim.x + it] = beta * lacc;
}

// ---------------------------------------------------------------------------------------------------------
// Fast instance for the latent features (fp32, D <= 64, 16-byte aligned rows): a block of four waves, each wave
// owning one 32-row tile whose Gram operand stays in REGISTERS for the whole sweep; the four waves share every
// 32-column tile, double-buffered in LDS and prefetched through registers (one block barrier per column tile).
// Two waves per SIMD: one wave's fp32 MFMAs (Gram block, gradient contraction) run under the other's exponentials.
// A kernel scale whose every term exp(-beta_k d) of the tile underflows to exactly 0 (beta_k * min d > 160 / log2 e:
// at the reference's large 1/(2 alpha) scales that is every tile without near-duplicate rows) is skipped -- the
// sums are bit-identical, only the work is not done.
#ifndef C2_MMD_SKIP_T
#define C2_MMD_SKIP_T 160.f  // 2^-160 is below the smallest fp32 denormal: the term is exactly 0
#endif
// C2_MMD_SPLIT_GRAM (default 1): the Gram block on the bf16 matrix pipe with both operands split three ways
// (x = hi + mid + lo, 8 significant bits each; products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid: the dropped ones
// are below 2^-23 of |x||y|, the rounding of an fp32 product): 24 v_mfma_f32_32x32x16_bf16 (32 cycles each) per tile
// instead of 32 v_mfma_f32_32x32x2_f32 (64 cycles each).  0: the fp32-MFMA form.
#ifndef C2_MMD_SPLIT_GRAM
#define C2_MMD_SPLIT_GRAM 1
#endif
#ifndef C2_MMD_POW
#define C2_MMD_POW 1
#endif
__device__ __forceinline__ void split3(float x, bf16_t& h, bf16_t& m, bf16_t& l) {
    h = f2bf(x);
    const float r1 = x - bf2f(h);
    m = f2bf(r1);
    l = f2bf(r1 - bf2f(m));
}
constexpr int FW = 4;  // waves (row tiles) per block
#ifdef C2_MMD_DBG
__device__ unsigned long long g_mmd_clk[8];
#define MMD_T(i) { const unsigned long long t_ = clock64(); ph[i] += t_ - tprev; tprev = t_; }
#else
#define MMD_T(i)
#endif
__global__ __launch_bounds__(64 * FW, 2) void mmd_fast_kernel(const float* __restrict__ f, int64_t ldf,
                                                              const int32_t* __restrict__ part_id,
                                                              const float* __restrict__ wt,
                                                              const float* __restrict__ norms, int64_t row0, int64_t Bl,
                                                              int64_t Bg, int D, int nit, int jt_per, float beta,
                                                              float* __restrict__ loss_partials,
                                                              float* __restrict__ gslabs /*[jsplit][Bl][D]*/) {
    __shared__ float ldsJ[2][TS * LDT];
#if C2_MMD_SPLIT_GRAM
    // the column tile again as three bf16 planes [32 rows][8 chunks of 8 features], chunk c of row r at c ^ (r & 7)
    // (the fragment read -- 16 B per lane, rows down the lanes -- is then bank-conflict free)
    __shared__ __attribute__((aligned(16))) bf16_t ldsP[2][3][TS * 64];
#endif
    __shared__ float ldsW[FW][TS * (TS + 1)];
    __shared__ float ldsNi[FW][TS];
    __shared__ int ldsPi[FW][TS];
    __shared__ float ldsNj[2][TS];   // norms / partitions of the column tile, staged with it
    __shared__ int ldsPj[2][TS];
    __shared__ float ldsWt[MAXP * MAXP];  // the pair-weight table (16 KB): no global load inside the sweep
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {  // (the table's sixteen loads of a thread together, then the sixteen LDS stores: as a load - store loop each trip
       // waited for its own load, eight round trips to L2 at the head of every block)
        float wv[MAXP * MAXP / (64 * FW)];
#pragma unroll
        for (int u = 0; u < MAXP * MAXP / (64 * FW); ++u) wv[u] = wt[tid + 64 * FW * u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < MAXP * MAXP / (64 * FW); ++u) ldsWt[tid + 64 * FW * u] = wv[u];
    }
    const int hi = lane >> 5, lo = lane & 31;
    const int it = blockIdx.x * FW + wave, js = blockIdx.y;
    const int64_t i0 = (int64_t)it * TS;
    const bool active = it < nit;  // wave-uniform; an inactive wave still stages tiles and joins the barriers

#if C2_MMD_SPLIT_GRAM
    // Gram A operand of this wave's rows, split: k-step ks, lane (lo, hi) holds f[i0 + lo][16 ks + 8 hi + 0..7]
    bf16x8 afh[4], afm[4], afl[4];
    {
        const bool ok = active && i0 + lo < Bl;
        const float* fr = f + (row0 + i0 + lo) * ldf;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s16x8 h8, m8, l8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = 16 * ks + 8 * hi + e;
                const float x = (ok && d < D) ? fr[d] : 0.f;
                bf16_t h, m, l;
                split3(x, h, m, l);
                h8[e] = (short)h; m8[e] = (short)m; l8[e] = (short)l;
            }
            afh[ks] = __builtin_bit_cast(bf16x8, h8);
            afm[ks] = __builtin_bit_cast(bf16x8, m8);
            afl[ks] = __builtin_bit_cast(bf16x8, l8);
        }
    }
#else
    // Gram A operand of this wave's rows: step s, lane (lo, hi) holds f[i0 + lo][2 s + hi]
    float afr[DC / 2];
    {
        const bool ok = active && i0 + lo < Bl;
        const float* fr = f + (row0 + i0 + lo) * ldf;
#pragma unroll
        for (int s2 = 0; s2 < DC / 2; ++s2) {
            const int d = 2 * s2 + hi;
            afr[s2] = (ok && d < D) ? fr[d] : 0.f;
        }
    }
#endif
    // row norms / partitions of the wave's 32 rows live in LDS (read back per pair: registers are the scarce resource)
    const int64_t gi0 = row0 + i0 + 4 * hi;
    if (lane < TS) {
        const bool ok = active && i0 + lane < Bl;
        ldsNi[wave][lane] = ok ? norms[row0 + i0 + lane] : 0.f;
        ldsPi[wave][lane] = ok ? part_id[row0 + i0 + lane] : -1;
    }
    const float* myNi = ldsNi[wave];
    const int* myPi = ldsPi[wave];
    f32x16 gacc[2];
    float racc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { gacc[0][e] = 0.f; gacc[1][e] = 0.f; racc[e] = 0.f; }
    float lacc = 0.f;

    // column tile staging: 32 rows x 16 chunks of 4 features, two chunks per thread
    f32x4 vn[2];
    float vnj = 0.f;
    int vpj = -1;
    auto tile_load = [&](int64_t j0) {
        if (tid < TS) {
            const bool ok = j0 + tid < Bg;
            vnj = ok ? norms[j0 + tid] : 0.f;
            vpj = ok ? part_id[j0 + tid] : -1;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid + 256 * k;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
            vn[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j0 + r < Bg && c4 < D) vn[k] = *reinterpret_cast<const f32x4*>(f + (j0 + r) * ldf + c4);
        }
    };
    auto tile_store = [&](float* dst, int buf) {
        if (tid < TS) {
            ldsNj[buf][tid] = vnj;
            ldsPj[buf][tid] = vpj;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid + 256 * k;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[r * LDT + c4 + q] = vn[k][q];
#if C2_MMD_SPLIT_GRAM
            bf16_t h[4], m[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) split3(vn[k][q], h[q], m[q], l[q]);
            const int off = r * 64 + (((c4 >> 3) ^ (r & 7)) << 3) + (c4 & 7);  // elements; 8-byte aligned
            u32x2 ph, pm, pl;
            ph[0] = (unsigned)h[0] | ((unsigned)h[1] << 16); ph[1] = (unsigned)h[2] | ((unsigned)h[3] << 16);
            pm[0] = (unsigned)m[0] | ((unsigned)m[1] << 16); pm[1] = (unsigned)m[2] | ((unsigned)m[3] << 16);
            pl[0] = (unsigned)l[0] | ((unsigned)l[1] << 16); pl[1] = (unsigned)l[2] | ((unsigned)l[3] << 16);
            *reinterpret_cast<u32x2*>(&ldsP[buf][0][off]) = ph;
            *reinterpret_cast<u32x2*>(&ldsP[buf][1][off]) = pm;
            *reinterpret_cast<u32x2*>(&ldsP[buf][2][off]) = pl;
#endif
        }
    };
    const int64_t njt = (Bg + TS - 1) / TS;
    const int64_t jt_beg = (int64_t)js * jt_per, jt_end = min(njt, jt_beg + jt_per);
    if (jt_beg < jt_end) tile_load(jt_beg * TS);
    float* myW = ldsW[wave];
#ifdef C2_MMD_DBG
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = clock64();
    const unsigned long long tstart = tprev;
#endif
    for (int64_t jt = jt_beg; jt < jt_end; ++jt) {
        const int64_t j0 = jt * TS;
        const int buf = (int)((jt - jt_beg) & 1);
        float* tJ = ldsJ[buf];
        tile_store(tJ, buf);
        __syncthreads();  // tile jt visible; every wave is past its reads of the other buffer's previous tile
        MMD_T(0)
        if (jt + 1 < jt_end) tile_load((jt + 1) * TS);
        if (!active) continue;
#if C2_MMD_SPLIT_GRAM
        // ---- Gram block on three-way split bf16 operands (fp32-product accuracy: the distance is a cancellation)
        f32x16 acc, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (16 * ks < D) {
                const int off = lo * 64 + (((2 * ks + hi) ^ (lo & 7)) << 3);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&ldsP[buf][0][off]);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(&ldsP[buf][1][off]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&ldsP[buf][2][off]);
                // small terms first; two chains
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afm[ks], bm, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[ks], bl, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl[ks], bh, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[ks], bm, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afm[ks], bh, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[ks], bh, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += acc1[e];
#else
        // ---- Gram block (fp32 MFMA: the distance is a cancellation)
        // two independent accumulation chains: a dependent MFMA waits out the previous one's 16 passes (and whatever
        // the SIMD's other wave slipped in between), two chains issue back to back
        f32x16 acc, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
        for (int s2 = 0; s2 < DC / 2; s2 += 2) {
            if (2 * s2 < D) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[s2], tJ[lo * LDT + 2 * s2 + hi], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[s2 + 1], tJ[lo * LDT + 2 * s2 + 2 + hi], acc1, 0, 0, 0);
            }
            if ((s2 & 7) == 6) __builtin_amdgcn_sched_barrier(0);  // keep at most 8 operand reads in flight
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += acc1[e];
#endif
        const int64_t gj = j0 + lo;
        const bool jok = gj < Bg;
        const float nj = ldsNj[buf][lo];
        const int pj = ldsPj[buf][lo];
#ifdef C2_MMD_DBG
        asm volatile("s_nop 0" :: "v"(acc[0]), "v"(nj), "v"(pj));
        MMD_T(1)
#endif
        // ---- squared distances of this lane's 16 pairs, and the tile's smallest one
        // (pairs are kept two to a register pair so that the per-term multiply, add and fma are packed fp32 ops)
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        f32x2 dd[8], S[8], Wp[8];
        float dmin = INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * hi;
            float d = fmaxf(myNi[r] + nj - 2.f * acc[e], 0.f);
            if ((gi0 + ((e & 3) + 8 * (e >> 2))) == gj) d = 0.f;  // exact self distance
            dd[e >> 1][e & 1] = d;
            S[e >> 1][e & 1] = 0.f;
            Wp[e >> 1][e & 1] = 0.f;
            if (jok && myPi[r] >= 0) dmin = fminf(dmin, d);
        }
        // wave minimum on DPP lane moves (six __shfl_xor steps are six dependent trips through the LDS crossbar, ~600
        // cycles of a ~18k-cycle tile); lane 63 ends up with the minimum over the wave (EXEC is all ones here)
        {
#define C2_DPP_MIN(ctrl, rows)                                                                                   \
    dmin = fminf(dmin, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, dmin),       \
                                                                             __builtin_bit_cast(int, dmin), ctrl, \
                                                                             rows, 0xF, false)))
            C2_DPP_MIN(0xB1, 0xF);   // quad_perm [1,0,3,2]
            C2_DPP_MIN(0x4E, 0xF);   // quad_perm [2,3,0,1]
            C2_DPP_MIN(0x141, 0xF);  // row_half_mirror
            C2_DPP_MIN(0x140, 0xF);  // row_mirror
            C2_DPP_MIN(0x142, 0xA);  // row_bcast15 into rows 1 and 3
            C2_DPP_MIN(0x143, 0xC);  // row_bcast31 into rows 2 and 3
#undef C2_DPP_MIN
        }
        const float dmin_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dmin), 63));
        MMD_T(2)
        // beta_k = 1/(2 alpha_k) folded at compile time in fp32, as the reference forms it (losses.py:156);
        // exp(-beta d) = exp2(-beta log2(e) d) on v_exp_f32.  The exponential is the quarter-rate instruction this
        // kernel sits on, and the reference's bandwidths are multiples of each other: with C2_MMD_POW (default 1) only
        // alpha = 1e6, 1e4, 100, 35, 30, 20 (and the rarely live alpha <= 0.1) are exponentials, the others are small
        // integer powers of those on packed multiplies -- 1e5 = (1e6)^10, 1e3 = (1e4)^10, 25 = (100)^4, 15 = (30)^2,
        // 10 = (20)^2, 5 = (20)^4, 1 = (20)^20: six exponentials per pair instead of thirteen.  A power n carries n
        // roundings of 2^-24 (n <= 20 from an exponential: 1.2e-6 relative on one of the 19 terms).
#define C2_MMD_ACTIVE(ALPHA) ((1.0f / (2.0f * (ALPHA))) * 1.4426950408889634f * dmin_u < C2_MMD_SKIP_T)
#define C2_MMD_EXP(ALPHA, X)                                                                 \
    {                                                                                        \
        constexpr float ck = -(1.0f / (2.0f * (ALPHA))) * 1.4426950408889634f;               \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                      \
            const f32x2 arg = ck * dd[e];                                                    \
            X[e][0] = __builtin_amdgcn_exp2f(arg[0]);                                        \
            X[e][1] = __builtin_amdgcn_exp2f(arg[1]);                                        \
        }                                                                                    \
    }
#define C2_MMD_ADD(ALPHA, X)                                                                 \
    {                                                                                        \
        constexpr float bk = 1.0f / (2.0f * (ALPHA));                                        \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                      \
            S[e] += X[e];                                                                    \
            Wp[e] = (2.f * bk) * X[e] + Wp[e];                                               \
        }                                                                                    \
    }
#define C2_MMD_TERM(ALPHA)                  \
    if (C2_MMD_ACTIVE(ALPHA)) {             \
        f32x2 x_[8];                        \
        C2_MMD_EXP(ALPHA, x_)               \
        C2_MMD_ADD(ALPHA, x_)               \
    }
#define C2_MMD_SQ(X) _Pragma("unroll") for (int e = 0; e < 8; ++e) X[e] = X[e] * X[e];
#define C2_MMD_P5(X) _Pragma("unroll") for (int e = 0; e < 8; ++e) { const f32x2 q_ = X[e] * X[e]; X[e] = (q_ * q_) * X[e]; }
        C2_MMD_TERM(1e-6f) C2_MMD_TERM(1e-5f) C2_MMD_TERM(1e-4f) C2_MMD_TERM(1e-3f) C2_MMD_TERM(1e-2f)
        C2_MMD_TERM(1e-1f)
#if C2_MMD_POW
        if (C2_MMD_ACTIVE(20.f)) {
            f32x2 x[8];
            C2_MMD_EXP(20.f, x) C2_MMD_ADD(20.f, x)
            if (C2_MMD_ACTIVE(10.f)) {
                C2_MMD_SQ(x) C2_MMD_ADD(10.f, x)
                if (C2_MMD_ACTIVE(5.f)) {
                    C2_MMD_SQ(x) C2_MMD_ADD(5.f, x)
                    if (C2_MMD_ACTIVE(1.f)) { C2_MMD_P5(x) C2_MMD_ADD(1.f, x) }
                }
            }
        }
        if (C2_MMD_ACTIVE(30.f)) {
            f32x2 x[8];
            C2_MMD_EXP(30.f, x) C2_MMD_ADD(30.f, x)
            if (C2_MMD_ACTIVE(15.f)) { C2_MMD_SQ(x) C2_MMD_ADD(15.f, x) }
        }
        C2_MMD_TERM(35.f)
        if (C2_MMD_ACTIVE(100.f)) {
            f32x2 x[8];
            C2_MMD_EXP(100.f, x) C2_MMD_ADD(100.f, x)
            if (C2_MMD_ACTIVE(25.f)) { C2_MMD_SQ(x) C2_MMD_SQ(x) C2_MMD_ADD(25.f, x) }
        }
        if (C2_MMD_ACTIVE(1e4f)) {
            f32x2 x[8];
            C2_MMD_EXP(1e4f, x) C2_MMD_ADD(1e4f, x)
            if (C2_MMD_ACTIVE(1e3f)) { C2_MMD_P5(x) C2_MMD_SQ(x) C2_MMD_ADD(1e3f, x) }
        }
        if (C2_MMD_ACTIVE(1e6f)) {
            f32x2 x[8];
            C2_MMD_EXP(1e6f, x) C2_MMD_ADD(1e6f, x)
            if (C2_MMD_ACTIVE(1e5f)) { C2_MMD_P5(x) C2_MMD_SQ(x) C2_MMD_ADD(1e5f, x) }
        }
#else
        C2_MMD_TERM(1.f) C2_MMD_TERM(5.f) C2_MMD_TERM(10.f) C2_MMD_TERM(15.f)
        C2_MMD_TERM(20.f) C2_MMD_TERM(25.f) C2_MMD_TERM(30.f) C2_MMD_TERM(35.f) C2_MMD_TERM(100.f)
        C2_MMD_TERM(1e3f) C2_MMD_TERM(1e4f) C2_MMD_TERM(1e5f) C2_MMD_TERM(1e6f)
#endif
#undef C2_MMD_TERM
#undef C2_MMD_ACTIVE
#undef C2_MMD_EXP
#undef C2_MMD_ADD
#undef C2_MMD_SQ
#undef C2_MMD_P5
#ifdef C2_MMD_DBG
        asm volatile("s_nop 0" :: "v"(S[0][0]), "v"(Wp[7][1]));
        MMD_T(3)
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's contraction reads of myW (previous tile) are done
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2);
            float w = 0.f;
            const int pie = myPi[r + 4 * hi];
            if (jok && pie >= 0) w = ldsWt[(unsigned)(pie * MAXP + pj)];
            lacc += w * S[e >> 1][e & 1];
            // a self pair adds K = 19 to the loss but no gradient (f_i - f_i = 0): drop its 1e6-scale weight,
            // which would otherwise only cancel to fp32 rounding in G_i - r_i f_i
            const float gw = ((gi0 + r) == gj) ? 0.f : w * Wp[e >> 1][e & 1];
            racc[e] += gw;
            myW[(r + 4 * hi) * (TS + 1) + lo] = gw;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private tile: LDS is in order within a wave
        MMD_T(4)
        // ---- gradient contraction  G_i += sum_j gw_ij f_j
#pragma unroll 4
        for (int s2 = 0; s2 < TS / 2; ++s2) {
            const float a = myW[lo * (TS + 1) + 2 * s2 + hi];
            const float b0 = tJ[(2 * s2 + hi) * LDT + lo];
            const float b1 = tJ[(2 * s2 + hi) * LDT + 32 + lo];
            gacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, gacc[0], 0, 0, 0);
            gacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, gacc[1], 0, 0, 0);
        }
#ifdef C2_MMD_DBG
        asm volatile("s_nop 0" :: "v"(gacc[0][0]), "v"(gacc[1][15]));
        MMD_T(5)
#endif
    }
#ifdef C2_MMD_DBG
    if (blockIdx.x == 3 && blockIdx.y == 5 && tid == 0) {
        for (int i = 0; i < 6; ++i) g_mmd_clk[i] = ph[i];
        g_mmd_clk[6] = clock64() - tstart;
    }
#endif
    if (!active) return;
    // ---- finish: r_i over the 32 column lanes, dfeat = 2*beta*(G_i - r_i f_i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) racc[e] += __shfl_xor(racc[e], o, 64);
    }
    // (every f_i of the wave's 32 x 64 block is requested before the first store: written as load - combine - store per
    // element the 32 elements of a lane were 32 serial round trips to memory, the wait for each load also waiting for the
    // store before it -- on this architecture stores count in vmcnt)
    float* slab = gslabs + (int64_t)js * Bl * D;
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // (two batches of sixteen: the kernel runs at 128 registers)
        const int dcol = h * 32 + lo;
        float fi[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t li = i0 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            const bool ok = li < Bl && dcol < D;
            fi[e] = f[ok ? (row0 + li) * ldf + dcol : (row0 + i0) * ldf];  // (past the edge: any valid element, unused)
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) fi[e] = 2.f * beta * (gacc[h][e] - racc[e] * fi[e]);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // s_waitcnt vmcnt(0), where the compiler's own counter sees it: no wait between the stores
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t li = i0 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (li < Bl && dcol < D) slab[li * D + dcol] = fi[e];
        }
    }
    lacc = wave_sum(lacc);
    if (lane == 0) loss_partials[(int64_t)js * nit + it] = beta * lacc;
}

}  // namespace c2

using namespace c2;

#ifdef C2_MMD_DBG
extern "C" int c2_dbg_mmd_clk(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mmd_clk), 64); }
#endif

extern "C" size_t c2_partition_ws(int64_t B) { return (size_t)(B * 8 + 64); }

extern "C" int c2_partition(const int64_t* cond, int64_t B, int C, int n_conditions, int main_col, int boundary,
                            int32_t* part_id, float* wt, void* ws, int32_t* status_dev, void* stream) {
    C2_CHECK(cond && part_id && wt && ws && status_dev, "null pointer");
    C2_CHECK(C >= 1 && C <= 64, "MMD partition supports 1..64 condition columns");
    C2_CHECK(main_col < C, "main_col out of range");
    C2_CHECK(B >= 1, "empty batch");
    static const bool general_only = getenv("C2_PARTITION_GENERAL") != nullptr;  // A/B and tests: the general kernel
    if (C <= 6 && !general_only)  // at most 2^6 = MAXP distinct rows: the table form
        hipLaunchKernelGGL(partition_small_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cond, B, C, n_conditions,
                           main_col, boundary, part_id, wt, status_dev);
    else
        hipLaunchKernelGGL(partition_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cond, B, C, n_conditions,
                           main_col, boundary, part_id, wt, (unsigned long long*)ws, status_dev);
    C2_LAUNCH_CHECK();
    return 0;
}

static void mmd_split(int64_t Bl, int64_t Bg, int& nit, int& jsplit, int& jt_per) {
    nit = (int)((Bl + TS - 1) / TS);
    const int64_t njt = (Bg + TS - 1) / TS;
    int64_t js = std::max<int64_t>(1, std::min<int64_t>(njt, (4096 + nit - 1) / nit));
    while ((int64_t)nit * js > kReduceWsFloats && js > 1) --js;
    jt_per = (int)((njt + js - 1) / js);
    jsplit = (int)((njt + jt_per - 1) / jt_per);
}

// the four-wave fast kernel: at most 512 blocks = one round at 2 blocks per CU (a second, partial round would cost a
// full sweep), never more slabs than the workspace holds
static void mmd_split_fast(int64_t Bl, int64_t Bg, int nit, int jsplit_max, int& jsplit, int& jt_per) {
    const int nitb = (nit + FW - 1) / FW;
    const int64_t njt = (Bg + TS - 1) / TS;
    int64_t js = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(njt, jsplit_max), std::max(1, 512 / nitb)));
    jt_per = (int)((njt + js - 1) / js);
    jsplit = (int)((njt + jt_per - 1) / jt_per);
}

extern "C" size_t c2_mmd_ws(int64_t Bl, int64_t Bg, int64_t D) {
    int nit, jsplit, jt_per;
    mmd_split(Bl, Bg, nit, jsplit, jt_per);
    return (size_t)(Bg + kReduceWsFloats + (int64_t)jsplit * Bl * D + 64);
}

// Launch sequence shared by c2_mmd_fwd_bwd (caller-owned workspace) and c2_mmd_fwd_bwd_static (the library's own
// file-scope scratch array): `ws` is a DEVICE pointer to >= c2_mmd_ws(Bl,Bg,D) floats, all of it written before read.
static int mmd_run(const void* feat, int f_dtype, int64_t ldf, const int32_t* part_id, const float* wt,
                   int64_t row0, int64_t Bl, int64_t Bg, int64_t D, float beta, float* loss_out,
                   float* dfeat, int64_t lddf, float* ws, void* stream) {
    C2_CHECK(feat && part_id && wt && loss_out && dfeat && ws, "null pointer");
    C2_CHECK(f_dtype == C2_F32 || f_dtype == C2_BF16, "bad dtype");
    C2_CHECK(ldf >= D && lddf >= D && row0 >= 0 && row0 + Bl <= Bg, "bad shape");
    C2_CHECK(lddf == D, "dfeat must be dense [Bl, D]");
    if (Bl == 0 || D == 0) return 0;
    int nit, jsplit, jt_per;
    mmd_split(Bl, Bg, nit, jsplit, jt_per);
    C2_CHECK((int64_t)nit * jsplit <= kReduceWsFloats && nit < 65536, "batch too large for the MMD reduction workspace");
    hipStream_t st = (hipStream_t)stream;
    float* norms = ws;
    float* partials = norms + Bg;
    float* slabs = partials + kReduceWsFloats;
    const int ndch = (int)((D + DC - 1) / DC);
    dim3 grid(nit, jsplit, ndch);
    if (f_dtype == C2_BF16) {
        hipLaunchKernelGGL((row_norms_kernel<bf16_t>), dim3((unsigned)((Bg + 3) / 4)), dim3(256), 0, st, (const bf16_t*)feat,
                           ldf, Bg, (int)D, norms);
        hipLaunchKernelGGL((mmd_tile_kernel<bf16_t, false>), grid, dim3(64), 0, st, (const bf16_t*)feat, ldf, part_id, wt,
                           norms, row0, Bl, Bg, (int)D, jt_per, beta, partials, slabs);
    } else {
        hipLaunchKernelGGL((row_norms_kernel<float>), dim3((unsigned)((Bg + 3) / 4)), dim3(256), 0, st, (const float*)feat,
                           ldf, Bg, (int)D, norms);
        const bool fast = D <= DC && (D & 3) == 0 && (ldf & 3) == 0 && aligned16(feat);
        if (fast) {
            mmd_split_fast(Bl, Bg, nit, jsplit, jsplit, jt_per);
            hipLaunchKernelGGL(mmd_fast_kernel, dim3((nit + FW - 1) / FW, jsplit), dim3(64 * FW), 0, st, (const float*)feat,
                               ldf, part_id, wt, norms, row0, Bl, Bg, (int)D, nit, jt_per, beta, partials, slabs);
        } else
            hipLaunchKernelGGL((mmd_tile_kernel<float, false>), grid, dim3(64), 0, st, (const float*)feat, ldf, part_id, wt,
                               norms, row0, Bl, Bg, (int)D, jt_per, beta, partials, slabs);
    }
    c2_sum_slabs(slabs, jsplit, Bl * D, dfeat, Bl * D, stream);
    launch_finalize_sum(partials, nit * jsplit, loss_out, st);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_mmd_fwd_bwd(const void* feat, int f_dtype, int64_t ldf, const int32_t* part_id, const float* wt,
                              int64_t row0, int64_t Bl, int64_t Bg, int64_t D, float beta, float* loss_out,
                              float* dfeat, int64_t lddf, float* ws, void* stream) {
    return mmd_run(feat, f_dtype, ldf, part_id, wt, row0, Bl, Bg, D, beta, loss_out, dfeat, lddf, ws, stream);
}

// ---- the entry point that allocates nothing --------------------------------------------------------------------
// The MMD's scratch (row norms, loss partials, gradient slabs) in a file-scope device array instead of a caller
// workspace.  `scratch` is a HOST-side stand-in for the array: only the address hipGetSymbolAddress returns for the
// CURRENT device may reach a kernel, and every GPU holds its own copy at its own address.
static __device__ __align__(256) float scratch[C2_MMD_STATIC_FLOATS];

constexpr int kMaxDevices = 64;
static float* g_scratch_dev[kMaxDevices];  // per device: address of that GPU's copy of `scratch` (0 = not asked yet)
static std::mutex g_scratch_mu;

// Address of the current device's scratch array; looked up once per device, with that device current.
static int scratch_on_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    C2_CHECK(e == hipSuccess, std::string("hipGetDevice: ") + hipGetErrorString(e));
    C2_CHECK(dev >= 0 && dev < kMaxDevices, "device ordinal beyond the scratch table");
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_dev[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        C2_CHECK(e == hipSuccess && p, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_dev[dev] = p;
    }
    *out = g_scratch_dev[dev];
    return 0;
}

extern "C" size_t c2_mmd_static_floats(void) { return (size_t)C2_MMD_STATIC_FLOATS; }

extern "C" int c2_mmd_fwd_bwd_static(const void* feat, int f_dtype, int64_t ldf, const int32_t* part_id,
                                     const float* wt, int64_t row0, int64_t Bl, int64_t Bg, int64_t D, float beta,
                                     float* loss_out, float* dfeat, int64_t lddf, void* stream) {
    C2_CHECK(Bl >= 0 && Bg >= 0 && D >= 0, "negative size");
    const size_t need = c2_mmd_ws(Bl, Bg, D);
    C2_CHECK(need <= (size_t)C2_MMD_STATIC_FLOATS,
             "MMD of " + std::to_string(Bl) + " x " + std::to_string(Bg) + " rows, " + std::to_string(D) +
                 " features needs " + std::to_string(need) + " floats of scratch; the static array holds " +
                 std::to_string((size_t)C2_MMD_STATIC_FLOATS) + " (C2_MMD_STATIC_FLOATS): use c2_mmd_fwd_bwd with a "
                 "caller workspace");
    float* ws = nullptr;
    if (int rc = scratch_on_current_device(&ws)) return rc;
    // Nothing of the array has to start at zero: row norms [Bg], loss partials [nit*jsplit] and gradient slabs
    // [jsplit][Bl][D] are each fully written by this call's kernels before its reducing kernels read them.
    // (its loss partials live in the shared array, which the next call on this GPU overwrites: their sum is launched
    // here, never left to a later combine launch -- c2_defer_loss_sums does not apply to this entry point)
    const bool was = set_defer_sums(false);
    const int rc = mmd_run(feat, f_dtype, ldf, part_id, wt, row0, Bl, Bg, D, beta, loss_out, dfeat, lddf, ws, stream);
    set_defer_sums(was);
    return rc;
}
