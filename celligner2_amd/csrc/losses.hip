// Reconstruction (NB / MSE) and classifier losses: forward and analytic backward fused in one pass.
//
// NB kernel layout: one workgroup walks whole rows (cells) in four passes, re-reading the row from L1/L2 so that
// HBM sees each byte once; thread t owns the same gene columns in every row, so the per-gene dispersion gradient
// accumulates in an LDS array across the workgroup's rows and leaves the kernel as one partial row per workgroup
// (summed by a second tiny kernel: deterministic, no atomics).  HBM traffic per cell = logits 4G + x 4G +
// goodloc 1G read, dlogits (2|4)G written: the algorithmic minimum.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace c2 {

// ---- special functions (x > 0) -------------------------------------------------------------------------
// lgamma by an 8-step upward shift for z < 8 and the Stirling series; |err| ~ 1e-7 relative to the terms.
__device__ __forceinline__ float lgamma_pos(float z) {
    const bool small = z < 8.0f;
    float prod = 1.0f;
    float zs = z;
    if (small) {
        prod = z * (z + 1.f) * (z + 2.f) * (z + 3.f);
        prod *= (z + 4.f) * (z + 5.f) * (z + 6.f) * (z + 7.f);
        zs = z + 8.0f;
    }
    const float r = 1.0f / zs;
    const float r2 = r * r;
    // 1/12 - r2/360 + r4/1260 - r6/1680
    const float series = r * (0.083333333333f + r2 * (-0.002777777778f + r2 * (0.000793650794f - r2 * 0.000595238095f)));
    float res = (zs - 0.5f) * logf(zs) - zs + 0.918938533205f + series;
    if (small) res -= logf(prod);
    return res;
}

// digamma by a 6-step shift (as one rational term) and the asymptotic series.
__device__ __forceinline__ float digamma_pos(float z) {
    float corr = 0.f;
    float zs = z;
    if (z < 6.0f) {
        // sum_{k<6} 1/(z+k) = P'(z)/P(z), P = prod (z+k)
        float P = z, dP = 1.f;
#pragma unroll
        for (int k = 1; k < 6; ++k) {
            const float t = z + (float)k;
            dP = dP * t + P;
            P = P * t;
        }
        corr = dP / P;
        zs = z + 6.0f;
    }
    const float r = 1.0f / zs;
    const float r2 = r * r;
    // ln z - 1/(2z) - 1/(12 z^2) + 1/(120 z^4) - 1/(252 z^6) + 1/(240 z^8)
    const float series = r2 * (0.083333333333f - r2 * (0.008333333333f - r2 * (0.003968253968f - r2 * 0.004166666667f)));
    return logf(zs) - 0.5f * r - series - corr;
}

// ---- NB: per-(pattern, gene) constants -----------------------------------------------------------------
// pattern 0: the row holds only 0s  -> q = e0;  1: only 1s -> q = e1;  2: both -> q = e0 + e1
// (celligner2/model/_utils.py:19-24 scatter_ quirk).  tab: [3][3][G] = {theta, K1, K2}
//   K1 = theta*log(theta+eps) - lgamma(theta),  K2 = log(theta+eps) + theta/(theta+eps) - digamma(theta)
// The table's rows are `Gp` apart, G rounded up to a multiple of 12 288 = lcm(4 x 1024, 4 x 768) (nb_tab_ld): the padding holds theta = 1,
// K1 = K2 = 0, values on which the per-gene arithmetic stays finite, so that the fast kernel can read whole chunks
// past the last gene without a bounds test.
__host__ __device__ inline int64_t nb_tab_ld(int64_t G) { return (G + 12287) / 12288 * 12288; }

// `status` (may be null) is set to 2 when a dispersion exp(theta) is below 1e-17 or not a number.
// The same launch also classifies the batch's condition rows for the fast loss kernel (rowflags[b]: 1 = the row holds a 0,
// 2 = it holds a 1, 4 = an entry outside {0,1} or >= C), so that the loss kernel fetches ONE word per row, a row ahead.
__global__ void nb_table_kernel(const float* __restrict__ theta, int C, int64_t G, int64_t Gp, float* __restrict__ tab,
                                const int64_t* __restrict__ cond, int64_t B, int* __restrict__ rowflags,
                                int32_t* __restrict__ status) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (rowflags) {
        for (int64_t r = g; r < B; r += (int64_t)gridDim.x * blockDim.x) {
            int f = 0;
            for (int j = 0; j < C; ++j) {
                const int64_t v = cond[r * C + j];
                f |= (v == 0 ? 1 : 0) | (v == 1 ? 2 : 0) | (((v < 0) | (v > 1) | (v >= C)) ? 4 : 0);
            }
            rowflags[r] = f;
        }
    }
    if (g >= Gp) return;
    if (g >= G) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            tab[(p * 3 + 0) * Gp + g] = 1.f;
            tab[(p * 3 + 1) * Gp + g] = 0.f;
            tab[(p * 3 + 2) * Gp + g] = 0.f;
        }
        return;
    }
    const float t0 = theta[g * C + 0];
    const float t1 = C > 1 ? theta[g * C + 1] : 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float t = p == 0 ? t0 : (p == 1 ? t1 : t0 + t1);
        const float th = expf(t);
        const float lte = logf(th + 1e-8f);
        // the fast loss kernel's domain (nb_pair): a dispersion below 1e-17 is reported, not silently mis-evaluated
        if (status && !(th >= 1e-17f)) *status = 2;
        tab[(p * 3 + 0) * Gp + g] = th;
        tab[(p * 3 + 1) * Gp + g] = th * lte - lgamma_pos(th);
        tab[(p * 3 + 2) * Gp + g] = lte + th / (th + 1e-8f) - digamma_pos(th);
    }
}

struct NbArgs {
    const float* logits;
    const float* x;
    const uint8_t* goodloc;
    const float* sizefactor;
    const int64_t* cond;
    const float* tab;
    void* dlogits;
    float* part_dt;    // [ngroups][3][G]
    int* part_flag;    // [ngroups][3]
    float* part_loss;  // [ngroups]
    int32_t* status;
    const float* lgx;  // optional [B]: sum_g goodloc * lgamma(x+1) of each row (a constant of the data), or NULL
    const int64_t* ridx;  // optional [B]: row b of the batch is row ridx[b] of x / goodloc (resident dataset), or NULL
    int64_t ldl, ldx, ldg, lddl, B, G;
    int C;
    float inv_B;
    const int* rowflags;  // [B] classification of the condition rows (nb_table_kernel); the fast NB kernel only
    float* part_db;       // NULL, or [ngroups][G]: per-workgroup column sums of dlogits (REGT form of the fast kernel only)
};

// fast transcendental forms: every argument here is a normal positive float (>= 1e-8)
// v_log_f32 / v_rcp_f32 / v_exp_f32 directly (1 ulp; no denormal fix-up code, no IEEE division sequence)
__device__ __forceinline__ float flog(float v) { return __builtin_amdgcn_logf(v) * 0.6931471805599453f; }
__device__ __forceinline__ float frcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float fexp(float v) { return __builtin_amdgcn_exp2f(v * 1.4426950408889634f); }

__device__ __forceinline__ float lgamma_fast(float z) {
    const bool small = z < 8.0f;
    float prod = 1.0f, zs = z;
    if (small) {
        prod = z * (z + 1.f) * (z + 2.f) * (z + 3.f);
        prod *= (z + 4.f) * (z + 5.f) * (z + 6.f) * (z + 7.f);
        zs = z + 8.0f;
    }
    const float r = frcp(zs);
    const float r2 = r * r;
    const float series = r * (0.083333333333f + r2 * (-0.002777777778f + r2 * (0.000793650794f - r2 * 0.000595238095f)));
    return (zs - 0.5f) * flog(zs) - zs + 0.918938533205f + series - flog(prod);
}
__device__ __forceinline__ float digamma_fast(float z) {
    float corr = 0.f, zs = z;
    if (z < 6.0f) {
        float P = z, dP = 1.f;
#pragma unroll
        for (int k = 1; k < 6; ++k) {
            const float t = z + (float)k;
            dP = dP * t + P;
            P = P * t;
        }
        corr = dP * frcp(P);
        zs = z + 6.0f;
    }
    const float r = frcp(zs);
    const float r2 = r * r;
    const float series = r2 * (0.083333333333f - r2 * (0.008333333333f - r2 * (0.003968253968f - r2 * 0.004166666667f)));
    return flog(zs) - 0.5f * r - series - corr;
}

// lgamma(z) and digamma(z), z > 0, with ONE shared upward shift (by 4 when z < 4): four transcendentals
// (rcp, log of the shifted argument; log, rcp of the shift product) instead of six, no divergent branch.
// Truncation of the asymptotic series at z >= 4: lgamma < 1/(1188 z^9) = 3.2e-9, digamma < 1/(132 z^10) = 7.2e-9.
__device__ __forceinline__ void lgam_digam_fast(float z, float& lg, float& dg) {
    const bool small = z < 4.0f;
    float P = z, dP = 1.f;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const float t = z + (float)k;
        dP = dP * t + P;
        P = P * t;
    }
    const float zs = small ? z + 4.0f : z;
    P = small ? P : 1.0f;
    dP = small ? dP : 0.0f;
    const float r = frcp(zs), lz = flog(zs);
    const float r2 = r * r;
    lg = (zs - 0.5f) * lz - zs + 0.918938533205f +
         r * (0.083333333333f + r2 * (-0.002777777778f + r2 * (0.000793650794f - r2 * 0.000595238095f))) - flog(P);
    dg = lz - 0.5f * r - r2 * (0.083333333333f - r2 * (0.008333333333f - r2 * (0.003968253968f - r2 * 0.004166666667f))) -
         dP * frcp(P);
}

// One workgroup walks whole rows.  Nothing of a row is kept in registers between its passes: the row (76 KB at
// 19k genes) is re-read from L1/L2, so HBM sees each byte once; the per-gene dispersion-gradient accumulator of the
// workgroup lives in LDS (G floats), owned column-wise by the threads.
//   pass 1  row max of the logits                       (HBM read of logits)
//   pass 2  sum exp(logit - max)                        (L2)
//   pass 3  S = sum_g p_g * gp_g                        (HBM read of x, goodloc; table from L2)
//   pass 4  log-likelihood, dlogits = p*gp - p*S (written once), d(theta) accumulated in LDS
template <typename TD, int NT, bool VEC>
__global__ __launch_bounds__(NT) void nb_fwd_bwd_kernel(NbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dacc[];  // [G rounded up to 4]
    __shared__ float red[32];
    const int tid = threadIdx.x;
    const int64_t G = a.G;
    const int64_t G4 = (G + 3) & ~int64_t(3);
    const float eps = 1e-8f;
    for (int64_t c = tid; c < G4; c += NT) dacc[c] = 0.f;
    int cur_pat = -1, flushed = 0;
    float loss_acc = 0.f;

    auto flush = [&](int pat) {
        __syncthreads();
        float* dst = a.part_dt + ((int64_t)blockIdx.x * 3 + pat) * G;
        const bool first = !((flushed >> pat) & 1);
        for (int64_t c = tid; c < G; c += NT) {
            dst[c] = first ? dacc[c] : dst[c] + dacc[c];
            dacc[c] = 0.f;
        }
        flushed |= 1 << pat;
        __syncthreads();
    };
    // element visitor: f(column, logit, x, good, theta, k1, k2) for the columns of this thread
    auto for_cols = [&](const float* orow, const float* xrow, const uint8_t* grow, const float* tth, const float* tk1,
                        const float* tk2, bool need_x, auto&& f) {
        if (VEC) {
            for (int64_t c = 4ll * tid; c < G; c += 4ll * NT) {
                const f32x4 ov = *reinterpret_cast<const f32x4*>(orow + c);
                f32x4 xv = {0.f, 0.f, 0.f, 0.f}, t0 = xv, t1 = xv, t2 = xv;
                unsigned mb = 0;
                if (need_x) {
                    xv = *reinterpret_cast<const f32x4*>(xrow + c);
                    t0 = *reinterpret_cast<const f32x4*>(tth + c);
                    t1 = *reinterpret_cast<const f32x4*>(tk1 + c);
                    t2 = *reinterpret_cast<const f32x4*>(tk2 + c);
                    mb = *reinterpret_cast<const unsigned*>(grow + c);
                }
                f(c, ov, xv, mb, t0, t1, t2);
            }
        } else {
            for (int64_t c0 = 4ll * tid; c0 < G; c0 += 4ll * NT) {
                f32x4 ov = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                f32x4 xv = {0.f, 0.f, 0.f, 0.f}, t0 = {1.f, 1.f, 1.f, 1.f}, t1 = xv, t2 = xv;
                unsigned mb = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t c = c0 + k;
                    if (c < G) {
                        ov[k] = orow[c];
                        if (need_x) {
                            xv[k] = xrow[c]; t0[k] = tth[c]; t1[k] = tk1[c]; t2[k] = tk2[c];
                            mb |= grow[c] ? (1u << (8 * k)) : 0u;
                        }
                    }
                }
                f(c0, ov, xv, mb, t0, t1, t2);
            }
        }
    };

    for (int64_t row = blockIdx.x; row < a.B; row += gridDim.x) {
        int has0 = 0, has1 = 0, bad = 0;
        for (int j = 0; j < a.C; ++j) {
            const int64_t v = a.cond[row * a.C + j];
            has0 |= (v == 0);
            has1 |= (v == 1);
            bad |= (v < 0) | (v > 1) | (v >= a.C);
        }
        if (bad || (!has0 && !has1)) {
            if (tid == 0) *a.status = 1;
            continue;
        }
        const int pat = (has0 && has1) ? 2 : (has1 ? 1 : 0);
        if (pat != cur_pat) {
            if (cur_pat >= 0) flush(cur_pat);
            cur_pat = pat;
        }
        const float* orow = a.logits + row * a.ldl;
        const int64_t srow = a.ridx ? a.ridx[row] : row;
        const float* xrow = a.x + srow * a.ldx;
        const uint8_t* grow = a.goodloc + srow * a.ldg;
        const float sf = a.sizefactor[row];
        const float* tth = a.tab + (int64_t)(pat * 3 + 0) * nb_tab_ld(G);
        const float* tk1 = a.tab + (int64_t)(pat * 3 + 1) * nb_tab_ld(G);
        const float* tk2 = a.tab + (int64_t)(pat * 3 + 2) * nb_tab_ld(G);
        const float scale = -a.inv_B;
        // pass 1
        float mx = -INFINITY;
        for_cols(orow, xrow, grow, tth, tk1, tk2, false,
                 [&](int64_t, f32x4 ov, f32x4, unsigned, f32x4, f32x4, f32x4) {
                     mx = fmaxf(fmaxf(mx, fmaxf(ov[0], ov[1])), fmaxf(ov[2], ov[3]));
                 });
        mx = block_max<NT>(mx, red);
        // pass 2
        float sm = 0.f;
        for_cols(orow, xrow, grow, tth, tk1, tk2, false,
                 [&](int64_t, f32x4 ov, f32x4, unsigned, f32x4, f32x4, f32x4) {
#pragma unroll
                     for (int k = 0; k < 4; ++k) sm += fexp(ov[k] - mx);
                 });
        sm = block_sum<NT>(sm, red);
        const float inv_sum = 1.0f / sm;
        // pass 3
        float S = 0.f;
        for_cols(orow, xrow, grow, tth, tk1, tk2, true,
                 [&](int64_t, f32x4 ov, f32x4 xv, unsigned mb, f32x4 t0, f32x4, f32x4) {
#pragma unroll
                     for (int k = 0; k < 4; ++k) {
                         const float mu = sf * fexp(ov[k] - mx) * inv_sum;
                         const float zt = xv[k] + t0[k];
                         const float v = xv[k] * mu * frcp(mu + eps) - zt * mu * frcp(t0[k] + mu + eps);
                         S += ((mb >> (8 * k)) & 0xffu) ? scale * v : 0.f;
                     }
                 });
        S = block_sum<NT>(S, red);
        // pass 4
        TD* drow = reinterpret_cast<TD*>(a.dlogits) + row * a.lddl;
        for_cols(orow, xrow, grow, tth, tk1, tk2, true,
                 [&](int64_t c0, f32x4 ov, f32x4 xv, unsigned mb, f32x4 t0, f32x4 t1, f32x4 t2) {
                     float d[4], dt[4];
#pragma unroll
                     for (int k = 0; k < 4; ++k) {
                         const bool m = (mb >> (8 * k)) & 0xffu;
                         const float p = fexp(ov[k] - mx) * inv_sum;
                         const float mu = sf * p;
                         const float th = t0[k], x = xv[k];
                         const float tm = th + mu + eps;
                         const float ltm = flog(tm);
                         const float zt = x + th;
                         // ll = K1 - (theta+x) log(theta+mu+eps) + x log(mu+eps) + lgamma(x+theta) - lgamma(x+1)
                         const float ll = t1[k] - zt * ltm + x * flog(mu + eps) + lgamma_fast(zt) -
                                          (a.lgx ? 0.f : lgamma_fast(x + 1.0f));
                         const float rtm = frcp(tm);
                         const float dmu_mu = x * mu * frcp(mu + eps) - zt * mu * rtm;  // mu * dll/dmu
                         // dll/dtheta = K2 - log(theta+mu+eps) - (theta+x)/(theta+mu+eps) + digamma(x+theta)
                         const float dth = t2[k] - ltm - zt * rtm + digamma_fast(zt);
                         const float pg = m ? scale * dmu_mu : 0.f;
                         d[k] = pg - p * S;
                         dt[k] = m ? scale * th * dth : 0.f;
                         loss_acc += m ? ll : 0.f;
                     }
                     if (VEC) {
                         f32x4 acc = *reinterpret_cast<f32x4*>(dacc + c0);
                         acc[0] += dt[0]; acc[1] += dt[1]; acc[2] += dt[2]; acc[3] += dt[3];
                         *reinterpret_cast<f32x4*>(dacc + c0) = acc;
                         if (sizeof(TD) == 4) {
                             f32x4 o = {d[0], d[1], d[2], d[3]};
                             *reinterpret_cast<f32x4*>(drow + c0) = o;
                         } else {
                             u32x2 pk;
                             pk[0] = (unsigned)f2bf(d[0]) | ((unsigned)f2bf(d[1]) << 16);
                             pk[1] = (unsigned)f2bf(d[2]) | ((unsigned)f2bf(d[3]) << 16);
                             *reinterpret_cast<u32x2*>(drow + c0) = pk;
                         }
                     } else {
#pragma unroll
                         for (int k = 0; k < 4; ++k)
                             if (c0 + k < G) {
                                 dacc[c0 + k] += dt[k];
                                 st_from_f32<TD>(drow + c0 + k, d[k]);
                             }
                     }
                 });
        if (a.lgx && tid == 0) loss_acc -= a.lgx[row];
    }
    if (cur_pat >= 0) flush(cur_pat);
    loss_acc = block_sum<NT>(loss_acc, red);
    if (tid == 0) {
        a.part_loss[blockIdx.x] = -a.inv_B * loss_acc;
#pragma unroll
        for (int p = 0; p < 3; ++p) a.part_flag[blockIdx.x * 3 + p] = (flushed >> p) & 1;
    }
}

// ---- two-wide arithmetic for the fast NB kernel ---------------------------------------------------------------
// The kernel is VALU-bound, so its per-gene arithmetic is written on pairs of genes: gfx950 issues v_pk_add_f32 /
// v_pk_mul_f32 / v_pk_fma_f32 on two fp32 values at the price of one.  Transcendentals stay one per value.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 v2(float s) { return f32x2{s, s}; }
__device__ __forceinline__ f32x2 vlog(f32x2 v) {  // natural log, arguments normal positive floats
    return f32x2{__builtin_amdgcn_logf(v[0]), __builtin_amdgcn_logf(v[1])} * v2(0.6931471805599453f);
}
__device__ __forceinline__ f32x2 vrcp(f32x2 v) { return f32x2{__builtin_amdgcn_rcpf(v[0]), __builtin_amdgcn_rcpf(v[1])}; }

__device__ __forceinline__ f32x2 vlog2(f32x2 v) {  // base-2 log, arguments normal positive floats
    return f32x2{__builtin_amdgcn_logf(v[0]), __builtin_amdgcn_logf(v[1])};
}

// The per-gene arithmetic of the fast NB kernel on a pair of genes, SIX transcendentals per gene (the exponential of
// the softmax, two reciprocals, three logarithms) where the formulas as written take eleven:
//   mu = e * cmu, mue = mu + eps, tm = theta + mue, z = x + theta
//   ll         = K1 - z log tm + x log mue + lgamma(z)               (- lgamma(x+1): a constant of the data)
//   mu dll/dmu = x mu / mue - z mu / tm  =  mu theta (x - mue) / (tm mue)        (exact algebra, no cancellation added)
//   dll/dtheta = K2 - log tm - z / tm + digamma(z)
// lgamma and digamma by ONE upward shift of three, taken unconditionally, and the asymptotic series at zs = z + 3 >= 3:
//   lgamma(z) = lgamma(zs) - log P,  digamma(z) = digamma(zs) - P'/P,  P = z (z+1) (z+2)   (products: no cancellation)
//   truncation after three terms: lgamma < 1/(1680 * 3^7) = 2.7e-7, digamma < 1/(240 * 3^8) = 6.4e-7 (absolute).
// Reciprocals: 1/tm and 1/mue from ONE reciprocal of tm * mue, 1/zs and 1/P from one of zs * P.
// Logarithms: only three combinations are ever needed, each the logarithm of one quotient:
//   A = log(zs / tm)           -> dll/dtheta = K2 + A - z/tm - [digamma series] - P'/P
//   C = log mue
//   Bq = log(zs^5 / P^2)       -> ll = K1 + z A + x C + Bq / 2 - zs + log sqrt(2 pi) + [lgamma series]
//   (z log tm - (zs - 1/2) log zs + log P  =  -z A - 5/2 log zs + log P, since zs - 1/2 - z = 5/2).
// K1 and K2 are NOT added here: they enter the row's loss and the dispersion gradient only through sums over the kept
// genes, sum_g m_g K1_g and theta_g K2_g sum_rows m_g, which the kernel forms from per-gene counts of kept entries when
// it flushes its accumulator -- so the row loop reads ONE table value per gene (theta), not three.
// Domain: zs^5 / P^2 and 1 / (zs P) must be normal floats: 1e-18 <= x + theta <= 1e9 (nb_table_kernel raises the
// status word for a dispersion below that; counts that large are not exact in fp32 anyway).
struct NbPair { f32x2 ll, dmu_mu, dth; };
__device__ __forceinline__ NbPair nb_pair(f32x2 x, f32x2 e, f32x2 th, f32x2 cmu) {
    const f32x2 ln2 = v2(0.6931471805599453f);
    const f32x2 mu = e * cmu;
    const f32x2 mue = mu + v2(1e-8f);
    const f32x2 tm = th + mue;
    const f32x2 z = x + th;
    const f32x2 z1 = z + v2(1.f), z2 = z + v2(2.f), zs = z + v2(3.f);
    const f32x2 m12 = z1 * z2;
    const f32x2 P = z * m12;
    const f32x2 dP = z * (z1 + z2) + m12;
    const f32x2 R1 = vrcp(tm * mue), R2 = vrcp(zs * P);
    const f32x2 rtm = mue * R1;       // 1 / tm
    const f32x2 r = P * R2, rP = zs * R2;  // 1 / zs, 1 / P
    const f32x2 A = vlog2(zs * rtm);
    const f32x2 C = vlog2(mue);
    const f32x2 w = (zs * zs) * rP;
    const f32x2 Bq = vlog2((w * w) * zs);
    const f32x2 r2 = r * r;
    NbPair o;
    const f32x2 lser = r * (v2(0.083333333333f) + r2 * (v2(-0.002777777778f) + r2 * v2(0.000793650794f)));
    o.ll = (z * A + x * C + v2(0.5f) * Bq) * ln2 + ((v2(0.918938533205f) - zs) + lser);
    o.dmu_mu = (mu * R1) * (th * (x - mue));
    const f32x2 dser = r * (v2(0.5f) + r * (v2(0.083333333333f) - r2 * (v2(0.008333333333f) - r2 * v2(0.003968253968f))));
    o.dth = A * ln2 - (z * rtm + dser + dP * rP);
    return o;
}

#ifndef C2_NB_ABL
#define C2_NB_ABL 0  // ablation builds only
#endif
// Block-wide max / sum for the fast NB kernel, three per row.  The wave step runs on DPP lane moves (a few VALU cycles each;
// __shfl_xor goes through the LDS crossbar, a ~100-cycle round trip per step and six steps in a chain), and ONE barrier
// serves a reduction: the waves' partials go to alternating halves of `red`, so the partials of reduction k are
// overwritten by reduction k + 2, which every wave reaches only through the barrier of k + 1.  EXEC must be all ones.
template <bool IS_MAX>
__device__ __forceinline__ float nb_wave_reduce(float v) {
#define C2_DPP(x, ctrl, rows, oldv)                                                                            \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(oldv)), __builtin_bit_cast(int, x), \
                                                          ctrl, rows, 0xF, false))
    auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : a + b; };
    v = op(v, C2_DPP(v, 0xB1, 0xF, v));   // quad_perm [1,0,3,2]
    v = op(v, C2_DPP(v, 0x4E, 0xF, v));   // quad_perm [2,3,0,1]
    v = op(v, C2_DPP(v, 0x141, 0xF, v));  // row_half_mirror: the other quad of each 8
    v = op(v, C2_DPP(v, 0x140, 0xF, v));  // row_mirror: the other 8 of each row of 16
    v = op(v, C2_DPP(v, 0x142, 0xA, IS_MAX ? v : 0.f));  // row_bcast15 into rows 1 and 3
    v = op(v, C2_DPP(v, 0x143, 0xC, IS_MAX ? v : 0.f));  // row_bcast31 into rows 2 and 3: lane 63 holds the wave's value
#undef C2_DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <bool IS_MAX, int NT>
__device__ __forceinline__ float nb_block_reduce(float v, float* red, int& half) {
    constexpr int NW = NT / 64;
    static_assert(NW <= 16, "two halves of 16 partials");
    v = nb_wave_reduce<IS_MAX>(v);
    float* r = red + half;
    half ^= 16;
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
    if (!(C2_NB_ABL & 32)) __syncthreads();
    float s = r[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) s = IS_MAX ? fmaxf(s, r[i]) : s + r[i];
    return s;
}

// Fast path (16-byte aligned rows, G % 4 == 0): one workgroup walks whole rows; thread t owns columns
// 4(t + NT i) .. +3 of chunk i.  Register plan (the row's state is held twice: this row's and the next row's):
//   ev  the row's logits, then exp(logit - max): alive from the softmax statistics to pass B            4 NV
//   nx  the NEXT row's logits, requested at the top of pass A: the statistics open a row, so this is the one fetch
//       nothing else could hide (requested when pass B releases ev, it cost ~3 us of a ~20 us row)          4 NV
//   xq / mq  counts and mask, three rolling slots: chunk i + 2 (or the next row's first chunks) is requested when
//       chunk i is taken up                                                                                   15
//   tq  theta, K1, K2 of the gene pair in hand and of the next one (L2-resident table)                        12
//   p * dloss/dp waits for pass B in a second LDS row (thread-private addresses: no barrier)               LDS
// 8 NV + 27 + ~55 temporaries: 768 threads x 7 chunks at 19k genes (three waves per SIMD, 168 registers).
//   softmax statistics (max, sum of exp) over ev;
//   pass A: everything per gene -- log-likelihood, d(theta) into the LDS accumulator, p * dloss/dp and its row sum S;
//   pass B: dlogits = p * dloss/dp - p * S, stored once.
// Nothing is recomputed between the passes.  Needs accumulator + the second row in the CU's 160 KB (G <= 20 448).
// REGT (512 threads x 10 chunks at 19k genes: two waves per SIMD, 256 registers each): theta of the thread's genes and
// the dispersion-gradient accumulator live in REGISTERS as well -- the row loop then reads no table at all and the only
// LDS traffic left is p * dloss/dp between the passes (one LDS row: G <= 40 000 fits, NV caps it at 20 480 here).
// LH: the logits are IEEE halves (c2_linear_fwd_f16out): 8 bytes per lane and chunk instead of 16, widened when the row's
// turn comes.
template <typename TD, int NT, int NV, bool HAS_LGX, bool REGT = false, bool LH = false>
__global__ __launch_bounds__(NT) void nb_rowreg_kernel(NbArgs a) {
    static_assert(NV >= 2, "the rolling prefetch of counts reaches two chunks ahead");
    extern __shared__ __attribute__((aligned(16))) float dacc[];
    __shared__ float red[32];
    const int tid = threadIdx.x;
    const int64_t G = a.G;
    const int Gi = (int)G;
    const int G4 = (Gi + 3) & ~3;
    float* pgl = dacc + G4;  // [G4] floats: p * dloss/dp of the row in flight
    if (!REGT)
        for (int64_t c = tid; c < G; c += NT) dacc[c] = 0.f;
    // REGT: the first LDS row is free; when the caller wants the column sums of dlogits (the bias gradient of the layer
    // that made the logits) they are accumulated there, every thread on the genes it owns (no barrier)
    const bool want_db = REGT && a.part_db != nullptr;
    if (want_db) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (4 * (tid + NT * i) < G4) *reinterpret_cast<f32x4*>(dacc + 4 * (tid + NT * i)) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 thr[REGT ? NV : 1], dreg[REGT ? NV : 1];  // REGT: theta of the current pattern, d(theta) accumulator
#pragma unroll
    for (int i = 0; i < (REGT ? NV : 1); ++i) dreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur_pat = -1, flushed = 0;
    int red_half = 0;  // which half of `red` the next block reduction writes
    float loss_acc = 0.f;
    // kept[i]: per gene of chunk i (one byte each), the number of rows since the last flush in which the gene was kept by
    // the mask.  K1 and K2 of the log-likelihood depend on the gene and the row's pattern only, so what they add to the
    // loss and to the dispersion gradient over those rows is K1_g * kept_g and scale * theta_g * K2_g * kept_g: added
    // when the accumulator is flushed (the pattern changes, 255 rows have gone by, the walk ends), not per row.
    unsigned kept[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) kept[i] = 0u;
    int rows_kept = 0;
    auto flush = [&](int pat) {
        const int64_t Gp = nb_tab_ld(G);
        float* dst = a.part_dt + ((int64_t)blockIdx.x * 3 + pat) * G;
        const bool first = !((flushed >> pat) & 1);
        // In groups of up to five chunks: every load of a group (three table rows, the partial written so far) is issued
        // before its first use and every store after the last load -- chunk by chunk each load waited for the store of the
        // chunk before it (stores count in vmcnt on this architecture): ten serial round trips to memory per flush.
        // Chunks past the row's end read chunk 0 (a valid address) and store nothing.
        constexpr int GRP = NV < 5 ? NV : 5;
#pragma unroll
        for (int i0 = 0; i0 < NV; i0 += GRP) {
            f32x4 th4[GRP], k14[GRP], k24[GRP], old[GRP];
#pragma unroll
            for (int j = 0; j < GRP; ++j) {
                const int c = 4 * (tid + NT * (i0 + j));
                const int cc = (i0 + j < NV && c < Gi) ? c : 0;
                th4[j] = *reinterpret_cast<const f32x4*>(a.tab + (int64_t)(pat * 3 + 0) * Gp + cc);
                k14[j] = *reinterpret_cast<const f32x4*>(a.tab + (int64_t)(pat * 3 + 1) * Gp + cc);
                k24[j] = *reinterpret_cast<const f32x4*>(a.tab + (int64_t)(pat * 3 + 2) * Gp + cc);
            }
            if (REGT && !first) {
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int c = 4 * (tid + NT * (i0 + j));
                    old[j] = *reinterpret_cast<const f32x4*>(dst + ((i0 + j < NV && c < Gi) ? c : 0));
                }
            }
            f32x4 acc[GRP];
#pragma unroll
            for (int j = 0; j < GRP; ++j) {
                if (i0 + j >= NV) continue;
                const int i = i0 + j;
                const int c = 4 * (tid + NT * i);
                f32x4 n4;
#pragma unroll
                for (int e = 0; e < 4; ++e) n4[e] = (float)((kept[i] >> (8 * e)) & 0xffu);
                const f32x4 add = (-a.inv_B) * (th4[j] * k24[j] * n4);
                if (REGT) {  // every thread owns its genes: no LDS, no barrier
                    acc[j] = dreg[i] + add;
                    if (!first) acc[j] += old[j];
                    dreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else if (c < Gi) {
                    f32x4 t = *reinterpret_cast<f32x4*>(dacc + c);
                    t += add;
                    *reinterpret_cast<f32x4*>(dacc + c) = t;
                }
                if (c < Gi) loss_acc += (k14[j][0] * n4[0] + k14[j][1] * n4[1]) + (k14[j][2] * n4[2] + k14[j][3] * n4[3]);
                kept[i] = 0u;
            }
            if (REGT) {
                __builtin_amdgcn_s_waitcnt(0x0F70);  // s_waitcnt vmcnt(0) where the compiler's counter sees it
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int c = 4 * (tid + NT * (i0 + j));
                    if (i0 + j < NV && c < Gi) *reinterpret_cast<f32x4*>(dst + c) = acc[j];
                }
            }
        }
        rows_kept = 0;
        flushed |= 1 << pat;
        if (REGT) return;
        __syncthreads();
        for (int64_t c = tid; c < G; c += NT) {
            dst[c] = first ? dacc[c] : dst[c] + dacc[c];
            dacc[c] = 0.f;
        }
        __syncthreads();
    };
    // REGT: theta of pattern `pat` into the registers, waited for here (a pattern change is rare), and handed through empty
    // statements so that the row loop's counted vmcnt waits are not sized for this path
    auto load_thr = [&](int pat) {
        if (!REGT) return;
        const float* t = a.tab + (int64_t)(pat * 3) * nb_tab_ld(G) + 4 * tid;
#pragma unroll
        for (int i = 0; i < (REGT ? NV : 1); ++i) thr[i] = *reinterpret_cast<const f32x4*>(t + 4 * NT * i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < (REGT ? NV : 1); ++i) asm volatile("" : "+v"(thr[i]));
    };
    typedef typename std::conditional<LH, u32x2, f32x4>::type nx_t;
    f32x4 ev[NV];
    nx_t nx[NV];
    f32x4 xq[3];       // counts of chunks i, i+1, i+2 (slot = chunk % 3)
    unsigned mq[3];    // their masks
    f32x4 tq[2];       // theta of this chunk's four genes and of the next chunk's (slot = chunk % 2)
    const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    // Rows are addressed through buffer resources (base = the row, extent = its G elements): every lane of every wave
    // issues every memory operation of every chunk -- a lane past the last gene loads zeros and its store is dropped
    // by the bounds check -- so control flow is uniform and ONE per-lane offset register (16 * tid, shifted for the
    // 1- and 2-byte rows) serves all chunks: the chunk offset is an instruction constant.
    constexpr unsigned kRsrcFlags = 0x00020000u;  // raw buffer, 32-bit data format (gfx9)
    auto row_rsrc = [&](const void* base, int bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, kRsrcFlags);
    };
    const int v16 = tid * 16;
    auto load_o = [&](int i, __amdgpu_buffer_rsrc_t r) -> nx_t {
        if constexpr (LH) return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, tid * 8, i * NT * 8, 0));
        else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, v16, i * NT * 16, 0));
    };
    auto logits_rsrc = [&](int64_t r) {  // the logits of batch row r as a buffer of its G elements
        if constexpr (LH) return row_rsrc(reinterpret_cast<const f16_t*>(a.logits) + r * a.ldl, Gi * 2);
        else return row_rsrc(a.logits + r * a.ldl, Gi * 4);
    };
    auto widen = [&](const nx_t& v) -> f32x4 {
        if constexpr (LH) {
            // (bit tests, not a cast of the dwords to half pairs: hipcc 7.2 returned the FIRST dword's halves for both
            // `__builtin_bit_cast(half2, v[0])` and `(..., v[1])` -- tools/dbg/half_load_test.hip)
            f32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[e] = (float)__builtin_bit_cast(f16_t, (unsigned short)((v[e >> 1] >> (16 * (e & 1))) & 0xffffu));
            return w;
        } else {
            return v;
        }
    };
    auto load_x = [&](int i, __amdgpu_buffer_rsrc_t rx) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, v16, i * NT * 16, 0));
    };
    auto load_m = [&](int i, __amdgpu_buffer_rsrc_t rg) {
        return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rg, tid * 4, i * NT * 4, 0);
    };
    // `4 * (tid + NT * i) < always`: true for every lane and chunk, but not to the compiler (it does not know G >= 0): a
    // real branch around each chunk's (and each gene pair's) body keeps its temporaries and loads from being
    // interleaved with the others' (as straight-line code the register allocator spilled 300 registers)
    const int always = Gi + 4 * NT * NV + 8;
    // The FIRST row's logits and its first two chunks of counts / mask are requested here exactly as every later row's
    // are requested by the row before it (into nx and into the rolling slots chunk NV + k would take), and waited for:
    // the loop then has ONE way of reaching a row.  With a second, load-it-now path inside the loop the compiler sized
    // the counted vmcnt waits at the head of a row for that path (where the logits are the newest loads), which on the
    // prefetched path made every row wait out the stores and prefetches issued just before it (~25 % of the kernel).
    if ((int64_t)blockIdx.x < a.B) {
        const int64_t row = blockIdx.x;
        const int64_t srow = a.ridx ? a.ridx[row] : row;
        const auto rx = row_rsrc(a.x + srow * a.ldx, Gi * 4);
        const auto rg = row_rsrc(a.goodloc + srow * a.ldg, Gi);
        const auto rl = logits_rsrc(row);
#pragma unroll
        for (int i = 0; i < NV; ++i) nx[i] = load_o(i, rl);
#pragma unroll
        for (int k = 0; k < 2; ++k) { xq[(NV + k) % 3] = load_x(k, rx); mq[(NV + k) % 3] = load_m(k, rg); }
    }
    // The per-row scalars -- condition pattern, source row, size factor, data-only term -- are fetched one row ahead too:
    // loaded at the head of the row that uses them they are the NEWEST loads, and the wait for the newest load is a full
    // drain of the stores and prefetches issued just before it.
    struct RowMeta { int flags; int srow; float sf, lgx; };  // flags: 1 = a 0 entry, 2 = a 1 entry, 4 = out of {0,1}
    auto load_meta = [&](int64_t r) {
        RowMeta m{0, 0, 0.f, 0.f};
        if (r < a.B) {
            // Read through the constant address space: wave-uniform addresses then become SCALAR loads (written by
            // earlier kernels, read-only here), which land in scalar registers on their own counter -- a vector load
            // of a uniform value is moved to a scalar register at once, i.e. waited for where it is issued.
            typedef const __attribute__((address_space(4))) int* cint_p;
            typedef const __attribute__((address_space(4))) int64_t* clong_p;
            typedef const __attribute__((address_space(4))) float* cfloat_p;
            m.flags = ((cint_p)a.rowflags)[r];
            m.srow = (int)(a.ridx ? ((clong_p)a.ridx)[r] : r);
            m.sf = ((cfloat_p)a.sizefactor)[r];
            m.lgx = HAS_LGX ? ((cfloat_p)a.lgx)[r] : 0.f;
        }
        return m;
    };
    RowMeta nmeta = load_meta(blockIdx.x);
    {
        const int pat0 = ((nmeta.flags & 1) && (nmeta.flags & 2)) ? 2 : ((nmeta.flags & 2) ? 1 : 0);
        if (!REGT) tq[NV % 2] = *reinterpret_cast<const f32x4*>(a.tab + (int64_t)(pat0 * 3) * nb_tab_ld(G) + 4 * tid);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (REGT) {
        // (the compiler does not see that wait: hand the loaded registers through empty statements, so that on this way
        // into the loop they are plain values and the row-head waits are counted on the loop's own order of requests --
        // sized for this path they were full drains that waited out the stores of pass B; at 768 x 7 the same change
        // measured 2 % slower and is not applied there)
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(nx[i]));
#pragma unroll
        for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(xq[k]), "+v"(mq[k]));
    }
    // The workgroups walk rows of equal cost and would stay in step, every CU of an XCD asking for its logits, counts and
    // stores in the same microsecond: every second one starts 2, 4 or 6 us late (same-box A/B: 277 -> 272 us).
#ifndef C2_NB_STAGGER
#define C2_NB_STAGGER 63
#endif
    for (int k = 0; k < (int)((blockIdx.x >> 3) & 3); ++k) __builtin_amdgcn_s_sleep(C2_NB_STAGGER);
#ifdef C2_NB_DBG
    long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = clock64();
#define C2_PH(k) { const long long t_ = clock64(); ph[k] += t_ - tprev; tprev = t_; }
#else
#define C2_PH(k)
#endif
    for (int64_t row = blockIdx.x; row < a.B; row += gridDim.x) {
        const RowMeta meta = nmeta;
        nmeta = load_meta(row + gridDim.x);  // in flight for the whole of this row
        const int has0 = meta.flags & 1, has1 = meta.flags & 2, bad = meta.flags & 4;
        // a row the reference would raise on: flagged, then processed like any other (its numbers are not used -- the
        // host raises when it reads the status word) so that the prefetch chain is not broken
        if ((bad || (!has0 && !has1)) && tid == 0) *a.status = 1;
        const int pat = (has0 && has1) ? 2 : (has1 ? 1 : 0);
        if (pat != cur_pat || rows_kept == 255) {
            if (cur_pat >= 0) flush(cur_pat);
            if (pat != cur_pat) load_thr(pat);
            cur_pat = pat;
        }
        ++rows_kept;
#pragma unroll
        for (int i = 0; i < NV; ++i) ev[i] = widen(nx[i]);
        const bool rotate = true;
        const int64_t srow_c = meta.srow;
        const auto rxc = row_rsrc(a.x + srow_c * a.ldx, Gi * 4);
        const auto rgc = row_rsrc(a.goodloc + srow_c * a.ldg, Gi);
        const int64_t nrow = row + gridDim.x;  // next row of this workgroup (for the prefetch)
        const bool has_next = nrow < a.B;
        // (the last row of a workgroup "prefetches" itself again: every row then issues the same memory operations in
        // the same order, which is what lets the compiler count its vmcnt waits exactly -- with a shorter last-row path
        // merged in, the waits of the final chunks of EVERY row were full drains)
        const int64_t nsrow = has_next ? nmeta.srow : srow_c;
        const auto nrl = logits_rsrc(has_next ? nrow : row);
        const auto nrx = row_rsrc(a.x + nsrow * a.ldx, Gi * 4);
        const auto nrg = row_rsrc(a.goodloc + nsrow * a.ldg, Gi);
        const float sf = meta.sf;
        const float lgx_row = meta.lgx;
        const int64_t Gp = nb_tab_ld(G);
        // theta of the row's pattern, four genes per lane and chunk (K1 and K2 are read when the accumulator is flushed);
        // the last chunk requests the first chunk's theta of the NEXT row's pattern
        const float* tth = a.tab + (int64_t)(pat * 3) * Gp + 4 * tid;
        const int nfl = has_next ? nmeta.flags : meta.flags;
        const int npat = ((nfl & 1) && (nfl & 2)) ? 2 : ((nfl & 2) ? 1 : 0);
        const float* tth_n = a.tab + (int64_t)(npat * 3) * Gp + 4 * tid;
        const float scale = -a.inv_B;
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (4 * (tid + NT * i) >= Gi) ev[i] = ninf;  // columns past the end: exp(-inf - max) = 0
            mx = fmaxf(fmaxf(mx, fmaxf(ev[i][0], ev[i][1])), fmaxf(ev[i][2], ev[i][3]));
        }
        C2_PH(0)
        mx = nb_block_reduce<true, NT>(mx, red, red_half);
        C2_PH(1)
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ev[i][k] = fexp(ev[i][k] - mx);
                sm += ev[i][k];
            }
        sm = nb_block_reduce<false, NT>(sm, red, red_half);
        C2_PH(2)
        const float inv_sum = 1.0f / sm;
        const f32x2 cmu = v2(sf * inv_sum);
        // ---- pass A
        f32x2 S2 = v2(0.f), L2 = v2(0.f);
        // (padded table: always in bounds; 16 bytes per lane and chunk, L2-resident, requested one chunk ahead)
        auto load_th = [&](int i, const float* t) { return *reinterpret_cast<const f32x4*>(t + 4 * NT * i); };
        if (!REGT && NV % 2 == 1) tq[0] = tq[1];  // (chunk NV of the previous row = chunk 0 of this one landed in slot NV % 2)
        if (rotate) {
            // The previous row's last two chunks requested this row's chunks 0 and 1 into the slots they had free (chunk
            // NV + k lands in slot (NV + k) % 3): rotate them to slots 0 and 1.  Done HERE and not at the head of the row:
            // the moves need the values, and those two loads were issued just before pass B of the previous row -- at the
            // head of the row the wave waited out their HBM round trip (phase clocks: ~20 % of the kernel).
            if (NV % 3 == 1) { const f32x4 t = xq[1]; const unsigned u = mq[1]; xq[1] = xq[2]; mq[1] = mq[2]; xq[0] = t; mq[0] = u; }
            if (NV % 3 == 2) { const f32x4 t = xq[2]; const unsigned u = mq[2]; xq[1] = xq[0]; mq[1] = mq[0]; xq[0] = t; mq[0] = u; }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            // this chunk's counts and mask were requested two chunks ago; request those of chunk i + 2 (of the next row
            // when this row ends) into the slot chunk i - 1 has released
            const f32x4 xcur = xq[i % 3];
            const unsigned mcur = mq[i % 3];
            // the next chunk's theta first, the counts of chunk i + 2 after it: vmcnt counts in issue order, so a table
            // load issued after the (HBM) count loads could only be waited for together with them
            if (!REGT && !(C2_NB_ABL & 2)) {
                tq[(i + 1) % 2] = (i + 1 < NV) ? load_th(i + 1, tth) : load_th(0, tth_n);
                __builtin_amdgcn_sched_barrier(0);  // (keep that order)
            }
            if (C2_NB_ABL & 1) {} else
            if (i + 2 < NV) { xq[(i + 2) % 3] = load_x(i + 2, rxc); mq[(i + 2) % 3] = load_m(i + 2, rgc); }
            else { xq[(i + 2) % 3] = load_x(i + 2 - NV, nrx); mq[(i + 2) % 3] = load_m(i + 2 - NV, nrg); }
            // ... and chunk i of the NEXT row's logits: one chunk per chunk, so that the CUs (which walk their rows in step)
            // draw on HBM evenly -- requested all at once at the top of the pass, 256 x 76 KB queued up in front of the
            // table loads of the first chunks
            if (!(C2_NB_ABL & 64)) nx[i] = load_o(i, nrl);
            {  // one more row for the genes the mask keeps (bytes of any non-zero value count as 1)
                unsigned t = mcur | (mcur >> 4);
                t |= t >> 2;
                t |= t >> 1;
                kept[i] += t & 0x01010101u;
            }
            if (4 * (tid + NT * i) < Gi) {  // (waves wholly past the last gene skip the chunk)
                f32x4 dt4, pg4;
#pragma unroll
                for (int h = 0; h < 2; ++h) if (4 * (tid + NT * i) + 2 * h < always) {  // (its own basic block, as above)
                    // keep-mask of the two genes as 0/1 floats (0 past the last gene: the mask load returned zeros); a
                    // count under a cleared mask may hold anything (the reference zeroes ll there after the fact): it is
                    // replaced by 0 so that everything below is finite
                    const float b0 = (float)((mcur >> (16 * h)) & 0xffu), b1 = (float)((mcur >> (16 * h + 8)) & 0xffu);
                    const f32x2 mf = f32x2{b0 != 0.f ? 1.f : 0.f, b1 != 0.f ? 1.f : 0.f};
                    const f32x2 x = f32x2{b0 != 0.f ? xcur[2 * h] : 0.f, b1 != 0.f ? xcur[2 * h + 1] : 0.f};
                    const f32x2 e = f32x2{ev[i][2 * h], ev[i][2 * h + 1]};
                    const f32x2 th = REGT ? f32x2{thr[REGT ? i : 0][2 * h], thr[REGT ? i : 0][2 * h + 1]}
                                          : f32x2{tq[i % 2][2 * h], tq[i % 2][2 * h + 1]};
#if C2_NB_ABL & 4
                    NbPair g; g.ll = x * e + th; g.dmu_mu = x + e * th; g.dth = x - e + th;
#else
                    const NbPair g = nb_pair(x, e, th, cmu);
#endif
                    f32x2 ll = g.ll;
                    if (!HAS_LGX) ll -= f32x2{lgamma_fast(x[0] + 1.0f), lgamma_fast(x[1] + 1.0f)};
                    const f32x2 dmu_mu = g.dmu_mu, dth = g.dth;
                    const f32x2 smf = mf * v2(scale);
                    const f32x2 pgh = smf * dmu_mu;  // p * dloss/dp
                    const f32x2 dth2 = smf * (th * dth);
                    pg4[2 * h] = pgh[0]; pg4[2 * h + 1] = pgh[1];
                    dt4[2 * h] = dth2[0]; dt4[2 * h + 1] = dth2[1];
                    S2 += pgh;
                    L2 += mf * ll;
                }
#if C2_NB_ABL & 8
                S2 += f32x2{dt4[0] + pg4[1], dt4[2] + pg4[3]} + f32x2{dt4[1] + pg4[0], dt4[3] + pg4[2]};
#else
                if (REGT) {
                    dreg[REGT ? i : 0] += dt4;
                } else {
                    f32x4 acc = *reinterpret_cast<f32x4*>(dacc + 4 * (tid + NT * i));
                    acc += dt4;
                    *reinterpret_cast<f32x4*>(dacc + 4 * (tid + NT * i)) = acc;
                }
                *reinterpret_cast<f32x4*>(pgl + 4 * (tid + NT * i)) = pg4;
#endif
            }
        }
        loss_acc += L2[0] + L2[1];
        C2_PH(3)
        const float S = nb_block_reduce<false, NT>(S2[0] + S2[1], red, red_half);
        C2_PH(4)
        // ---- pass B
        const float cS = inv_sum * S;
        const auto rd = row_rsrc(reinterpret_cast<TD*>(a.dlogits) + row * a.lddl, Gi * (int)sizeof(TD));
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 pg4 = {0.f, 0.f, 0.f, 0.f};
            if (4 * (tid + NT * i) < Gi) pg4 = *reinterpret_cast<const f32x4*>(pgl + 4 * (tid + NT * i));
            const f32x4 d = pg4 - ev[i] * cS;
            if (want_db && 4 * (tid + NT * i) < Gi) {
                f32x4 acc = *reinterpret_cast<f32x4*>(dacc + 4 * (tid + NT * i));
                acc += d;
                *reinterpret_cast<f32x4*>(dacc + 4 * (tid + NT * i)) = acc;
            }
            if (sizeof(TD) == 4) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), rd, v16, i * NT * 16, 0);
                // Observed on gfx950 (ROCm 7.2): a VALU write of this store's first data register in the very next issue
                // slot reached memory instead of the value stored (zeros in component 0 of lanes 12-15 of each 16, varying
                // from run to run).  hipcc sees no hazard for a buffer store whose soffset is a scalar register; with one
                // instruction in between the stores were always right.  Two idle issue slots, nothing moved across them:
                asm volatile("s_nop 1" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else {
                u32x2 pk;
                pk[0] = (unsigned)f2bf(d[0]) | ((unsigned)f2bf(d[1]) << 16);
                pk[1] = (unsigned)f2bf(d[2]) | ((unsigned)f2bf(d[3]) << 16);
                if (!(C2_NB_ABL & 128)) __builtin_amdgcn_raw_buffer_store_b64(pk, rd, tid * 8, i * NT * 8, 0);
                else asm volatile("" :: "v"(pk));
            }
        }
        if (HAS_LGX && tid == 0) loss_acc -= lgx_row;
        C2_PH(5)
    }
#ifdef C2_NB_DBG
    if ((tid & 63) == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && (tid == 0 || tid == 704))
        printf("nb phases blk %d wave %d: head+max %lld | max-barrier %lld | exp+sum-barrier %lld | passA %lld | S-barrier %lld | passB %lld\n",
               (int)blockIdx.x, tid >> 6, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5]);
#endif
#undef C2_PH
    if (want_db) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * (tid + NT * i);
            if (c < Gi) *reinterpret_cast<f32x4*>(a.part_db + (int64_t)blockIdx.x * G + c) = *reinterpret_cast<const f32x4*>(dacc + c);
        }
    }
    if (cur_pat >= 0) flush(cur_pat);
    loss_acc = block_sum<NT>(loss_acc, red);
    if (tid == 0) {
        a.part_loss[blockIdx.x] = -a.inv_B * loss_acc;
#pragma unroll
        for (int p = 0; p < 3; ++p) a.part_flag[blockIdx.x * 3 + p] = (flushed >> p) & 1;
    }
}

// dtheta reduction, stage 1: partial sums over slices of the workgroup partials
constexpr int kDthSlices = 32;
__global__ __launch_bounds__(256) void nb_dtheta_stage1(const float* __restrict__ part_dt, const int* __restrict__ part_flag,
                                                        int ngroups, int64_t G, float* __restrict__ part2 /*[S][4][G]*/,
                                                        const float* __restrict__ part_db /*NULL or [ngroups][G]*/) {
    const int64_t g = blockIdx.x * 256ll + threadIdx.x;
    const int sl = blockIdx.y;
    const int per = (ngroups + kDthSlices - 1) / kDthSlices;
    const int g0 = sl * per, g1 = min(ngroups, g0 + per);
    if (g >= G) return;
    // (the load stays behind the flag's branch: most workgroups see one or two of the three patterns, and loading the
    // unflagged rows as well -- four in flight, dropped by a select -- was measured slower, 16 -> 20 us: more bytes than
    // the serial round trips cost)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float s = 0.f;
        for (int grp = g0; grp < g1; ++grp)
            if (part_flag[grp * 3 + p]) s += part_dt[((int64_t)grp * 3 + p) * G + g];
        part2[((int64_t)sl * 4 + p) * G + g] = s;
    }
    if (part_db) {  // fourth plane: the workgroups' column sums of dlogits
        float s = 0.f;
        for (int grp = g0; grp < g1; ++grp) s += part_db[(int64_t)grp * G + g];
        part2[((int64_t)sl * 4 + 3) * G + g] = s;
    }
}
// The same on 16-byte accesses (G a multiple of 4, aligned partials; at most 8 workgroups per slice: ngroups <= 256).  Every
// load of a thread -- up to eight workgroup partials of each of the four planes -- is issued before the first is used.  A
// partial whose flag is down is not left out by a branch (the load behind it would be waited for inside it: one round
// trip to memory per partial, which is what held the scalar form at 3 TB/s) and not loaded either (more bytes than it
// saves, see above): its load re-reads a flagged partial of the same slice, which is in flight or in the cache, and a
// select drops the value.  Same order of additions as the scalar form.
__global__ __launch_bounds__(256) void nb_dtheta_stage1_v4(const float* __restrict__ part_dt, const int* __restrict__ part_flag,
                                                           int ngroups, int64_t G, float* __restrict__ part2,
                                                           const float* __restrict__ part_db) {
    constexpr int PER = 8;
    const int64_t g = (blockIdx.x * 256ll + threadIdx.x) * 4;
    const int sl = blockIdx.y;
    const int per = (ngroups + kDthSlices - 1) / kDthSlices;  // <= PER
    const int g0 = sl * per, g1 = min(ngroups, g0 + per);
    if (g >= G) return;
    bool on[4][PER];
    f32x4 v[4][PER];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (p == 3 && !part_db) break;
        int any = min(g0, ngroups - 1);  // (no flagged partial in the slice: one partial's chunk is read eight times and dropped)
#pragma unroll
        for (int u = PER - 1; u >= 0; --u) {
            const int grp = g0 + u;
            on[p][u] = grp < g1 && (p == 3 || part_flag[min(grp, ngroups - 1) * 3 + p] != 0);
            if (on[p][u]) any = grp;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int grp = on[p][u] ? g0 + u : any;
            const float* src = p == 3 ? part_db + (int64_t)grp * G : part_dt + ((int64_t)grp * 3 + p) * G;
            v[p][u] = *reinterpret_cast<const f32x4*>(src + g);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 s[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        s[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p == 3 && !part_db) break;
#pragma unroll
        for (int u = 0; u < PER; ++u) s[p] = on[p][u] ? s[p] + v[p][u] : s[p];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (p == 3 && !part_db) break;
        *reinterpret_cast<f32x4*>(part2 + ((int64_t)sl * 4 + p) * G + g) = s[p];
    }
}
// column sums of dlogits where the loss kernel did not form them: slice sl of the rows into the fourth plane of part2
template <typename TD>
__global__ __launch_bounds__(256) void nb_colsum_kernel(const TD* __restrict__ d, int64_t ld, int64_t B, int64_t G,
                                                        float* __restrict__ part2) {
    const int64_t g = blockIdx.x * 256ll + threadIdx.x;
    const int sl = blockIdx.y;
    const int64_t per = (B + kDthSlices - 1) / kDthSlices;
    const int64_t b0 = sl * per, b1 = min(B, b0 + per);
    if (g >= G) return;
    float s = 0.f;
    for (int64_t b = b0; b < b1; ++b) s += ld_as_f32(d + b * ld + g);
    part2[((int64_t)sl * 4 + 3) * G + g] = s;
}
// Few workgroup partials (a batch of at most kDthDirect rows: the reference's default 128): ONE launch instead of stage 1 +
// column sums + stage 2.  A block is 64 genes x 4 groups; group r sums every 4th workgroup partial (and every 4th row of
// dlogits for the bias gradient), the four are combined through LDS and the results written in their final layout.
constexpr int kDthDirect = 256;
template <typename TD>
__global__ __launch_bounds__(256) void nb_dtheta_direct(const float* __restrict__ part_dt, const int* __restrict__ part_flag,
                                                        int ngroups, int64_t G, int C, float* __restrict__ dtheta,
                                                        const TD* __restrict__ d /*dlogits or NULL*/, int64_t ld, int64_t B,
                                                        float* __restrict__ db /*or NULL*/) {
    __shared__ float red[4][4][64];
    __shared__ int flag[kDthDirect * 3];  // the partials' valid flags: read once, so that the loads below are independent
    const int lx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int64_t g = blockIdx.x * 64ll + lx;
    for (int i = threadIdx.x; i < ngroups * 3; i += 256) flag[i] = part_flag[i];
    __syncthreads();
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (g < G) {
        // (sixteen groups = up to 48 loads per trip to memory, then sixteen rows of dlogits: with four and eight the walk over
        // 128 workgroup partials and 128 rows was twelve dependent trips)
#pragma unroll 16
        for (int grp = ry; grp < ngroups; grp += 4) {
            float v[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) v[p] = flag[grp * 3 + p] ? part_dt[((int64_t)grp * 3 + p) * G + g] : 0.f;
#pragma unroll
            for (int p = 0; p < 3; ++p) s[p] += v[p];
        }
        if (db && d) {
#pragma unroll 16
            for (int64_t b = ry; b < B; b += 4) s[3] += ld_as_f32(d + b * ld + g);
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) red[ry][p][lx] = s[p];
    __syncthreads();
    if (ry != 0 || g >= G) return;
#pragma unroll
    for (int p = 0; p < 4; ++p) s[p] = (red[0][p][lx] + red[1][p][lx]) + (red[2][p][lx] + red[3][p][lx]);
    for (int j = 0; j < C; ++j) {
        float v = 0.f;
        if (j == 0) v = s[0] + s[2];
        if (j == 1) v = s[1] + s[2];
        dtheta[g * C + j] = v;
    }
    if (db) db[g] = s[3];
}
// stage 2: 64 genes x the four planes per block (a plane per wave), the 32 slice partials of a (gene, plane) requested
// together and added in slice order; the planes meet in LDS.  (One thread per gene walking 4 x 32 partials eight at a time
// was sixteen dependent trips to memory on 75 blocks: 7.5 us for 10 MB.)
__global__ __launch_bounds__(256) void nb_dtheta_stage2(const float* __restrict__ part2, int64_t G, int C,
                                                        float* __restrict__ dtheta, float* __restrict__ db /*or NULL*/) {
    __shared__ float red[4][64];
    const int lx = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int64_t g = blockIdx.x * 64ll + lx;
    float s = 0.f;
    if (g < G && (p < 3 || db)) {
        float v[kDthSlices];
#pragma unroll
        for (int sl = 0; sl < kDthSlices; ++sl) v[sl] = part2[((int64_t)sl * 4 + p) * G + g];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sl = 0; sl < kDthSlices; ++sl) s += v[sl];
    }
    red[p][lx] = s;
    __syncthreads();
    if (p != 0 || g >= G) return;
    const float s0 = red[0][lx], s1 = red[1][lx], s2 = red[2][lx];
    for (int j = 0; j < C; ++j) {
        float v = 0.f;
        if (j == 0) v = s0 + s2;
        if (j == 1) v = s1 + s2;
        dtheta[g * C + j] = v;
    }
    if (db) db[g] = red[3][lx];
}

static void launch_dtheta_stage1(const float* part_dt, const int* part_flag, int ngroups, int64_t G, float* part2,
                                 const float* part_db, hipStream_t st) {
    const bool v4 = (G & 3) == 0 && ngroups <= 8 * kDthSlices &&
                    ((reinterpret_cast<uintptr_t>(part_dt) | reinterpret_cast<uintptr_t>(part2) |
                      reinterpret_cast<uintptr_t>(part_db)) & 15) == 0;
    if (v4)
        hipLaunchKernelGGL(nb_dtheta_stage1_v4, dim3((unsigned)((G / 4 + 255) / 256), kDthSlices), dim3(256), 0, st, part_dt,
                           part_flag, ngroups, G, part2, part_db);
    else
        hipLaunchKernelGGL(nb_dtheta_stage1, dim3((unsigned)((G + 255) / 256), kDthSlices), dim3(256), 0, st, part_dt,
                           part_flag, ngroups, G, part2, part_db);
}

// Row constant of the NB log-likelihood: out[b] = sum_g goodloc * lgamma(x + 1) -- a function of the data only.
__global__ __launch_bounds__(256) void nb_row_const_kernel(const float* __restrict__ x, int64_t ldx,
                                                           const uint8_t* __restrict__ goodloc, int64_t ldg, int64_t G,
                                                           float* __restrict__ out) {
    __shared__ float red[32];
    const int64_t row = blockIdx.x;
    float acc = 0.f;
    for (int64_t c = threadIdx.x; c < G; c += 256)
        if (goodloc[row * ldg + c]) acc += lgamma_pos(x[row * ldx + c] + 1.0f);
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) out[row] = acc;
}

#ifndef C2_NB_GROUPS
#define C2_NB_GROUPS 256  // one row-walking workgroup per CU (measured: 512 -> 256 saves 22 us of partial flushes and stage-1 reads at B = 4096)
#endif
constexpr int kNbMaxGroups = C2_NB_GROUPS;

// ---- ZINB (celligner2.py:286-296, losses.py:68-124): the NB kernel's row walk with a second logit matrix pi (dropout
// head) and its gradient.  x < eps: ll = softplus(ptl) - softplus(-pi), ptl = -pi + theta (log(theta+eps) -
// log(theta+mu+eps));  x > eps: ll = NB log-likelihood - pi - softplus(-pi).  Scalar accesses, four passes per row
// (the row is re-read from L2): the general form, for rows the fast kernel below does not take (unaligned, G > 20 480).
struct ZinbExtra {
    const float* pi;
    void* dpi;
    int64_t ldpi, lddpi;
};
__device__ __forceinline__ float softplus_f(float v) { return fmaxf(v, 0.f) + log1pf(expf(-fabsf(v))); }
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

template <typename TD, int NT>
__global__ __launch_bounds__(NT) void zinb_fwd_bwd_kernel(NbArgs a, ZinbExtra z) {
    extern __shared__ __attribute__((aligned(16))) float dacc[];
    __shared__ float red[32];
    const int tid = threadIdx.x;
    const int64_t G = a.G;
    const float eps = 1e-8f;
    for (int64_t c = tid; c < G; c += NT) dacc[c] = 0.f;
    int cur_pat = -1, flushed = 0;
    float loss_acc = 0.f;
    auto flush = [&](int pat) {
        __syncthreads();
        float* dst = a.part_dt + ((int64_t)blockIdx.x * 3 + pat) * G;
        const bool first = !((flushed >> pat) & 1);
        for (int64_t c = tid; c < G; c += NT) {
            dst[c] = first ? dacc[c] : dst[c] + dacc[c];
            dacc[c] = 0.f;
        }
        flushed |= 1 << pat;
        __syncthreads();
    };
    for (int64_t row = blockIdx.x; row < a.B; row += gridDim.x) {
        int has0 = 0, has1 = 0, bad = 0;
        for (int j = 0; j < a.C; ++j) {
            const int64_t v = a.cond[row * a.C + j];
            has0 |= (v == 0);
            has1 |= (v == 1);
            bad |= (v < 0) | (v > 1) | (v >= a.C);
        }
        if (bad || (!has0 && !has1)) {
            if (tid == 0) *a.status = 1;
            continue;
        }
        const int pat = (has0 && has1) ? 2 : (has1 ? 1 : 0);
        if (pat != cur_pat) {
            if (cur_pat >= 0) flush(cur_pat);
            cur_pat = pat;
        }
        const float* orow = a.logits + row * a.ldl;
        const float* prow = z.pi + row * z.ldpi;
        const int64_t srow = a.ridx ? a.ridx[row] : row;
        const float* xrow = a.x + srow * a.ldx;
        const uint8_t* grow = a.goodloc + srow * a.ldg;
        const float sf = a.sizefactor[row];
        const float* tth = a.tab + (int64_t)(pat * 3 + 0) * nb_tab_ld(G);
        const float* tk1 = a.tab + (int64_t)(pat * 3 + 1) * nb_tab_ld(G);
        const float* tk2 = a.tab + (int64_t)(pat * 3 + 2) * nb_tab_ld(G);
        const float scale = -a.inv_B;
        float mx = -INFINITY;
        for (int64_t c = tid; c < G; c += NT) mx = fmaxf(mx, orow[c]);
        mx = block_max<NT>(mx, red);
        float sm = 0.f;
        for (int64_t c = tid; c < G; c += NT) sm += expf(orow[c] - mx);
        sm = block_sum<NT>(sm, red);
        const float inv_sum = 1.0f / sm;
        // per element: ll and its derivatives w.r.t. mu, theta, pi
        auto eval = [&](int64_t c, float& ll, float& dmu_mu, float& dth, float& dpi, float& p) {
            p = expf(orow[c] - mx) * inv_sum;
            const float mu = sf * p, th = tth[c], x = xrow[c], pv = prow[c];
            const float tm = th + mu + eps, ltm = logf(tm), lte = logf(th + eps);
            const float sp = softplus_f(-pv);
            const float ptl = -pv + th * (lte - ltm);
            const float dptl_dth = (lte - ltm) + th / (th + eps) - th / tm;
            if (x < eps) {
                const float sg = sigmoid_f(ptl);
                ll = softplus_f(ptl) - sp;
                dmu_mu = sg * (-th / tm) * mu;
                dth = sg * dptl_dth;
                dpi = -sg + sigmoid_f(-pv);
            } else if (x > eps) {
                const float zt = x + th;
                ll = tk1[c] - zt * ltm + x * logf(mu + eps) + lgamma_pos(zt) - (a.lgx ? 0.f : lgamma_pos(x + 1.0f)) - pv - sp;
                dmu_mu = x * mu / (mu + eps) - zt * mu / tm;
                dth = tk2[c] - ltm - zt / tm + digamma_pos(zt);
                dpi = -sigmoid_f(pv);
            } else {
                ll = dmu_mu = dth = dpi = 0.f;
            }
        };
        float S = 0.f;
        for (int64_t c = tid; c < G; c += NT) {
            float ll, dm, dth, dpi, p;
            eval(c, ll, dm, dth, dpi, p);
            S += grow[c] ? scale * dm : 0.f;
        }
        S = block_sum<NT>(S, red);
        TD* drow = reinterpret_cast<TD*>(a.dlogits) + row * a.lddl;
        TD* dprow = reinterpret_cast<TD*>(z.dpi) + row * z.lddpi;
        for (int64_t c = tid; c < G; c += NT) {
            float ll, dm, dth, dpi, p;
            eval(c, ll, dm, dth, dpi, p);
            const bool m = grow[c];
            const float pg = m ? scale * dm : 0.f;
            st_from_f32<TD>(drow + c, pg - p * S);
            st_from_f32<TD>(dprow + c, m ? scale * dpi : 0.f);
            dacc[c] += m ? scale * tth[c] * dth : 0.f;
            loss_acc += m ? ll : 0.f;
        }
        if (a.lgx && tid == 0) loss_acc -= a.lgx[row];
    }
    if (cur_pat >= 0) flush(cur_pat);
    loss_acc = block_sum<NT>(loss_acc, red);
    if (tid == 0) {
        a.part_loss[blockIdx.x] = -a.inv_B * loss_acc;
#pragma unroll
        for (int p = 0; p < 3; ++p) a.part_flag[blockIdx.x * 3 + p] = (flushed >> p) & 1;
    }
}

// ---- ZINB fast path (16-byte aligned rows, G % 4 == 0, G <= 4 NT NV) ---------------------------------------------------
// The NB fast kernel's shape without its hand-tuning: one workgroup walks whole rows, thread t owns genes 4(t + NT i)..+3
// of chunk i in every row.  The row's logits (then exp(logit - max)) and p * dloss/dp stay in REGISTERS between the
// passes, so every input byte is read once and every transcendental is taken once; the dispersion-gradient accumulator
// of the workgroup sits in LDS at thread-private addresses (no barrier); chunk i + 1's inputs (counts, mask, dropout
// logits, four table values) are requested before chunk i is evaluated; rows are addressed through buffer resources
// (a lane past the last gene loads zeros, its stores are dropped).  Table: [3 patterns][4][Gp] = {theta,
// log(theta + eps), K1, K2} (zinb_table_kernel; the padding holds theta = 1, 0, 0, 0).
// Per gene: exp of the softmax + 1/(theta+mu+eps), log(theta+mu+eps), exp(-|pi|), log1p and reciprocal of it (softplus
// and sigmoid of pi from ONE exponential) + the zero branch (sigmoid / softplus of ptl: exp, log, rcp; theta/(theta+eps))
// + the count branch (log(mu+eps), its reciprocal, lgamma / digamma of x + theta with one shared shift, lgamma(x + 1)):
// both branches are evaluated for every gene and selected -- a wave of 256 genes all but never holds only one kind.
__global__ void zinb_table_kernel(const float* __restrict__ theta, int C, int64_t G, int64_t Gp, float* __restrict__ tab,
                                  const int64_t* __restrict__ cond, int64_t B, int* __restrict__ rowflags) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int64_t r = g; r < B; r += (int64_t)gridDim.x * blockDim.x) {
        int f = 0;
        for (int j = 0; j < C; ++j) {
            const int64_t v = cond[r * C + j];
            f |= (v == 0 ? 1 : 0) | (v == 1 ? 2 : 0) | (((v < 0) | (v > 1) | (v >= C)) ? 4 : 0);
        }
        rowflags[r] = f;
    }
    if (g >= Gp) return;
    const bool pad = g >= G;
    const float t0 = pad ? 0.f : theta[g * C + 0];
    const float t1 = (pad || C < 2) ? 0.f : theta[g * C + 1];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float t = p == 0 ? t0 : (p == 1 ? t1 : t0 + t1);
        const float th = expf(t);
        const float lte = logf(th + 1e-8f);
        tab[(p * 4 + 0) * Gp + g] = pad ? 1.f : th;
        tab[(p * 4 + 1) * Gp + g] = pad ? 0.f : lte;
        tab[(p * 4 + 2) * Gp + g] = pad ? 0.f : th * lte - lgamma_pos(th);
        tab[(p * 4 + 3) * Gp + g] = pad ? 0.f : lte + th / (th + 1e-8f) - digamma_pos(th);
    }
}

struct ZinbGene { float ll, dmu_mu, dth, dpi; };
// one gene: mu = sizefactor * softmax, th / lte / k1 / k2 from the table, x the count, pv the dropout logit
template <bool HAS_LGX>  // the caller subtracts the row's sum of lgamma(x + 1) itself (a constant of the data)
__device__ __forceinline__ ZinbGene zinb_gene(float mu, float th, float lte, float k1, float k2, float x, float pv) {
    const float eps = 1e-8f;
    const float tm = th + mu + eps;
    const float rtm = frcp(tm), ltm = flog(tm);
    // softplus(-pv), sigmoid(pv), sigmoid(-pv) from t = exp(-|pv|)
    const float t = fexp(-fabsf(pv));
    const float l1t = t < 2.44140625e-4f ? t - 0.5f * t * t : flog(1.0f + t);
    const float r1 = frcp(1.0f + t);
    const bool ppos = pv >= 0.f;
    const float sig_p = ppos ? r1 : t * r1;   // sigmoid(pv)
    const float sig_n = ppos ? t * r1 : r1;   // sigmoid(-pv)
    const float sp = fmaxf(-pv, 0.f) + l1t;   // softplus(-pv)
    // zero branch (x < eps): ll = softplus(ptl) - softplus(-pv), ptl = -pv + theta (log(theta+eps) - log(theta+mu+eps))
    const float dl = lte - ltm;
    const float ptl = fmaf(th, dl, -pv);
    const float u = fexp(-fabsf(ptl));
    const float l1u = u < 2.44140625e-4f ? u - 0.5f * u * u : flog(1.0f + u);
    const float ru = frcp(1.0f + u);
    const float sg = ptl >= 0.f ? ru : u * ru;  // sigmoid(ptl)
    const float thr = th * rtm;
    ZinbGene zr;
    zr.ll = fmaxf(ptl, 0.f) + l1u - sp;
    zr.dmu_mu = -sg * thr * mu;
    zr.dth = sg * (dl + th * frcp(th + eps) - thr);
    zr.dpi = sig_n - sg;
    // count branch (x > eps): NB log-likelihood - pv - softplus(-pv)
    const float zt = x + th;
    float lg, dg;
    lgam_digam_fast(zt, lg, dg);
    const float mue = mu + eps;
    ZinbGene ct;
    ct.ll = k1 - zt * ltm + x * flog(mue) + lg - (HAS_LGX ? 0.f : lgamma_fast(x + 1.0f)) - pv - sp;
    ct.dmu_mu = x * mu * frcp(mue) - zt * mu * rtm;
    ct.dth = k2 - ltm - zt * rtm + dg;
    ct.dpi = -sig_p;
    const bool zero = x < eps, pos = x > eps;
    ZinbGene o;
    o.ll = zero ? zr.ll : (pos ? ct.ll : 0.f);
    o.dmu_mu = zero ? zr.dmu_mu : (pos ? ct.dmu_mu : 0.f);
    o.dth = zero ? zr.dth : (pos ? ct.dth : 0.f);
    o.dpi = zero ? zr.dpi : (pos ? ct.dpi : 0.f);
    return o;
}

template <typename TD, int NT, int NV, bool HAS_LGX>
__global__ __launch_bounds__(NT) void zinb_rowreg_kernel(NbArgs a, ZinbExtra z) {
    extern __shared__ __attribute__((aligned(16))) float dacc[];  // [NV][NT] chunks of four: d(theta) accumulator
    __shared__ float red[32];
    const int tid = threadIdx.x;
    const int Gi = (int)a.G;
    const int64_t Gp = nb_tab_ld(a.G);
    f32x4* acc = reinterpret_cast<f32x4*>(dacc);
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i * NT + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur_pat = -1, flushed = 0, red_half = 0;
    float loss_acc = 0.f;
    auto flush = [&](int pat) {  // every thread owns its genes: no barrier
        float* dst = a.part_dt + ((int64_t)blockIdx.x * 3 + pat) * a.G;
        const bool first = !((flushed >> pat) & 1);
        // (every load of the partial written so far before the first store: see the NB kernel's flush)
        f32x4 old[NV];
        if (!first) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = 4 * (tid + NT * i);
                old[i] = *reinterpret_cast<const f32x4*>(dst + (c < Gi ? c : 0));
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // s_waitcnt vmcnt(0) where the compiler's counter sees it
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * (tid + NT * i);
            if (c < Gi) {
                f32x4 v = acc[i * NT + tid];
                if (!first) v += old[i];
                *reinterpret_cast<f32x4*>(dst + c) = v;
                acc[i * NT + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        flushed |= 1 << pat;
    };
    constexpr unsigned kRsrcFlags = 0x00020000u;  // raw buffer, 32-bit data format (gfx9)
    auto row_rsrc = [&](const void* base, int bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, kRsrcFlags);
    };
    const int v16 = tid * 16;
    auto ld16 = [&](__amdgpu_buffer_rsrc_t r, int i) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, v16, i * NT * 16, 0));
    };
    const int always = Gi + 4 * NT * NV + 8;  // see nb_rowreg_kernel: real branches keep the chunks' temporaries apart
    const float scale = -a.inv_B;
    for (int64_t row = blockIdx.x; row < a.B; row += gridDim.x) {
        const int fl = a.rowflags[row];
        if ((fl & 4) || !(fl & 3)) {
            if (tid == 0) *a.status = 1;
            continue;
        }
        const int pat = (fl & 3) == 3 ? 2 : ((fl & 2) ? 1 : 0);
        if (pat != cur_pat) {
            if (cur_pat >= 0) flush(cur_pat);
            cur_pat = pat;
        }
        const int64_t srow = a.ridx ? a.ridx[row] : row;
        const auto ro = row_rsrc(a.logits + row * a.ldl, Gi * 4);
        const auto rp = row_rsrc(z.pi + row * z.ldpi, Gi * 4);
        const auto rx = row_rsrc(a.x + srow * a.ldx, Gi * 4);
        const auto rg = row_rsrc(a.goodloc + srow * a.ldg, Gi);
        const auto rt = row_rsrc(a.tab + (int64_t)pat * 4 * Gp, (int)(4 * Gp * 4));
        const auto rdl = row_rsrc(reinterpret_cast<TD*>(a.dlogits) + row * a.lddl, Gi * (int)sizeof(TD));
        const auto rdp = row_rsrc(reinterpret_cast<TD*>(z.dpi) + row * z.lddpi, Gi * (int)sizeof(TD));
        const float sf = a.sizefactor[row];
        f32x4 ev[NV], pg[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) ev[i] = ld16(ro, i);
        // inputs of a chunk: counts, dropout logits, mask, theta / log / K1 / K2
        f32x4 qx, qp, q0, q1, q2, q3;
        unsigned qm;
        auto request = [&](int i) {
            qx = ld16(rx, i);
            qp = ld16(rp, i);
            qm = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rg, tid * 4, i * NT * 4, 0);
            q0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, v16, i * NT * 16, 0));
            q1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, v16 + (int)(Gp * 4), i * NT * 16, 0));
            q2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, v16 + (int)(Gp * 8), i * NT * 16, 0));
            q3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, v16 + (int)(Gp * 12), i * NT * 16, 0));
        };
        request(0);
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (4 * (tid + NT * i) >= Gi) ev[i] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            mx = fmaxf(fmaxf(mx, fmaxf(ev[i][0], ev[i][1])), fmaxf(ev[i][2], ev[i][3]));
        }
        mx = nb_block_reduce<true, NT>(mx, red, red_half);
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ev[i][k] = fexp(ev[i][k] - mx);
                sm += ev[i][k];
            }
        }
        sm = nb_block_reduce<false, NT>(sm, red, red_half);
        const float inv_sum = 1.0f / sm;
        const float cmu = sf * inv_sum;
        float S = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 cx = qx, cp = qp, c0 = q0, c1 = q1, c2 = q2, c3 = q3;
            const unsigned cm = qm;
            if (i + 1 < NV) request(i + 1);
            if (4 * (tid + NT * i) < always) {
                f32x4 dp4, dt4;
#pragma unroll
                for (int k2 = 0; k2 < 4; k2 += 2) {
                    if (4 * (tid + NT * i) + k2 < always) {  // (a branch per gene pair, for the same reason)
#pragma unroll
                        for (int k = k2; k < k2 + 2; ++k) {
                            const bool m = (cm >> (8 * k)) & 0xffu;
                            const ZinbGene gq = zinb_gene<HAS_LGX>(ev[i][k] * cmu, c0[k], c1[k], c2[k], c3[k], cx[k], cp[k]);
                            const float pgk = m ? scale * gq.dmu_mu : 0.f;
                            pg[i][k] = pgk;
                            S += pgk;
                            dp4[k] = m ? scale * gq.dpi : 0.f;
                            dt4[k] = m ? scale * c0[k] * gq.dth : 0.f;
                            loss_acc += m ? gq.ll : 0.f;
                        }
                    }
                }
                acc[i * NT + tid] += dt4;
                if (sizeof(TD) == 4) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dp4), rdp, v16, i * NT * 16, 0);
                    asm volatile("s_nop 1" ::: "memory");  // see nb_rowreg_kernel: store-data hazard
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    u32x2 pk;
                    pk[0] = (unsigned)f2bf(dp4[0]) | ((unsigned)f2bf(dp4[1]) << 16);
                    pk[1] = (unsigned)f2bf(dp4[2]) | ((unsigned)f2bf(dp4[3]) << 16);
                    __builtin_amdgcn_raw_buffer_store_b64(pk, rdp, tid * 8, i * NT * 8, 0);
                }
            }
        }
        S = nb_block_reduce<false, NT>(S, red, red_half);
        if (HAS_LGX && tid == 0) loss_acc -= a.lgx[row];
        const float pS = inv_sum * S;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = fmaf(-ev[i][k], pS, pg[i][k]);
            if (sizeof(TD) == 4) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), rdl, v16, i * NT * 16, 0);
                asm volatile("s_nop 1" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else {
                u32x2 pk;
                pk[0] = (unsigned)f2bf(d[0]) | ((unsigned)f2bf(d[1]) << 16);
                pk[1] = (unsigned)f2bf(d[2]) | ((unsigned)f2bf(d[3]) << 16);
                __builtin_amdgcn_raw_buffer_store_b64(pk, rdl, tid * 8, i * NT * 8, 0);
            }
        }
    }
    if (cur_pat >= 0) flush(cur_pat);
    __syncthreads();
    loss_acc = block_sum<NT>(loss_acc, red);
    if (tid == 0) {
        a.part_loss[blockIdx.x] = -a.inv_B * loss_acc;
#pragma unroll
        for (int p = 0; p < 3; ++p) a.part_flag[blockIdx.x * 3 + p] = (flushed >> p) & 1;
    }
}

// ---- MSE ---------------------------------------------------------------------------------------------------
template <typename TD>
__global__ __launch_bounds__(256) void mse_fwd_bwd_kernel(const float* __restrict__ logits, int64_t ldl,
                                                          const float* __restrict__ x, int64_t ldx,
                                                          const uint8_t* __restrict__ goodloc, int64_t ldg, int64_t B,
                                                          int64_t G, float inv_B, TD* __restrict__ dl, int64_t lddl,
                                                          float* __restrict__ partials) {
    __shared__ float red[32];
    float acc = 0.f;
    const int64_t total = B * G;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / G, g = i - b * G;
        const float o = logits[b * ldl + g];
        const float r = fmaxf(o, 0.f);
        const float m = goodloc[b * ldg + g] ? 1.f : 0.f;
        const float d = r - x[b * ldx + g];
        acc += m * d * d;
        st_from_f32<TD>(dl + b * lddl + g, (o > 0.f) ? 2.f * inv_B * m * d : 0.f);
    }
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc * inv_B;
}

// ---- classifier loss: wave per row -------------------------------------------------------------------------
template <typename TD>
__global__ __launch_bounds__(256) void classifier_loss_kernel(const float* __restrict__ u, int64_t ldu,
                                                              int64_t* __restrict__ classes, int64_t B, int P,
                                                              float bc, TD* __restrict__ du, int64_t lddu,
                                                              float* __restrict__ partials) {
    __shared__ float red[32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc = 0.f;
    for (int64_t row = blockIdx.x * 4ll + wv; row < B; row += gridDim.x * 4ll) {
        const float* ur = u + row * ldu;
        int64_t* yr = classes + row * P;
        float mx = -INFINITY;
        for (int c = lane; c < P; c += 64) mx = fmaxf(mx, ur[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int c = lane; c < P; c += 64) se += expf(ur[c] - mx);
        se = wave_sum(se);
        const float lse = mx + logf(se);
        float ent = 0.f, ce = 0.f;
        for (int c = lane; c < P; c += 64) {
            const float v = ur[c];
            const float logp = v - lse;
            ent -= expf(logp) * logp;
            const int64_t y = yr[c];
            const float w = (y == -1) ? 0.f : 1.f;
            const float yt = (y == -1) ? 0.f : (float)y;
            const float sp = fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));
            ce += w * (sp - yt * v);
        }
        ent = wave_sum(ent);
        ce = wave_sum(ce);
        for (int c = lane; c < P; c += 64) {
            const float v = ur[c];
            const float logp = v - lse;
            const float p = expf(logp);
            const int64_t y = yr[c];
            const float w = (y == -1) ? 0.f : 1.f;
            const float yt = (y == -1) ? 0.f : (float)y;
            const float sig = 1.0f / (1.0f + expf(-v));
            const float g = bc * w * (sig - yt) + (1.f - bc) * (-p * (logp + ent));
            st_from_f32<TD>(du + row * lddu + c, g);
            if (y == -1) yr[c] = 0;  // the reference mutates the caller's tensor (losses.py:324-325)
        }
        if (lane == 0) acc += bc * ce + (1.f - bc) * ent;
    }
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

}  // namespace c2

using namespace c2;

extern "C" size_t c2_recon_nb_ws(int64_t B, int64_t G, int C) {
    (void)C;
    const int64_t ng = std::min<int64_t>(std::max<int64_t>(B, 1), kNbMaxGroups);
    // table 9G (12G: ZINB's fast kernel) + partial dtheta ng*3*G + flags ng*3 (as floats) + loss partials ng
    // ... + dtheta slices + the row classification (B ints)
    // ... + the workgroups' column sums of dlogits (c2_recon_nb_fwd_bwd_dbias)
    return (size_t)(12 * nb_tab_ld(G) + ng * 3 * G + ng * 3 + ng + 64 + (int64_t)kDthSlices * 4 * G + B + 64 + ng * G + 64);
}

// Workgroup size and chunks per thread of the fast kernel for G genes, or {0, 0}: the general kernel runs.
struct NbFast { int nt, nv; };
static NbFast nb_fast_shape(int64_t G, bool vec, int64_t B) {
    const size_t lds = (size_t)((G + 3) & ~int64_t(3)) * sizeof(float);
    if (!vec || 2 * lds + 256 > 160 * 1024) return {0, 0};  // two LDS rows must fit the CU (G <= 20 448)
    if (G <= 1024) return {128, 2};  // (the rolling prefetch needs at least two chunks)
    if (G <= 2048) return {256, 2};
    if (G <= 4096) return {256, 4};
    if (G <= 8192) return {1024, 2};
    if (G <= 12288) return {1024, 3};
    if (G <= 15360) return {768, 5};
    if (G <= 18432) return {768, 6};
    // (measured at 4096 x 19000: 768 x 7 320 us; 1024 x 5, four waves per SIMD but 128 registers and 22 of them spilled,
    // 411 us)
    // (same box, 4096 x 19000: 768 x 7 with theta re-read per row 238 us, 512 x 10 with theta and the accumulator in
    // registers 232 us; C2_NB_REGT=0 selects the former: A/B and tests)
    // (a workgroup that walks a single row pays for loading theta and flushing the registers without using them twice:
    // 27 against 22 us at B = 128, so the register form needs more rows than workgroups)
    const char* e = getenv("C2_NB_REGT");
    if (e == nullptr ? B > kNbMaxGroups : atoi(e) != 0) return {512, 10};
    return {768, 7};
}

// ZINB fast kernel: 4 NT NV >= G (and <= the table's padded row, nb_tab_ld)
static NbFast zinb_fast_shape(int64_t G, bool vec) {
    if (!vec) return {0, 0};
    if (G <= 1024) return {128, 2};
    if (G <= 2048) return {256, 2};
    if (G <= 4096) return {256, 4};
    if (G <= 8192) return {512, 4};
    if (G <= 12288) return {512, 6};
    if (G <= 16384) return {512, 8};
    if (G <= 20480) return {512, 10};
    return {0, 0};
}

// the fast kernel on half-precision logits: the two shapes of 18 432 < G <= 20 480 genes (768 x 7, 512 x 10), bf16 dlogits
static bool nb_logits_f16_shape(int64_t G, bool vec, int64_t B) {
    const NbFast f = nb_fast_shape(G, vec, B);
    return (f.nt == 768 && f.nv == 7) || (f.nt == 512 && f.nv == 10);
}
static int nb_launch_lh(NbArgs a, int ngroups, hipStream_t st) {
    const size_t l = 2 * (size_t)((a.G + 3) & ~int64_t(3)) * sizeof(float);
    const NbFast f = nb_fast_shape(a.G, true, a.B);
#define NB_H(NT, NV)                                                                                          \
    do {                                                                                                      \
        constexpr bool regt = (NT) == 512;                                                                     \
        auto kf = nb_rowreg_kernel<bf16_t, NT, NV, false, regt, true>;                                         \
        auto kl = nb_rowreg_kernel<bf16_t, NT, NV, true, regt, true>;                                          \
        (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);              \
        (void)hipFuncSetAttribute((const void*)kl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);              \
        if (a.lgx) hipLaunchKernelGGL(kl, dim3(ngroups), dim3(NT), l, st, a);                                  \
        else hipLaunchKernelGGL(kf, dim3(ngroups), dim3(NT), l, st, a);                                        \
        return 0;                                                                                             \
    } while (0)
    if (f.nt == 768 && f.nv == 7) NB_H(768, 7);
    if (f.nt == 512 && f.nv == 10) NB_H(512, 10);
#undef NB_H
    return 1;
}

template <typename TD>
static int nb_launch(NbArgs a, int ngroups, bool vec, hipStream_t st) {
    const size_t lds = (size_t)((a.G + 3) & ~int64_t(3)) * sizeof(float);
#define NB_L(NT)                                                                                              \
    do {                                                                                                      \
        auto kv = nb_fwd_bwd_kernel<TD, NT, true>;                                                             \
        auto ks = nb_fwd_bwd_kernel<TD, NT, false>;                                                            \
        if (lds > 48 * 1024) {                                                                                \
            (void)hipFuncSetAttribute((const void*)kv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
            (void)hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        }                                                                                                     \
        if (vec) hipLaunchKernelGGL(kv, dim3(ngroups), dim3(NT), lds, st, a);                                  \
        else hipLaunchKernelGGL(ks, dim3(ngroups), dim3(NT), lds, st, a);                                      \
    } while (0)
#define NB_F(NT, NV)                                                                                          \
    do {                                                                                                      \
        constexpr bool regt = (NT) == 512;                                                                     \
        auto kf = nb_rowreg_kernel<TD, NT, NV, false, regt>;                                                   \
        auto kl = nb_rowreg_kernel<TD, NT, NV, true, regt>;                                                    \
        const size_t l = 2 * lds; /* accumulator (or, regt, the column sums of dlogits) + the row of p * dloss/dp */\
        if (l > 48 * 1024) {                                                                                  \
            (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);          \
            (void)hipFuncSetAttribute((const void*)kl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);          \
        }                                                                                                     \
        if (a.lgx) hipLaunchKernelGGL(kl, dim3(ngroups), dim3(NT), l, st, a);                                  \
        else hipLaunchKernelGGL(kf, dim3(ngroups), dim3(NT), l, st, a);                                        \
        return 0;                                                                                             \
    } while (0)
    const NbFast f = nb_fast_shape(a.G, vec, a.B);
#define NB_FC(NT, NV) if (f.nt == NT && f.nv == NV) NB_F(NT, NV)
    NB_FC(128, 2);
    NB_FC(256, 2);
    NB_FC(256, 4);
    NB_FC(1024, 2);
    NB_FC(1024, 3);
    NB_FC(768, 5);
    NB_FC(768, 6);
    NB_FC(768, 7);
    NB_FC(512, 10);
#undef NB_FC
    if (a.G <= 4096) NB_L(256);
    else NB_L(1024);
#undef NB_L
#undef NB_F
    return 0;
}

static int recon_nb_run(const void* logits_any, int l_dtype, int64_t ldl, const float* x, int64_t ldx,
                                   const uint8_t* goodloc, int64_t ldg, const float* sizefactor, const float* theta,
                                   const int64_t* cond, int C, int64_t B, int64_t G, float inv_B, float* loss_out,
                                   void* dlogits, int dl_dtype, int64_t lddl, float* dtheta, float* ws,
                                   int32_t* status_dev, const float* lgx1_rowsum, const int64_t* row_index,
                                   float* dbias, void* stream) {
    const float* logits = reinterpret_cast<const float*>(logits_any);
    C2_CHECK(logits && x && goodloc && sizefactor && theta && cond && loss_out && dlogits && dtheta && ws && status_dev,
             "null pointer");
    C2_CHECK(dl_dtype == C2_F32 || dl_dtype == C2_BF16, "bad dtype");
    C2_CHECK(l_dtype == C2_F32 || l_dtype == C2_F16, "logits are fp32 or IEEE halves");
    C2_CHECK(C >= 1, "NB needs at least one condition column (theta is [G,C])");
    C2_CHECK(ldl >= G && ldx >= G && ldg >= G && lddl >= G, "leading dimension too small");
    C2_CHECK(G <= 40000, "G > 40000 genes is not supported by the NB kernel (its LDS accumulator holds G floats)");
    if (B == 0 || G == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int ngroups = (int)std::min<int64_t>(B, kNbMaxGroups);
    float* tab = ws;
    float* part_dt = tab + 9 * nb_tab_ld(G);
    int* part_flag = reinterpret_cast<int*>(part_dt + (int64_t)ngroups * 3 * G);
    float* part_loss = reinterpret_cast<float*>(part_flag + ngroups * 3);
    float* part2 = part_loss + ngroups + 32;
    int* rowflags = reinterpret_cast<int*>(part2 + (int64_t)kDthSlices * 4 * G) + 16;
    float* part_db = reinterpret_cast<float*>(rowflags + B + 16);
    part_db += (4 - ((uintptr_t)part_db >> 2 & 3)) & 3;  // 16-byte aligned when ws is
    const bool vec = (G % 4 == 0) && (ldl % 4 == 0) && (ldx % 4 == 0) && (ldg % 4 == 0) &&
                     (lddl % 4 == 0) && aligned16(logits) && aligned16(x) && aligned16(dlogits) &&
                     (((uintptr_t)goodloc & 3) == 0) && aligned16(ws);
    hipLaunchKernelGGL(nb_table_kernel, dim3((unsigned)(nb_tab_ld(G) / 256)), dim3(256), 0, st, theta, C, G, nb_tab_ld(G), tab,
                       cond, B, rowflags, status_dev);
    NbArgs a{logits, x, goodloc, sizefactor, cond, tab, dlogits, part_dt, part_flag, part_loss, status_dev,
             lgx1_rowsum, row_index, ldl, ldx, ldg, lddl, B, G, C, inv_B, rowflags, nullptr};
    // the loss kernel forms the column sums itself in its register-resident form; otherwise a column-sum pass follows
    const bool db_in_kernel = dbias && nb_fast_shape(G, vec, B).nt == 512;
    if (db_in_kernel) a.part_db = part_db;
    int rc;
    if (l_dtype == C2_F16) {
        C2_CHECK(dl_dtype == C2_BF16 && nb_logits_f16_shape(G, vec, B),
                 "half-precision logits: bf16 dlogits, 16-byte aligned rows and 18 432 < G <= 20 480 only "
                 "(c2_recon_nb_logits_f16_ok); hand fp32 logits otherwise");
        rc = nb_launch_lh(a, ngroups, st);
    } else {
        rc = dl_dtype == C2_BF16 ? nb_launch<bf16_t>(a, ngroups, vec, st) : nb_launch<float>(a, ngroups, vec, st);
    }
    C2_CHECK(rc == 0, "unsupported G");
    if (ngroups <= kDthDirect && B <= kDthDirect && !db_in_kernel) {
        const dim3 grid((unsigned)((G + 63) / 64));
        if (dl_dtype == C2_BF16)
            hipLaunchKernelGGL(nb_dtheta_direct<bf16_t>, grid, dim3(256), 0, st, part_dt, part_flag, ngroups, G, C, dtheta,
                               (const bf16_t*)dlogits, lddl, B, dbias);
        else
            hipLaunchKernelGGL(nb_dtheta_direct<float>, grid, dim3(256), 0, st, part_dt, part_flag, ngroups, G, C, dtheta,
                               (const float*)dlogits, lddl, B, dbias);
        launch_finalize_sum(part_loss, ngroups, loss_out, st);
        C2_LAUNCH_CHECK();
        return 0;
    }
    launch_dtheta_stage1(part_dt, part_flag, ngroups, G, part2, db_in_kernel ? part_db : (const float*)nullptr, st);
    if (dbias && !db_in_kernel) {
        const dim3 grid((unsigned)((G + 255) / 256), kDthSlices);
        if (dl_dtype == C2_BF16)
            hipLaunchKernelGGL(nb_colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dlogits, lddl, B, G, part2);
        else
            hipLaunchKernelGGL(nb_colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)dlogits, lddl, B, G, part2);
    }
    hipLaunchKernelGGL(nb_dtheta_stage2, dim3((unsigned)((G + 63) / 64)), dim3(256), 0, st, part2, G, C, dtheta, dbias);
    launch_finalize_sum(part_loss, ngroups, loss_out, st);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_recon_nb_fwd_bwd(const float* logits, int64_t ldl, const float* x, int64_t ldx,
                                   const uint8_t* goodloc, int64_t ldg, const float* sizefactor, const float* theta,
                                   const int64_t* cond, int C, int64_t B, int64_t G, float inv_B, float* loss_out,
                                   void* dlogits, int dl_dtype, int64_t lddl, float* dtheta, float* ws,
                                   int32_t* status_dev, const float* lgx1_rowsum, const int64_t* row_index,
                                   void* stream) {
    return recon_nb_run(logits, C2_F32, ldl, x, ldx, goodloc, ldg, sizefactor, theta, cond, C, B, G, inv_B, loss_out, dlogits,
                        dl_dtype, lddl, dtheta, ws, status_dev, lgx1_rowsum, row_index, nullptr, stream);
}

extern "C" int c2_recon_nb_logits_f16_ok(int64_t B, int64_t G) { return nb_logits_f16_shape(G, G % 4 == 0, B) ? 1 : 0; }

extern "C" int c2_recon_nb_fwd_bwd_dbias(const void* logits, int l_dtype, int64_t ldl, const float* x, int64_t ldx,
                                         const uint8_t* goodloc, int64_t ldg, const float* sizefactor,
                                         const float* theta, const int64_t* cond, int C, int64_t B, int64_t G,
                                         float inv_B, float* loss_out, void* dlogits, int dl_dtype, int64_t lddl,
                                         float* dtheta, float* ws, int32_t* status_dev, const float* lgx1_rowsum,
                                         const int64_t* row_index, float* dbias, void* stream) {
    C2_CHECK(dbias, "null pointer");
    return recon_nb_run(logits, l_dtype, ldl, x, ldx, goodloc, ldg, sizefactor, theta, cond, C, B, G, inv_B, loss_out, dlogits,
                        dl_dtype, lddl, dtheta, ws, status_dev, lgx1_rowsum, row_index, dbias, stream);
}

static int recon_zinb_run(const float* logits, int64_t ldl, const float* pi, int64_t ldpi, const float* x,
                                     int64_t ldx, const uint8_t* goodloc, int64_t ldg, const float* sizefactor,
                                     const float* theta, const int64_t* cond, int C, int64_t B, int64_t G, float inv_B,
                                     float* loss_out, void* dlogits, void* dpi, int d_dtype, int64_t lddl,
                                     int64_t lddpi, float* dtheta, float* ws, int32_t* status_dev,
                                     const int64_t* row_index, const float* lgx1_rowsum, void* stream) {
    C2_CHECK(logits && pi && x && goodloc && sizefactor && theta && cond && loss_out && dlogits && dpi && dtheta && ws &&
                 status_dev, "null pointer");
    C2_CHECK(d_dtype == C2_F32 || d_dtype == C2_BF16, "bad dtype");
    C2_CHECK(C >= 1, "ZINB needs at least one condition column (theta is [G,C])");
    C2_CHECK(ldl >= G && ldpi >= G && ldx >= G && ldg >= G && lddl >= G && lddpi >= G, "leading dimension too small");
    C2_CHECK(G <= 40000, "G > 40000 genes is not supported (the LDS accumulator holds G floats)");
    if (B == 0 || G == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int ngroups = (int)std::min<int64_t>(B, kNbMaxGroups);
    const int64_t Gp = nb_tab_ld(G);
    const bool vec = (G % 4 == 0) && (ldl % 4 == 0) && (ldpi % 4 == 0) && (ldx % 4 == 0) && (ldg % 4 == 0) &&
                     (lddl % 4 == 0) && (lddpi % 4 == 0) && aligned16(logits) && aligned16(pi) && aligned16(x) &&
                     aligned16(dlogits) && aligned16(dpi) && (((uintptr_t)goodloc & 3) == 0) && aligned16(ws);
    const NbFast f = zinb_fast_shape(G, vec);
    float* tab = ws;
    float* part_dt = tab + (f.nt ? 12 : 9) * Gp;
    int* part_flag = reinterpret_cast<int*>(part_dt + (int64_t)ngroups * 3 * G);
    float* part_loss = reinterpret_cast<float*>(part_flag + ngroups * 3);
    ZinbExtra z{pi, dpi, ldpi, lddpi};
    if (f.nt) {
        int* rowflags = reinterpret_cast<int*>(part_loss + ngroups + 32 + (int64_t)kDthSlices * 4 * G) + 16;
        hipLaunchKernelGGL(zinb_table_kernel, dim3((unsigned)(Gp / 256)), dim3(256), 0, st, theta, C, G, Gp, tab, cond, B,
                           rowflags);
        NbArgs a{logits, x, goodloc, sizefactor, cond, tab, dlogits, part_dt, part_flag, part_loss, status_dev,
                 lgx1_rowsum, row_index, ldl, ldx, ldg, lddl, B, G, C, inv_B, rowflags, nullptr};
#define ZINB_K(TD, NT, NV, LGX)                                                                               \
    do {                                                                                                      \
        auto k = zinb_rowreg_kernel<TD, NT, NV, LGX>;                                                          \
        if (l > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l); \
        hipLaunchKernelGGL(k, dim3(ngroups), dim3(NT), l, st, a, z);                                           \
    } while (0)
#define ZINB_F(NT, NV)                                                                                        \
    if (f.nt == NT && f.nv == NV) {                                                                           \
        const size_t l = (size_t)NT * NV * 16;                                                                \
        if (d_dtype == C2_BF16) {                                                                             \
            if (lgx1_rowsum) ZINB_K(bf16_t, NT, NV, true); else ZINB_K(bf16_t, NT, NV, false);                 \
        } else {                                                                                              \
            if (lgx1_rowsum) ZINB_K(float, NT, NV, true); else ZINB_K(float, NT, NV, false);                   \
        }                                                                                                     \
    }
        ZINB_F(128, 2)
        ZINB_F(256, 2)
        ZINB_F(256, 4)
        ZINB_F(512, 4)
        ZINB_F(512, 6)
        ZINB_F(512, 8)
        ZINB_F(512, 10)
#undef ZINB_F
#undef ZINB_K
    } else {
    hipLaunchKernelGGL(nb_table_kernel, dim3((unsigned)(nb_tab_ld(G) / 256)), dim3(256), 0, st, theta, C, G, nb_tab_ld(G), tab,
                       (const int64_t*)nullptr, (int64_t)0, (int*)nullptr, (int32_t*)nullptr);
    NbArgs a{logits, x, goodloc, sizefactor, cond, tab, dlogits, part_dt, part_flag, part_loss, status_dev,
             lgx1_rowsum, row_index, ldl, ldx, ldg, lddl, B, G, C, inv_B};
    const size_t lds = (size_t)G * sizeof(float);
    constexpr int NT = 512;
    if (d_dtype == C2_BF16) {
        auto k = zinb_fwd_bwd_kernel<bf16_t, NT>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(ngroups), dim3(NT), lds, st, a, z);
    } else {
        auto k = zinb_fwd_bwd_kernel<float, NT>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(ngroups), dim3(NT), lds, st, a, z);
    }
    }
    float* part2 = part_loss + ngroups + 32;
    launch_dtheta_stage1(part_dt, part_flag, ngroups, G, part2, (const float*)nullptr, st);
    hipLaunchKernelGGL(nb_dtheta_stage2, dim3((unsigned)((G + 63) / 64)), dim3(256), 0, st, part2, G, C, dtheta,
                       (float*)nullptr);
    launch_finalize_sum(part_loss, ngroups, loss_out, st);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_recon_zinb_fwd_bwd(const float* logits, int64_t ldl, const float* pi, int64_t ldpi, const float* x,
                                     int64_t ldx, const uint8_t* goodloc, int64_t ldg, const float* sizefactor,
                                     const float* theta, const int64_t* cond, int C, int64_t B, int64_t G, float inv_B,
                                     float* loss_out, void* dlogits, void* dpi, int d_dtype, int64_t lddl,
                                     int64_t lddpi, float* dtheta, float* ws, int32_t* status_dev,
                                     const int64_t* row_index, void* stream) {
    return recon_zinb_run(logits, ldl, pi, ldpi, x, ldx, goodloc, ldg, sizefactor, theta, cond, C, B, G, inv_B, loss_out,
                          dlogits, dpi, d_dtype, lddl, lddpi, dtheta, ws, status_dev, row_index, nullptr, stream);
}

extern "C" int c2_recon_zinb_fwd_bwd_rc(const float* logits, int64_t ldl, const float* pi, int64_t ldpi, const float* x,
                                        int64_t ldx, const uint8_t* goodloc, int64_t ldg, const float* sizefactor,
                                        const float* theta, const int64_t* cond, int C, int64_t B, int64_t G, float inv_B,
                                        float* loss_out, void* dlogits, void* dpi, int d_dtype, int64_t lddl,
                                        int64_t lddpi, float* dtheta, float* ws, int32_t* status_dev,
                                        const int64_t* row_index, const float* lgx1_rowsum, void* stream) {
    C2_CHECK(lgx1_rowsum, "null pointer");
    return recon_zinb_run(logits, ldl, pi, ldpi, x, ldx, goodloc, ldg, sizefactor, theta, cond, C, B, G, inv_B, loss_out,
                          dlogits, dpi, d_dtype, lddl, lddpi, dtheta, ws, status_dev, row_index, lgx1_rowsum, stream);
}

extern "C" int c2_nb_row_const(const float* x, int64_t ldx, const uint8_t* goodloc, int64_t ldg, int64_t B, int64_t G,
                               float* out, void* stream) {
    C2_CHECK(x && goodloc && out, "null pointer");
    C2_CHECK(ldx >= G && ldg >= G && B < (1ll << 31), "bad shape");
    if (B == 0) return 0;
    hipLaunchKernelGGL(nb_row_const_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, ldx, goodloc, ldg, G,
                       out);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_recon_mse_fwd_bwd(const float* logits, int64_t ldl, const float* x, int64_t ldx,
                                    const uint8_t* goodloc, int64_t ldg, int64_t B, int64_t G, float inv_B,
                                    float* loss_out, void* dlogits, int dl_dtype, int64_t lddl, float* ws,
                                    void* stream) {
    C2_CHECK(logits && x && goodloc && loss_out && dlogits && ws, "null pointer");
    C2_CHECK(dl_dtype == C2_F32 || dl_dtype == C2_BF16, "bad dtype");
    C2_CHECK(ldl >= G && ldx >= G && ldg >= G && lddl >= G, "leading dimension too small");
    if (B == 0 || G == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((B * G + 255) / 256, 2048));
    if (dl_dtype == C2_BF16)
        hipLaunchKernelGGL((mse_fwd_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, logits, ldl, x, ldx, goodloc, ldg, B,
                           G, inv_B, (bf16_t*)dlogits, lddl, ws);
    else
        hipLaunchKernelGGL((mse_fwd_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, logits, ldl, x, ldx, goodloc, ldg, B,
                           G, inv_B, (float*)dlogits, lddl, ws);
    launch_finalize_sum(ws, grid, loss_out, st);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_classifier_loss_fwd_bwd(const float* u, int64_t ldu, int64_t* classes, int64_t B, int P,
                                          float betaclass, float* loss_out, void* du, int du_dtype, int64_t lddu,
                                          float* ws, void* stream) {
    C2_CHECK(loss_out && ws, "null pointer");
    C2_CHECK(du_dtype == C2_F32 || du_dtype == C2_BF16, "bad dtype");
    if (B == 0 || P == 0) return 0;  // zero-width classifier: loss 0 (celligner2.py:350-356 with P = 0)
    C2_CHECK(u && classes && du && ldu >= P && lddu >= P, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((B + 3) / 4, 2048));
    if (du_dtype == C2_BF16)
        hipLaunchKernelGGL((classifier_loss_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, u, ldu, classes, B, P, betaclass,
                           (bf16_t*)du, lddu, ws);
    else
        hipLaunchKernelGGL((classifier_loss_kernel<float>), dim3(grid), dim3(256), 0, st, u, ldu, classes, B, P, betaclass,
                           (float*)du, lddu, ws);
    launch_finalize_sum(ws, grid, loss_out, st);
    C2_LAUNCH_CHECK();
    return 0;
}

// ---- decoder output activation of the inference path (Decoder.forward, modules.py:467-482, as reconstructLatent returns
// it: celligner2.py:411-420): Softmax(dim=-1) of the NB / ZINB mean head, ReLU of the MSE head.  One workgroup per row;
// the row is read twice (max + sum of exponentials in one online pass, then the normalised write) -- the second read of a
// 76 KB row comes from L2.
namespace c2 {
__global__ void __launch_bounds__(256) decoder_output_kernel(const float* __restrict__ logits, int64_t ldl, int64_t B,
                                                             int64_t G, int mode, float* __restrict__ out, int64_t ldo) {
    __shared__ float red[32];
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const float* row = logits + b * ldl;
        float* o = out + b * ldo;
        if (mode == 1) {  // ReLU
            for (int64_t g = threadIdx.x; g < G; g += 256) o[g] = fmaxf(row[g], 0.f);
            continue;
        }
        float mx = -INFINITY;
        for (int64_t g = threadIdx.x; g < G; g += 256) mx = fmaxf(mx, row[g]);
        mx = block_max<256>(mx, red);
        float s = 0.f;
        for (int64_t g = threadIdx.x; g < G; g += 256) s += __expf(row[g] - mx);
        s = block_sum<256>(s, red);
        const float inv = 1.0f / s;
        for (int64_t g = threadIdx.x; g < G; g += 256) o[g] = __expf(row[g] - mx) * inv;
        __syncthreads();  // red is reused by the next row
    }
}
}  // namespace c2

extern "C" int c2_decoder_output(const float* logits, int64_t ldl, int64_t B, int64_t G, int mode, float* out, int64_t ldo,
                                 void* stream) {
    C2_CHECK(mode == 0 || mode == 1, "mode: 0 = softmax over the genes, 1 = ReLU");
    C2_CHECK(B >= 0 && G >= 0, "negative size");
    if (B == 0 || G == 0) return 0;
    C2_CHECK(logits && out && ldl >= G && ldo >= G, "bad arguments");
    const int grid = (int)std::min<int64_t>(B, 4096);
    hipLaunchKernelGGL(c2::decoder_output_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, ldl, B, G, mode, out,
                       ldo);
    C2_LAUNCH_CHECK();
    return 0;
}
