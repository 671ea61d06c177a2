// Reconstruction (NB / MSE) and classifier losses: forward and analytic backward fused in one pass.
//
// NB kernel layout: one workgroup walks whole rows (cells) in four passes, re-reading the row from L1/L2 so that
// HBM sees each byte once; thread t owns the same gene columns in every row, so the per-gene dispersion gradient
// accumulates in an LDS array across the workgroup's rows and leaves the kernel as one partial row per workgroup
// (summed by a second tiny kernel: deterministic, no atomics).  HBM traffic per cell = logits 4G + x 4G +
// goodloc 1G read, dlogits (2|4)G written: the algorithmic minimum.
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
__global__ void nb_table_kernel(const float* __restrict__ theta, int C, int64_t G, float* __restrict__ tab) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= G) return;
    const float t0 = theta[g * C + 0];
    const float t1 = C > 1 ? theta[g * C + 1] : 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float t = p == 0 ? t0 : (p == 1 ? t1 : t0 + t1);
        const float th = expf(t);
        const float lte = logf(th + 1e-8f);
        tab[(p * 3 + 0) * G + g] = th;
        tab[(p * 3 + 1) * G + g] = th * lte - lgamma_pos(th);
        tab[(p * 3 + 2) * G + g] = lte + th / (th + 1e-8f) - digamma_pos(th);
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
        const float* tth = a.tab + (int64_t)(pat * 3 + 0) * G;
        const float* tk1 = a.tab + (int64_t)(pat * 3 + 1) * G;
        const float* tk2 = a.tab + (int64_t)(pat * 3 + 2) * G;
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

// Fast path (16-byte aligned rows, G % 4 == 0): the row's logits, counts and mask live in registers, every load of
// a pass is in flight at once, and while chunk i of pass 4 is being evaluated the NEXT row's chunk i is already
// being fetched into the registers chunk i has just released -- HBM latency hides under the log-likelihood maths.
template <typename TD, int NT, int NV, bool HAS_LGX>
__global__ __launch_bounds__(NT) void nb_rowreg_kernel(NbArgs a) {
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
    f32x4 ev[NV], xv[NV], pg[NV];
    unsigned mb[NV];
    const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int i, int64_t row) {
        const int64_t c = 4ll * (tid + NT * i);
        if (c < G) {
            const int64_t srow = a.ridx ? a.ridx[row] : row;
            ev[i] = *reinterpret_cast<const f32x4*>(a.logits + row * a.ldl + c);
            xv[i] = *reinterpret_cast<const f32x4*>(a.x + srow * a.ldx + c);
            mb[i] = *reinterpret_cast<const unsigned*>(a.goodloc + srow * a.ldg + c);
        } else {
            ev[i] = ninf; xv[i] = zero4; mb[i] = 0u;
        }
    };
    int64_t pref_row = -1;
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
        if (pref_row != row) {
#pragma unroll
            for (int i = 0; i < NV; ++i) fetch(i, row);
        }
        const float sf = a.sizefactor[row];
        const float* tth = a.tab + (int64_t)(pat * 3 + 0) * G;
        const float* tk1 = a.tab + (int64_t)(pat * 3 + 1) * G;
        const float* tk2 = a.tab + (int64_t)(pat * 3 + 2) * G;
        const float scale = -a.inv_B;
        const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) mx = fmaxf(fmaxf(mx, fmaxf(ev[i][0], ev[i][1])), fmaxf(ev[i][2], ev[i][3]));
        mx = block_max<NT>(mx, red);
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ev[i][k] = fexp(ev[i][k] - mx);
                sm += ev[i][k];
            }
        sm = block_sum<NT>(sm, red);
        const float inv_sum = 1.0f / sm;
        float S = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int64_t c = 4ll * (tid + NT * i);
            const f32x4 t0 = c < G ? *reinterpret_cast<const f32x4*>(tth + c) : one4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float mu = sf * ev[i][k] * inv_sum;
                const float zt = xv[i][k] + t0[k];
                const float v = xv[i][k] * mu * frcp(mu + eps) - zt * mu * frcp(t0[k] + mu + eps);
                pg[i][k] = ((mb[i] >> (8 * k)) & 0xffu) ? scale * v : 0.f;  // p * dloss/dp, kept for pass 4
                S += pg[i][k];
            }
        }
        S = block_sum<NT>(S, red);
        // next row of this workgroup (for the prefetch)
        const int64_t nrow = row + gridDim.x;
        TD* drow = reinterpret_cast<TD*>(a.dlogits) + row * a.lddl;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int64_t c0 = 4ll * (tid + NT * i);
            if (c0 < G) {
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(tth + c0);
                const f32x4 t1 = *reinterpret_cast<const f32x4*>(tk1 + c0);
                const f32x4 t2 = *reinterpret_cast<const f32x4*>(tk2 + c0);
                float d[4], dt[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool m = (mb[i] >> (8 * k)) & 0xffu;
                    const float p = ev[i][k] * inv_sum;
                    const float mu = sf * p;
                    const float th = t0[k], x = xv[i][k];
                    const float tm = th + mu + eps;
                    const float ltm = flog(tm);
                    const float zt = x + th;
                    float lg, dg;
                    lgam_digam_fast(zt, lg, dg);
                    float ll = t1[k] - zt * ltm + x * flog(mu + eps) + lg;
                    if (!HAS_LGX) ll -= lgamma_fast(x + 1.0f);
                    const float dth = t2[k] - ltm - zt * frcp(tm) + dg;
                    d[k] = pg[i][k] - p * S;
                    dt[k] = m ? scale * th * dth : 0.f;
                    loss_acc += m ? ll : 0.f;
                }
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
            }
            if (nrow < a.B) fetch(i, nrow);  // registers of chunk i are free: start the next row's loads
        }
        if (HAS_LGX && tid == 0) loss_acc -= a.lgx[row];
        pref_row = nrow < a.B ? nrow : -1;
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
                                                        int ngroups, int64_t G, float* __restrict__ part2 /*[S][3][G]*/) {
    const int64_t g = blockIdx.x * 256ll + threadIdx.x;
    const int sl = blockIdx.y;
    const int per = (ngroups + kDthSlices - 1) / kDthSlices;
    const int g0 = sl * per, g1 = min(ngroups, g0 + per);
    if (g >= G) return;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float s = 0.f;
        for (int grp = g0; grp < g1; ++grp)
            if (part_flag[grp * 3 + p]) s += part_dt[((int64_t)grp * 3 + p) * G + g];
        part2[((int64_t)sl * 3 + p) * G + g] = s;
    }
}
__global__ __launch_bounds__(256) void nb_dtheta_stage2(const float* __restrict__ part2, int64_t G, int C,
                                                        float* __restrict__ dtheta) {
    const int64_t g = blockIdx.x * 256ll + threadIdx.x;
    if (g >= G) return;
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll 8
        for (int sl = 0; sl < kDthSlices; ++sl) s[p] += part2[((int64_t)sl * 3 + p) * G + g];
    for (int j = 0; j < C; ++j) {
        float v = 0.f;
        if (j == 0) v = s[0] + s[2];
        if (j == 1) v = s[1] + s[2];
        dtheta[g * C + j] = v;
    }
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
// (the row is re-read from L2): this is the option-matrix path, not the benchmarked one.
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
        const float* tth = a.tab + (int64_t)(pat * 3 + 0) * G;
        const float* tk1 = a.tab + (int64_t)(pat * 3 + 1) * G;
        const float* tk2 = a.tab + (int64_t)(pat * 3 + 2) * G;
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
                ll = tk1[c] - zt * ltm + x * logf(mu + eps) + lgamma_pos(zt) - lgamma_pos(x + 1.0f) - pv - sp;
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
    }
    if (cur_pat >= 0) flush(cur_pat);
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
    // table 9G + partial dtheta ng*3*G + flags ng*3 (as floats) + loss partials ng
    return (size_t)(9 * G + ng * 3 * G + ng * 3 + ng + 64 + (int64_t)kDthSlices * 3 * G);
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
        auto kf = nb_rowreg_kernel<TD, NT, NV, false>;                                                         \
        auto kl = nb_rowreg_kernel<TD, NT, NV, true>;                                                          \
        if (lds > 48 * 1024) {                                                                                \
            (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
            (void)hipFuncSetAttribute((const void*)kl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        }                                                                                                     \
        if (a.lgx) hipLaunchKernelGGL(kl, dim3(ngroups), dim3(NT), lds, st, a);                                \
        else hipLaunchKernelGGL(kf, dim3(ngroups), dim3(NT), lds, st, a);                                      \
        return 0;                                                                                             \
    } while (0)
    if (vec) {
        const int64_t G = a.G;
        if (G <= 1024) NB_F(256, 1);
        if (G <= 2048) NB_F(256, 2);
        if (G <= 4096) NB_F(256, 4);
        if (G <= 8192) NB_F(1024, 2);
        if (G <= 12288) NB_F(1024, 3);
        if (G <= 16384) NB_F(1024, 4);
        if (G <= 20480) NB_F(1024, 5);
        if (G <= 24576) NB_F(1024, 6);
    }
    if (a.G <= 4096) NB_L(256);
    else NB_L(1024);
#undef NB_L
#undef NB_F
    return 0;
}

extern "C" int c2_recon_nb_fwd_bwd(const float* logits, int64_t ldl, const float* x, int64_t ldx,
                                   const uint8_t* goodloc, int64_t ldg, const float* sizefactor, const float* theta,
                                   const int64_t* cond, int C, int64_t B, int64_t G, float inv_B, float* loss_out,
                                   void* dlogits, int dl_dtype, int64_t lddl, float* dtheta, float* ws,
                                   int32_t* status_dev, const float* lgx1_rowsum, const int64_t* row_index,
                                   void* stream) {
    C2_CHECK(logits && x && goodloc && sizefactor && theta && cond && loss_out && dlogits && dtheta && ws && status_dev,
             "null pointer");
    C2_CHECK(dl_dtype == C2_F32 || dl_dtype == C2_BF16, "bad dtype");
    C2_CHECK(C >= 1, "NB needs at least one condition column (theta is [G,C])");
    C2_CHECK(ldl >= G && ldx >= G && ldg >= G && lddl >= G, "leading dimension too small");
    C2_CHECK(G <= 40000, "G > 40000 genes is not supported by the NB kernel (its LDS accumulator holds G floats)");
    if (B == 0 || G == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int ngroups = (int)std::min<int64_t>(B, kNbMaxGroups);
    float* tab = ws;
    float* part_dt = tab + 9 * G;
    int* part_flag = reinterpret_cast<int*>(part_dt + (int64_t)ngroups * 3 * G);
    float* part_loss = reinterpret_cast<float*>(part_flag + ngroups * 3);
    hipLaunchKernelGGL(nb_table_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, theta, C, G, tab);
    NbArgs a{logits, x, goodloc, sizefactor, cond, tab, dlogits, part_dt, part_flag, part_loss, status_dev,
             lgx1_rowsum, row_index, ldl, ldx, ldg, lddl, B, G, C, inv_B};
    const bool vec = (G % 4 == 0) && (ldl % 4 == 0) && (ldx % 4 == 0) && (ldg % 4 == 0) &&
                     (lddl % 4 == 0) && aligned16(logits) && aligned16(x) && aligned16(dlogits) &&
                     (((uintptr_t)goodloc & 3) == 0) && aligned16(ws);
    int rc = dl_dtype == C2_BF16 ? nb_launch<bf16_t>(a, ngroups, vec, st) : nb_launch<float>(a, ngroups, vec, st);
    C2_CHECK(rc == 0, "unsupported G");
    float* part2 = part_loss + ngroups + 32;
    hipLaunchKernelGGL(nb_dtheta_stage1, dim3((unsigned)((G + 255) / 256), kDthSlices), dim3(256), 0, st, part_dt, part_flag,
                       ngroups, G, part2);
    hipLaunchKernelGGL(nb_dtheta_stage2, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, part2, G, C, dtheta);
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
    C2_CHECK(logits && pi && x && goodloc && sizefactor && theta && cond && loss_out && dlogits && dpi && dtheta && ws &&
                 status_dev, "null pointer");
    C2_CHECK(d_dtype == C2_F32 || d_dtype == C2_BF16, "bad dtype");
    C2_CHECK(C >= 1, "ZINB needs at least one condition column (theta is [G,C])");
    C2_CHECK(ldl >= G && ldpi >= G && ldx >= G && ldg >= G && lddl >= G && lddpi >= G, "leading dimension too small");
    C2_CHECK(G <= 40000, "G > 40000 genes is not supported (the LDS accumulator holds G floats)");
    if (B == 0 || G == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int ngroups = (int)std::min<int64_t>(B, kNbMaxGroups);
    float* tab = ws;
    float* part_dt = tab + 9 * G;
    int* part_flag = reinterpret_cast<int*>(part_dt + (int64_t)ngroups * 3 * G);
    float* part_loss = reinterpret_cast<float*>(part_flag + ngroups * 3);
    hipLaunchKernelGGL(nb_table_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, theta, C, G, tab);
    NbArgs a{logits, x, goodloc, sizefactor, cond, tab, dlogits, part_dt, part_flag, part_loss, status_dev,
             nullptr, row_index, ldl, ldx, ldg, lddl, B, G, C, inv_B};
    ZinbExtra z{pi, dpi, ldpi, lddpi};
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
    float* part2 = part_loss + ngroups + 32;
    hipLaunchKernelGGL(nb_dtheta_stage1, dim3((unsigned)((G + 255) / 256), kDthSlices), dim3(256), 0, st, part_dt, part_flag,
                       ngroups, G, part2);
    hipLaunchKernelGGL(nb_dtheta_stage2, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, part2, G, C, dtheta);
    launch_finalize_sum(part_loss, ngroups, loss_out, st);
    C2_LAUNCH_CHECK();
    return 0;
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
