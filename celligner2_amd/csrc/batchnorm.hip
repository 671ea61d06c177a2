// BatchNorm1d form of the layer tail (the reference's use_bn=True: celligner2/model/modules.py:298-302, 405-413,
// 445-449, 557-566, 596-600 -- nn.BatchNorm1d(affine=True), eps 1e-5, momentum 0.1).  The statistics run over the
// batch, so the tail is cut where a multi-GPU run exchanges them:
//   forward   c2_bn_stats  (h = sum of slabs + bias + condition columns; per-column mean and M2 of the local rows)
//             [all-gather of the [2,H] partials between ranks]
//             c2_bn_fwd    (combine partials, running-stat update, normalise, affine, ReLU, dropout)
//   backward  c2_bn_bwd_reduce (dy through dropout/ReLU; column sums of dy and dy*yhat = dbeta, dgamma)
//             [all-reduce of the [2,H] sums]
//             c2_bn_bwd    (dh)
// Column statistics are combined with Chan's parallel update (mean, M2, count), never as E[x^2] - E[x]^2.
#include <algorithm>

#include "common.h"

namespace c2 {

constexpr int kBN_RB = 32;  // rows per statistics block

__global__ __launch_bounds__(256) void bn_pre_kernel(
    const float* __restrict__ Y, int64_t ldy, int nslab, int64_t slab_stride, const float* __restrict__ bias,
    const int64_t* __restrict__ cond, int C, const float* __restrict__ Wc, int64_t ldwc, int64_t B, int H,
    float* __restrict__ h_out, float* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    const int64_t r0 = (int64_t)blockIdx.y * kBN_RB;
    const int nr = (int)std::min<int64_t>(kBN_RB, B - r0);
    const float bc = bias ? bias[c] : 0.f;
    float v[kBN_RB];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kBN_RB; ++i) {
        float h = 0.f;
        if (i < nr) {
            const int64_t r = r0 + i;
            for (int s = 0; s < nslab; ++s) h += Y[s * slab_stride + r * ldy + c];
            h += bc;
            if (cond) {
                for (int j = 0; j < C; ++j) {
                    const float cj = (float)cond[r * C + j];
                    if (cj != 0.f) h += cj * Wc[(int64_t)c * ldwc + j];
                }
            }
            h_out[r * (int64_t)H + c] = h;
            sum += h;
        }
        v[i] = h;
    }
    const float mean = sum / (float)nr;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < kBN_RB; ++i) {
        const float d = v[i] - mean;
        m2 += (i < nr) ? d * d : 0.f;
    }
    part[((int64_t)blockIdx.y * 2) * H + c] = mean;
    part[((int64_t)blockIdx.y * 2 + 1) * H + c] = m2;
}

// Chan's update over n partials (mean, M2) laid out [n][2][H]; partial i covers rows [i*rpp, min((i+1)*rpp, total)).
__device__ __forceinline__ void bn_combine(const float* __restrict__ part, int n, int64_t rpp, int64_t total, int H,
                                           int c, float& mean, float& m2) {
    float cnt = 0.f;
    mean = 0.f;
    m2 = 0.f;
    for (int i = 0; i < n; ++i) {
        const float ni = (float)std::min<int64_t>(rpp, total - (int64_t)i * rpp);
        const float mi = part[((int64_t)i * 2) * H + c];
        const float qi = part[((int64_t)i * 2 + 1) * H + c];
        const float tot = cnt + ni;
        const float delta = mi - mean;
        mean += delta * (ni / tot);
        m2 += qi + delta * delta * (cnt * ni / tot);
        cnt = tot;
    }
}

__global__ void bn_combine_kernel(const float* __restrict__ part, int n, int64_t rpp, int64_t total, int H,
                                  float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    float mean, m2;
    bn_combine(part, n, rpp, total, H, c, mean, m2);
    out[c] = mean;
    out[H + c] = m2;
}

__global__ void bn_finalize_kernel(const float* __restrict__ parts, int nparts, int64_t rpp, int64_t total, int H,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   int64_t* __restrict__ nbt, float momentum, int update, float eps,
                                   float* __restrict__ stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    float mean, rstd;
    if (nparts > 0) {
        float m2;
        bn_combine(parts, nparts, rpp, total, H, c, mean, m2);
        rstd = 1.0f / sqrtf(m2 / (float)total + eps);
        if (update) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / (float)std::max<int64_t>(total - 1, 1));
            if (c == 0 && nbt) nbt[0] += 1;
        }
    } else {
        mean = running_mean[c];
        rstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    stats[c] = mean;
    stats[H + c] = rstd;
}

__device__ __forceinline__ void bn_store4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void bn_store4(bf16_t* p, const f32x4& v) {
    u32x2 pk;
    pk[0] = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    pk[1] = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    *reinterpret_cast<u32x2*>(p) = pk;
}
__device__ __forceinline__ void bn_planes4(bf16_t* hi_p, int64_t plane, const f32x4& v) {
    u32x2 ph, pl;
    bf16_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = f2bf(v[e]);
        l[e] = f2bf(v[e] - bf2f(h[e]));
    }
    ph[0] = (unsigned)h[0] | ((unsigned)h[1] << 16);
    ph[1] = (unsigned)h[2] | ((unsigned)h[3] << 16);
    pl[0] = (unsigned)l[0] | ((unsigned)l[1] << 16);
    pl[1] = (unsigned)l[2] | ((unsigned)l[3] << 16);
    *reinterpret_cast<u32x2*>(hi_p) = ph;
    *reinterpret_cast<u32x2*>(hi_p + plane) = pl;
}

// normalise + affine + ReLU + dropout; h is overwritten by yhat.  V = 4: H % 4 == 0 and 16-byte aligned rows.
template <typename TO, int V>
__global__ __launch_bounds__(256) void bn_fwd_apply_kernel(float* __restrict__ h, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const uint8_t* __restrict__ mask, float keep_scale,
                                                           int64_t B, int H, TO* __restrict__ act, int64_t ldact,
                                                           bf16_t* __restrict__ planes) {
    const int hv = H / V;
    const int64_t n = B * (int64_t)hv;
    const int64_t plane = B * (int64_t)H;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / hv;
        const int c = (int)(i - row * hv) * V;
        float* hp = h + row * (int64_t)H + c;
        if (V == 4) {
            const f32x4 hh = *reinterpret_cast<const f32x4*>(hp);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(stats + c);
            const f32x4 rs = *reinterpret_cast<const f32x4*>(stats + H + c);
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 yh, r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                yh[e] = (hh[e] - mu[e]) * rs[e];
                r[e] = fmaxf(fmaf(yh[e], g[e], b[e]), 0.f);
            }
            *reinterpret_cast<f32x4*>(hp) = yh;
            if (mask) {
                const unsigned mb = *reinterpret_cast<const unsigned*>(mask + row * (int64_t)H + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = ((mb >> (8 * e)) & 0xffu) ? r[e] * keep_scale : 0.f;
            }
            bn_store4(act + row * ldact + c, r);
            if (planes) bn_planes4(planes + row * (int64_t)H + c, plane, r);
        } else {
            const float yh = (*hp - stats[c]) * stats[H + c];
            float r = fmaxf(fmaf(yh, gamma[c], beta[c]), 0.f);
            *hp = yh;
            if (mask) r = mask[row * (int64_t)H + c] ? r * keep_scale : 0.f;
            st_from_f32<TO>(act + row * ldact + c, r);
        }
    }
}

// backward, stage 1: dy = (sum of slabs + extra) through dropout and ReLU; per-block column sums of dy and dy*yhat
__global__ __launch_bounds__(256) void bn_bwd_pre_kernel(
    const float* __restrict__ dA, int64_t ldda, int nslab, int64_t slab_stride, const float* __restrict__ extra,
    int64_t ldextra, const float* __restrict__ extra_scale_dev, const uint8_t* __restrict__ mask, float keep_scale,
    const float* __restrict__ yhat, const float* __restrict__ gamma, const float* __restrict__ beta, int64_t B, int H,
    float* __restrict__ dy_out, float* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    const int64_t r0 = (int64_t)blockIdx.y * kBN_RB;
    const int nr = (int)std::min<int64_t>(kBN_RB, B - r0);
    const float es = (extra && extra_scale_dev) ? *extra_scale_dev : 1.f;
    const float g = gamma[c], b = beta[c];
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < nr; ++i) {
        const int64_t r = r0 + i;
        float d = 0.f;
        for (int s = 0; s < nslab; ++s) d += dA[s * slab_stride + r * ldda + c];
        if (extra) d += es * extra[r * ldextra + c];
        if (mask) d = mask[r * (int64_t)H + c] ? d * keep_scale : 0.f;
        const float yh = yhat[r * (int64_t)H + c];
        d = fmaf(yh, g, b) > 0.f ? d : 0.f;
        dy_out[r * (int64_t)H + c] = d;
        s1 += d;
        s2 += d * yh;
    }
    part[((int64_t)blockIdx.y * 2) * H + c] = s1;
    part[((int64_t)blockIdx.y * 2 + 1) * H + c] = s2;
}

__global__ void bn_bwd_sum_kernel(const float* __restrict__ part, int n, int H, float* __restrict__ sums,
                                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < n; ++i) {
        s1 += part[((int64_t)i * 2) * H + c];
        s2 += part[((int64_t)i * 2 + 1) * H + c];
    }
    sums[c] = s1;
    sums[H + c] = s2;
    if (dbeta) dbeta[c] = s1;
    if (dgamma) dgamma[c] = s2;
}

template <typename TO, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy,
                                                           const float* __restrict__ yhat,
                                                           const float* __restrict__ sums, float inv_n,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ stats, int64_t B, int H,
                                                           TO* __restrict__ dh, int64_t lddh,
                                                           bf16_t* __restrict__ planes) {
    const int hv = H / V;
    const int64_t n = B * (int64_t)hv;
    const int64_t plane = B * (int64_t)H;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / hv;
        const int c = (int)(i - row * hv) * V;
        const int64_t o = row * (int64_t)H + c;
        if (V == 4) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
            const f32x4 yh = *reinterpret_cast<const f32x4*>(yhat + o);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + c);
            const f32x4 s2 = *reinterpret_cast<const f32x4*>(sums + H + c);
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 rs = *reinterpret_cast<const f32x4*>(stats + H + c);
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = g[e] * rs[e] * (d[e] - s1[e] * inv_n - yh[e] * (s2[e] * inv_n));
            bn_store4(dh + row * lddh + c, out);
            if (planes) bn_planes4(planes + o, plane, out);
        } else {
            const float out = gamma[c] * stats[H + c] * (dy[o] - sums[c] * inv_n - yhat[o] * (sums[H + c] * inv_n));
            st_from_f32<TO>(dh + row * lddh + c, out);
        }
    }
}

static inline int bn_blocks(int64_t B) { return (int)((B + kBN_RB - 1) / kBN_RB); }
static inline bool al(const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; }

}  // namespace c2

using namespace c2;

extern "C" size_t c2_bn_ws(int64_t B, int64_t H) { return (size_t)(bn_blocks(B) * 2 * H); }

extern "C" int c2_bn_stats(const float* Y, int64_t ldy, int nslab, int64_t slab_stride, const float* bias,
                           const int64_t* cond, int C, const float* Wc, int64_t ldwc, int64_t B, int64_t H,
                           float* h_out, float* part_out, float* ws, void* stream) {
    C2_CHECK(Y && h_out && part_out && ws, "null pointer");
    C2_CHECK(nslab >= 1 && ldy >= H && B >= 1 && H >= 1 && H < (1ll << 31), "bad shape");
    C2_CHECK(!(cond && C > 0) || Wc, "cond given without Wc");
    if (C == 0) cond = nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int nb = bn_blocks(B);
    hipLaunchKernelGGL(bn_pre_kernel, dim3((unsigned)((H + 255) / 256), (unsigned)nb), dim3(256), 0, st, Y, ldy, nslab,
                       slab_stride, bias, cond, C, Wc, ldwc, B, (int)H, h_out, ws);
    hipLaunchKernelGGL(bn_combine_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, st, ws, nb, (int64_t)kBN_RB,
                       B, (int)H, part_out);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_bn_fwd(float* h, const float* parts, int nparts, int64_t rows_per_part, int64_t total_rows,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         int64_t* num_batches_tracked, float momentum, int update_running, float eps,
                         const uint8_t* mask, float p_drop, int64_t B, int64_t H, void* act, int act_dtype,
                         int64_t ldact, float* stats_out, void* act_planes, void* stream) {
    C2_CHECK(h && gamma && beta && act && stats_out, "null pointer");
    C2_CHECK(act_dtype == C2_F32 || act_dtype == C2_BF16, "bad dtype");
    C2_CHECK(B >= 1 && H >= 1 && H < (1ll << 31) && ldact >= H, "bad shape");
    C2_CHECK(p_drop >= 0.f && p_drop < 1.f, "bad dropout rate");
    C2_CHECK(nparts >= 0 && (nparts == 0 || (parts && rows_per_part >= 1 && total_rows >= 1 &&
                                             (int64_t)(nparts - 1) * rows_per_part < total_rows &&
                                             (int64_t)nparts * rows_per_part >= total_rows)),
             "partials do not tile total_rows");
    C2_CHECK(nparts > 0 || (running_mean && running_var), "running statistics needed without batch statistics");
    C2_CHECK(!(update_running && nparts > 0) || (running_mean && running_var), "running statistics to update are NULL");
    C2_CHECK(!act_planes || act_dtype == C2_F32, "planes accompany an fp32 activation");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, st, parts, nparts,
                       rows_per_part, total_rows, (int)H, running_mean, running_var, num_batches_tracked, momentum,
                       update_running, eps, stats_out);
    const float ks = 1.0f / (1.0f - p_drop);
    const bool vec = (H % 4 == 0) && (ldact % 4 == 0) && al(h, 15) && al(stats_out, 15) && al(gamma, 15) &&
                     al(beta, 15) && al(act, act_dtype == C2_F32 ? 15 : 7) && (!mask || al(mask, 3)) &&
                     (!act_planes || al(act_planes, 7));
    C2_CHECK(vec || !act_planes, "act_planes needs the vector path (H % 4 == 0, aligned rows)");
    const int64_t n = B * (vec ? H / 4 : H);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096));
#define LAUNCH(TO, V)                                                                                               \
    hipLaunchKernelGGL((bn_fwd_apply_kernel<TO, V>), dim3(grid), dim3(256), 0, st, h, stats_out, gamma, beta, mask, ks, \
                       B, (int)H, (TO*)act, ldact, (bf16_t*)act_planes)
    if (act_dtype == C2_F32) {
        if (vec) LAUNCH(float, 4); else LAUNCH(float, 1);
    } else {
        if (vec) LAUNCH(bf16_t, 4); else LAUNCH(bf16_t, 1);
    }
#undef LAUNCH
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_bn_bwd_reduce(const float* dA, int64_t ldda, int nslab, int64_t slab_stride, const float* extra,
                                int64_t ldextra, const float* extra_scale_dev, const uint8_t* mask, float p_drop,
                                const float* yhat, const float* gamma, const float* beta, int64_t B, int64_t H,
                                float* dy_out, float* sums_out, float* dgamma, float* dbeta, float* ws, void* stream) {
    C2_CHECK(dA && yhat && gamma && beta && dy_out && sums_out && ws, "null pointer");
    C2_CHECK(nslab >= 1 && ldda >= H && B >= 1 && H >= 1 && H < (1ll << 31), "bad shape");
    C2_CHECK(!extra || ldextra >= H, "bad extra stride");
    C2_CHECK(p_drop >= 0.f && p_drop < 1.f, "bad dropout rate");
    hipStream_t st = (hipStream_t)stream;
    const int nb = bn_blocks(B);
    const float ks = 1.0f / (1.0f - p_drop);
    hipLaunchKernelGGL(bn_bwd_pre_kernel, dim3((unsigned)((H + 255) / 256), (unsigned)nb), dim3(256), 0, st, dA, ldda,
                       nslab, slab_stride, extra, ldextra, extra_scale_dev, mask, ks, yhat, gamma, beta, B, (int)H, dy_out,
                       ws);
    hipLaunchKernelGGL(bn_bwd_sum_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, st, ws, nb, (int)H, sums_out,
                       dgamma, dbeta);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_bn_bwd(const float* dy, const float* yhat, const float* sums, int64_t total_rows, int batch_stats,
                         const float* gamma, const float* stats, int64_t B, int64_t H, void* dh, int dh_dtype,
                         int64_t lddh, void* dh_planes, void* stream) {
    C2_CHECK(dy && yhat && sums && gamma && stats && dh, "null pointer");
    C2_CHECK(dh_dtype == C2_F32 || dh_dtype == C2_BF16, "bad dtype");
    C2_CHECK(B >= 1 && H >= 1 && H < (1ll << 31) && lddh >= H && total_rows >= B, "bad shape");
    C2_CHECK(!dh_planes || dh_dtype == C2_F32, "planes accompany an fp32 gradient");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (H % 4 == 0) && (lddh % 4 == 0) && al(dy, 15) && al(yhat, 15) && al(sums, 15) && al(gamma, 15) &&
                     al(stats, 15) && al(dh, dh_dtype == C2_F32 ? 15 : 7) && (!dh_planes || al(dh_planes, 7));
    C2_CHECK(vec || !dh_planes, "dh_planes needs the vector path (H % 4 == 0, aligned rows)");
    const float inv_n = batch_stats ? 1.0f / (float)total_rows : 0.f;
    const int64_t n = B * (vec ? H / 4 : H);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096));
#define LAUNCH(TO, V)                                                                                                  \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<TO, V>), dim3(grid), dim3(256), 0, st, dy, yhat, sums, inv_n, gamma, stats, B, \
                       (int)H, (TO*)dh, lddh, (bf16_t*)dh_planes)
    if (dh_dtype == C2_F32) {
        if (vec) LAUNCH(float, 4); else LAUNCH(float, 1);
    } else {
        if (vec) LAUNCH(bf16_t, 4); else LAUNCH(bf16_t, 1);
    }
#undef LAUNCH
    C2_LAUNCH_CHECK();
    return 0;
}
