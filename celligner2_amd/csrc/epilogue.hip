// HBM-bound row kernels around the GEMMs: input transform, layer tail (bias + condition columns +
// LayerNorm + ReLU + dropout) forward/backward, bias/condition-weight gradients, reparameterisation + KL,
// dataset row gather and small vector helpers.  One 64-lane wave owns one row: every row statistic is a
// wavefront-shuffle reduction, no LDS, no atomics (bit-stable results).
#include "common.h"

#include <mutex>

namespace c2 {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

__global__ void finalize_sum_kernel(const float* __restrict__ partials, int n, float* __restrict__ out) {
    __shared__ float red[32];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) *out += s;
}
// Deferred second stages (c2_defer_loss_sums): while on, the loss entry points of this thread do not launch their one-block
// finalize kernel; the (partials, n, out) triples wait here until c2_combine_losses[_dev] sums them in the same launch that
// forms the step's loss vector -- one launch instead of five or six (four finalizes, the zero-fill of the loss slots, the
// combine) in a step whose small kernels each cost a ~4.5 us graph node.
constexpr int kMaxPendingSums = 8;
struct PendingSums {
    const float* partials[kMaxPendingSums];
    float* out[kMaxPendingSums];
    int n[kMaxPendingSums];
    int count;
};
static thread_local bool g_defer_sums = false;
static thread_local PendingSums g_pending_sums = {};

// Suspend / resume deferral for the calling thread without touching what already waits (callers whose partials do not
// outlive the call: c2_mmd_fwd_bwd_static, whose scratch the next static call overwrites).  Returns the previous state.
bool set_defer_sums(bool on) {
    const bool was = g_defer_sums;
    g_defer_sums = on;
    return was;
}

int launch_finalize_sum(const float* partials, int n, float* out, hipStream_t stream) {
    if (g_defer_sums && g_pending_sums.count < kMaxPendingSums) {
        PendingSums& p = g_pending_sums;
        p.partials[p.count] = partials;
        p.out[p.count] = out;
        p.n[p.count] = n;
        ++p.count;
        return 0;
    }
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, stream, partials, n, out);
    return 0;
}

// ---------------------------------------------------------------------------------------------
template <typename TO, bool VEC4>
__global__ void prep_input_kernel(const float* __restrict__ x, int64_t rows, int64_t cols, int64_t ldx,
                                  TO* __restrict__ out, int64_t ldo, int apply_log, const float* __restrict__ center) {
    if (VEC4) {
        const int64_t cv = cols >> 2;
        const int64_t total = rows * cv;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
             i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t r = i / cv, c = (i - r * cv) << 2;
            f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
            if (apply_log) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = log1p_count(v[e]);
            }
            if (center) v -= *reinterpret_cast<const f32x4*>(center + c);
            TO* o = out + r * ldo + c;
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<f32x4*>(o) = v;
            } else {
                u32x2 pk;
                pk[0] = pack2<TO>(v[0], v[1]);
                pk[1] = pack2<TO>(v[2], v[3]);
                *reinterpret_cast<u32x2*>(o) = pk;
            }
        }
    } else {
        const int64_t total = rows * cols;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
             i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t r = i / cols, c = i - r * cols;
            float v = x[r * ldx + c];
            if (apply_log) v = log1p_count(v);
            if (center) v -= center[c];
            st_from_f32<TO>(out + r * ldo + c, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// layer tail forward: wave per row
template <typename TO, int EPL /*cached float4 chunks per lane, 0 = no cache*/>
__global__ __launch_bounds__(256) void epilogue_fwd_kernel(
    const float* __restrict__ Y, int64_t ldy, int nslab, int64_t slab_stride, const float* __restrict__ bias,
    const int64_t* __restrict__ cond, int C, const float* __restrict__ Wc, int64_t ldwc,
    const uint8_t* __restrict__ mask, float keep_scale, int use_ln, int64_t B, int H, TO* __restrict__ act,
    int64_t ldact, float* __restrict__ yhat, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int64_t* crow = cond ? cond + row * C : nullptr;

    auto hval = [&](int c) -> float {
        float h = 0.f;
        for (int s = 0; s < nslab; ++s) h += Y[s * slab_stride + row * ldy + c];
        if (bias) h += bias[c];
        if (crow) {
            for (int j = 0; j < C; ++j) {
                const float cj = (float)crow[j];
                if (cj != 0.f) h += cj * Wc[(int64_t)c * ldwc + j];
            }
        }
        return h;
    };

    float cache[EPL > 0 ? EPL * 4 : 1];
    float sum = 0.f;
    if (EPL > 0) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (lane + 64 * i) * 4 + e;
                const float h = (c < H) ? hval(c) : 0.f;
                cache[i * 4 + e] = h;
                sum += h;
            }
        }
    } else {
        for (int c = lane; c < H; c += 64) sum += hval(c);
    }
    float mean = 0.f, rstd = 1.f;
    if (use_ln) {
        mean = wave_sum(sum) / (float)H;
        float vs = 0.f;
        if (EPL > 0) {
#pragma unroll
            for (int i = 0; i < EPL; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = (lane + 64 * i) * 4 + e;
                    const float d = cache[i * 4 + e] - mean;
                    vs += (c < H) ? d * d : 0.f;
                }
        } else {
            for (int c = lane; c < H; c += 64) {
                const float d = hval(c) - mean;
                vs += d * d;
            }
        }
        const float var = wave_sum(vs) / (float)H;
        rstd = 1.0f / sqrtf(var + 1e-5f);
    }
    if (rstd_out && lane == 0) rstd_out[row] = rstd;

    auto emit = [&](int c, float h) {
        const float yh = (h - mean) * rstd;
        if (yhat) yhat[row * (int64_t)H + c] = yh;
        float r = fmaxf(yh, 0.f);
        if (mask) r = mask[row * (int64_t)H + c] ? r * keep_scale : 0.f;
        st_from_f32<TO>(act + row * ldact + c, r);
    };
    if (EPL > 0) {
#pragma unroll
        for (int i = 0; i < EPL; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (lane + 64 * i) * 4 + e;
                if (c < H) emit(c, cache[i * 4 + e]);
            }
    } else {
        for (int c = lane; c < H; c += 64) emit(c, hval(c));
    }
}

// layer tail backward: wave per row
template <typename TO, int EPL>
__global__ __launch_bounds__(256) void epilogue_bwd_kernel(
    const float* __restrict__ dA, int64_t ldda, int nslab, int64_t slab_stride, const float* __restrict__ extra,
    int64_t ldextra, const float* __restrict__ extra_scale_dev, const uint8_t* __restrict__ mask, float keep_scale,
    int use_ln, const float* __restrict__ yhat, const float* __restrict__ rstd_in, int64_t B, int H,
    TO* __restrict__ dh, int64_t lddh) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float es = (extra && extra_scale_dev) ? *extra_scale_dev : 1.f;

    auto dyval = [&](int c, float& yh) -> float {
        float d = 0.f;
        for (int s = 0; s < nslab; ++s) d += dA[s * slab_stride + row * ldda + c];
        if (extra) d += es * extra[row * ldextra + c];
        if (mask) d = mask[row * (int64_t)H + c] ? d * keep_scale : 0.f;
        yh = yhat[row * (int64_t)H + c];
        return yh > 0.f ? d : 0.f;
    };
    float cdy[EPL > 0 ? EPL * 4 : 1], cyh[EPL > 0 ? EPL * 4 : 1];
    float s1 = 0.f, s2 = 0.f;
    if (EPL > 0) {
#pragma unroll
        for (int i = 0; i < EPL; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (lane + 64 * i) * 4 + e;
                float yh = 0.f;
                const float d = (c < H) ? dyval(c, yh) : 0.f;
                cdy[i * 4 + e] = d;
                cyh[i * 4 + e] = yh;
                s1 += d;
                s2 += d * yh;
            }
    } else {
        for (int c = lane; c < H; c += 64) {
            float yh;
            const float d = dyval(c, yh);
            s1 += d;
            s2 += d * yh;
        }
    }
    float m1 = 0.f, m2 = 0.f, rstd = 1.f;
    if (use_ln) {
        m1 = wave_sum(s1) / (float)H;
        m2 = wave_sum(s2) / (float)H;
        rstd = rstd_in[row];
    }
    if (EPL > 0) {
#pragma unroll
        for (int i = 0; i < EPL; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (lane + 64 * i) * 4 + e;
                if (c < H) {
                    const float g = use_ln ? (cdy[i * 4 + e] - m1 - cyh[i * 4 + e] * m2) * rstd : cdy[i * 4 + e];
                    st_from_f32<TO>(dh + row * lddh + c, g);
                }
            }
    } else {
        for (int c = lane; c < H; c += 64) {
            float yh;
            const float d = dyval(c, yh);
            const float g = use_ln ? (d - m1 - yh * m2) * rstd : d;
            st_from_f32<TO>(dh + row * lddh + c, g);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Vector forms of the two layer-tail kernels (H % 4 == 0, 16-byte aligned rows, H <= 256 * EPL): every access is a
// 16-byte load/store (8 bytes for bf16 outputs, 4 for the mask), the row lives in registers between the passes.
// `planes` (optional, with fp32 output): the same values as two bf16 planes [2][B][H] (hi, lo = x - hi), the
// operand form of the narrow bf16x3 GEMMs (C2_BF16X3P), written here instead of by a separate split pass.
__device__ __forceinline__ void store_out4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store_out4(bf16_t* p, const f32x4& v) {
    u32x2 pk;
    pk[0] = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    pk[1] = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    *reinterpret_cast<u32x2*>(p) = pk;
}
__device__ __forceinline__ void store_planes4(bf16_t* hi_p, int64_t plane, const f32x4& v) {
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
    *reinterpret_cast<u32x2*>(hi_p) = ph;
    *reinterpret_cast<u32x2*>(hi_p + plane) = pl;
}

// Both kernels are written in PHASES -- every load of the row, the arithmetic, every store -- with no branch between
// the loads of a phase and none between the stores: columns past the row's end read column 0 and are zeroed by a select,
// slabs past the last one re-read slab 0 likewise.  Written per 16-byte chunk (load, use, store, next chunk) the
// compiler keeps each chunk's loads behind the branches of the one before it and opens every chunk with s_waitcnt
// vmcnt(0) -- which on this architecture also waits for the previous chunk's STORES to come back from memory: a row of
// 1024 columns cost its wave some sixteen serial round trips to memory (ISA of the round-2 form).
__device__ __forceinline__ f32x4 sel4(bool keep, const f32x4& v) {
    return f32x4{keep ? v[0] : 0.f, keep ? v[1] : 0.f, keep ? v[2] : 0.f, keep ? v[3] : 0.f};
}
// the row's sums over the slabs at the EPL chunks (valid columns cc[]): the branch on the slab count OUTSIDE the walk over
// the chunks, groups of four independent loads per chunk, slabs past the end re-read slab 0 and are zeroed
#ifndef C2_TAIL_NT
#define C2_TAIL_NT 1
#endif
#if C2_TAIL_NT
#define C2_TAIL_LD(p) __builtin_nontemporal_load(p)  /* the GEMM's slabs are read once, here */
#else
#define C2_TAIL_LD(p) (*(p))
#endif
template <int EPL>
__device__ __forceinline__ void slab_sums(f32x4 (&h)[EPL], const float* __restrict__ rowp, const int (&cc)[EPL], int nslab,
                                          int64_t slab_stride) {
    if (nslab == 1) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) h[i] = f32x4{0.f, 0.f, 0.f, 0.f} + *reinterpret_cast<const f32x4*>(rowp + cc[i]);
        return;
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) h[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < nslab; s0 += 4) {
        f32x4 t[EPL][4];
#pragma unroll
        for (int i = 0; i < EPL; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                const f32x4* src = reinterpret_cast<const f32x4*>(rowp + cc[i] + (s0 + u < nslab ? s0 + u : 0) * slab_stride);
                if constexpr (EPL == 4) t[i][u] = C2_TAIL_LD(src);  // (rows of 1024: the slabs are tens of MB; narrower layers' slabs
                else t[i][u] = *src;                                 //  are still in the caches, and the hint measured slower there)
            }
#pragma unroll
        for (int i = 0; i < EPL; ++i)
            h[i] += (sel4(s0 < nslab, t[i][0]) + sel4(s0 + 1 < nslab, t[i][1])) +
                    (sel4(s0 + 2 < nslab, t[i][2]) + sel4(s0 + 3 < nslab, t[i][3]));
    }
}
#define C2_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70) /* s_waitcnt vmcnt(0): as an instruction the compiler's own counter sees */

template <typename TO, int EPL>
__global__ __launch_bounds__(256) void epilogue_fwd_vec_kernel(
    const float* __restrict__ Y, int64_t ldy, int nslab, int64_t slab_stride, const float* __restrict__ bias,
    const int64_t* __restrict__ cond, int C, const float* __restrict__ Wc, int64_t ldwc,
    const uint8_t* __restrict__ mask, float keep_scale, int use_ln, int64_t B, int H, TO* __restrict__ act,
    int64_t ldact, float* __restrict__ yhat, float* __restrict__ rstd_out, bf16_t* __restrict__ planes, int64_t ldp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int64_t* crow = cond ? cond + row * C : nullptr;
    bool ok[EPL];
    int cc[EPL];  // the chunk's first column, or 0 where the chunk lies past the row's end (a valid address: H >= 4)
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int c = (lane + 64 * i) * 4;
        ok[i] = c < H;
        cc[i] = ok[i] ? c : 0;
    }
    // ---- loads
    f32x4 cache[EPL];
    unsigned mb[EPL];
    slab_sums<EPL>(cache, Y + row * ldy, cc, nslab, slab_stride);
    if (mask) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) mb[i] = *reinterpret_cast<const unsigned*>(mask + row * (int64_t)H + cc[i]);
    }
    if (bias) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) cache[i] += *reinterpret_cast<const f32x4*>(bias + cc[i]);
    }
    if (crow) {
        for (int j = 0; j < C; ++j) {
            const float cj = (float)crow[j];
            if (cj != 0.f) {
#pragma unroll
                for (int i = 0; i < EPL; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) cache[i][e] += cj * Wc[(int64_t)(cc[i] + e) * ldwc + j];
            }
        }
    }
    // ---- statistics
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        cache[i] = sel4(ok[i], cache[i]);
        sum += (cache[i][0] + cache[i][1]) + (cache[i][2] + cache[i][3]);
    }
    float mean = 0.f, rstd = 1.f;
    if (use_ln) {
        mean = wave_sum(sum) / (float)H;
        float vs = 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = cache[i][e] - mean;
                vs += ok[i] ? d * d : 0.f;
            }
        }
        const float var = wave_sum(vs) / (float)H;
        rstd = 1.0f / sqrtf(var + 1e-5f);
    }
    // ---- values
    f32x4 yh[EPL], r[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            yh[i][e] = (cache[i][e] - mean) * rstd;
            r[i][e] = fmaxf(yh[i][e], 0.f);
        }
        if (mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[i][e] = ((mb[i] >> (8 * e)) & 0xffu) ? r[i][e] * keep_scale : 0.f;
        }
    }
    // ---- stores
    C2_VMCNT0();
    if (rstd_out && lane == 0) rstd_out[row] = rstd;
    const int64_t plane = B * ldp;  // (planes: rows ldp elements apart, the low plane B rows on)
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        if (ok[i]) {
            if (yhat) *reinterpret_cast<f32x4*>(yhat + row * (int64_t)H + cc[i]) = yh[i];
            store_out4(act + row * ldact + cc[i], r[i]);
            if (planes) store_planes4(planes + row * ldp + cc[i], plane, r[i]);
        }
    }
}

template <typename TO, int EPL>
__global__ __launch_bounds__(256) void epilogue_bwd_vec_kernel(
    const float* __restrict__ dA, int64_t ldda, int nslab, int64_t slab_stride, const float* __restrict__ extra,
    int64_t ldextra, const float* __restrict__ extra_scale_dev, const uint8_t* __restrict__ mask, float keep_scale,
    int use_ln, const float* __restrict__ yhat, const float* __restrict__ rstd_in, int64_t B, int H,
    TO* __restrict__ dh, int64_t lddh, bf16_t* __restrict__ planes, int64_t ldp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    bool ok[EPL];
    int cc[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int c = (lane + 64 * i) * 4;
        ok[i] = c < H;
        cc[i] = ok[i] ? c : 0;
    }
    // ---- loads
    const float es = (extra && extra_scale_dev) ? *extra_scale_dev : 1.f;
    const float rstd = use_ln ? rstd_in[row] : 1.f;
    f32x4 cdy[EPL], cyh[EPL], ex[EPL];
    unsigned mb[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) cyh[i] = *reinterpret_cast<const f32x4*>(yhat + row * (int64_t)H + cc[i]);
    slab_sums<EPL>(cdy, dA + row * ldda, cc, nslab, slab_stride);
    if (extra) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) ex[i] = *reinterpret_cast<const f32x4*>(extra + row * ldextra + cc[i]);
    }
    if (mask) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) mb[i] = *reinterpret_cast<const unsigned*>(mask + row * (int64_t)H + cc[i]);
    }
    // ---- arithmetic
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        f32x4 d = cdy[i];
        if (extra) d += es * ex[i];
        if (mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = ((mb[i] >> (8 * e)) & 0xffu) ? d[e] * keep_scale : 0.f;
        }
        cyh[i] = sel4(ok[i], cyh[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d[e] = (ok[i] && cyh[i][e] > 0.f) ? d[e] : 0.f;
            s1 += d[e];
            s2 += d[e] * cyh[i][e];
        }
        cdy[i] = d;
    }
    float m1 = 0.f, m2 = 0.f;
    if (use_ln) {
        m1 = wave_sum(s1) / (float)H;
        m2 = wave_sum(s2) / (float)H;
    }
    f32x4 g[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) g[i][e] = use_ln ? (cdy[i][e] - m1 - cyh[i][e] * m2) * rstd : cdy[i][e];
    // ---- stores
    C2_VMCNT0();
    const int64_t plane = B * ldp;  // (planes: rows ldp elements apart, the low plane B rows on)
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        if (ok[i]) {
            store_out4(dh + row * lddh + cc[i], g[i]);
            if (planes) store_planes4(planes + row * ldp + cc[i], plane, g[i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// db / dWc: stage 1 -- block = 128 columns x RB rows: thread (cx, ry) sums 2 adjacent columns over every 8th row of
// the chunk in ONE pass (bias and up to 8 condition columns together), the 8 row groups are combined through LDS;
// stage 2 -- sum over row chunks (independent loads, unrolled).
constexpr int kBCG_RB = 128;
constexpr int kBCG_MAXC = 8;
template <typename TI>
__device__ __forceinline__ void ld2(const TI* p, bool two, float& v0, float& v1);
template <>
__device__ __forceinline__ void ld2<float>(const float* p, bool two, float& v0, float& v1) {
    v0 = p[0];
    v1 = two ? p[1] : 0.f;
}
template <>
__device__ __forceinline__ void ld2<bf16_t>(const bf16_t* p, bool two, float& v0, float& v1) {
    v0 = bf2f(p[0]);
    v1 = two ? bf2f(p[1]) : 0.f;
}
// DIRECT (short batches, B <= kBCG_DIRECT_B): one launch -- a block walks ALL rows of its 128 columns and writes alpha * sum
// straight to db / dWc; at the reference's default batch of 128 the second stage was a launch with nothing to combine.
constexpr int kBCG_DIRECT_B = 256;
// CXN: column pairs per block (64: 128 columns x 8 row groups; 16: 32 columns x 32 row groups -- the DIRECT form, whose few
// blocks each walk every row: at 128 rows x 256 columns two blocks took 13.7 us, a chain of 16 dependent row steps).
template <typename TI, int CC, bool DIRECT = false, int CXN = 64>
__global__ __launch_bounds__(512) void bias_cond_stage1(const TI* __restrict__ dh, int64_t lddh,
                                                        const int64_t* __restrict__ cond, int C, int c0, int64_t B, int H,
                                                        float* __restrict__ ws /*[nchunk][1+C][H]*/,
                                                        float* __restrict__ db = nullptr, float* __restrict__ dWc = nullptr,
                                                        int64_t lddwc = 0, const float* __restrict__ alpha_dev = nullptr) {
    constexpr int RYN = 512 / CXN, NCOL = 2 * CXN;
    __shared__ float red[RYN][(1 + CC)][NCOL];
    const int cx = threadIdx.x % CXN, ry = threadIdx.x / CXN;
    const int col = blockIdx.x * NCOL + 2 * cx;
    const int chunk = blockIdx.y;
    const int64_t b0 = DIRECT ? 0 : (int64_t)chunk * kBCG_RB, b1 = DIRECT ? B : min(B, b0 + kBCG_RB);
    float a0[2] = {0.f, 0.f};
    float acc[CC > 0 ? CC : 1][2];
#pragma unroll
    for (int j = 0; j < CC; ++j) acc[j][0] = acc[j][1] = 0.f;
    if (col < H) {
        const bool two = col + 1 < H;
        // four rows per trip, every load of the trip (the rows, their condition values -- unconditionally: a column past the
        // last re-reads the last and counts as 0) issued before the first use: row by row, with the condition load behind a
        // branch, each row was a serial round trip to memory
        auto row = [&](int64_t b, float& v0, float& v1, float (&cj)[CC > 0 ? CC : 1]) {
            ld2<TI>(dh + b * lddh + col, two, v0, v1);
#pragma unroll
            for (int j = 0; j < CC; ++j) {
                const float c = (float)cond[b * C + min(c0 + j, C - 1)];
                cj[j] = (c0 + j < C) ? c : 0.f;
            }
        };
        auto use = [&](float v0, float v1, const float (&cj)[CC > 0 ? CC : 1]) {
            a0[0] += v0;
            a0[1] += v1;
#pragma unroll
            for (int j = 0; j < CC; ++j) {
                acc[j][0] += v0 * cj[j];
                acc[j][1] += v1 * cj[j];
            }
        };
        int64_t b = b0 + ry;
        for (; b + 3 * RYN < b1; b += 4 * RYN) {
            float v0[4], v1[4], cj[4][CC > 0 ? CC : 1];
#pragma unroll
            for (int u = 0; u < 4; ++u) row(b + u * RYN, v0[u], v1[u], cj[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) use(v0[u], v1[u], cj[u]);
        }
        for (; b < b1; b += RYN) {
            float v0, v1, cj[CC > 0 ? CC : 1];
            row(b, v0, v1, cj);
            use(v0, v1, cj);
        }
    }
    red[ry][0][2 * cx] = a0[0];
    red[ry][0][2 * cx + 1] = a0[1];
#pragma unroll
    for (int j = 0; j < CC; ++j) {
        red[ry][1 + j][2 * cx] = acc[j][0];
        red[ry][1 + j][2 * cx + 1] = acc[j][1];
    }
    __syncthreads();
    float* out = ws + (int64_t)chunk * (1 + C) * H;
    for (int i = threadIdx.x; i < (1 + CC) * NCOL; i += 512) {
        const int k = i / NCOL, c = i % NCOL;
        const int gc = blockIdx.x * NCOL + c;
        if (gc >= H) continue;
        if (k == 0 && c0 != 0) continue;          // the bias row is written by the first condition group only
        if (k > 0 && c0 + k - 1 >= C) continue;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < RYN; ++r) s += red[r][k][c];
        if (DIRECT) {
            s *= alpha_dev ? *alpha_dev : 1.f;
            if (k == 0) {
                if (db) db[gc] = s;
            } else if (dWc) {
                dWc[(int64_t)gc * lddwc + (c0 + k - 1)] = s;
            }
        } else {
            out[(int64_t)(k == 0 ? 0 : c0 + k) * H + gc] = s;
        }
    }
}
// 16-byte loads: block = 64 column groups (8 bf16 / 4 fp32 columns each) x 4 row groups; a thread walks every 4th row of
// its chunk of `rb` rows with 8 independent loads in flight and sums the bias and up to CC condition columns.
template <typename TI, int CC>
__global__ __launch_bounds__(256) void bias_stage1_vec(const TI* __restrict__ dh, int64_t lddh,
                                                       const int64_t* __restrict__ cond, int C, int c0, int64_t B, int H,
                                                       int rb, float* __restrict__ ws /*[nchunk][1+C][H]*/) {
    constexpr int W = 16 / sizeof(TI);  // columns per thread
    __shared__ float red[4][1 + CC][64 * W];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + cx) * W;
    const int chunk = blockIdx.y;
    const int64_t b0 = (int64_t)chunk * rb, b1 = min(B, b0 + rb);
    float acc[1 + CC][W];
#pragma unroll
    for (int k = 0; k <= CC; ++k)
#pragma unroll
        for (int e = 0; e < W; ++e) acc[k][e] = 0.f;
    if (col < H) {
        // (the rows' condition values are handed in, fetched by the caller together with the rows: a load inside this
        // accumulation made each of the eight rows in flight wait for its own round trip to memory)
        auto add = [&](const u32x4& v, const float (&cj)[CC > 0 ? CC : 1]) {
            float x[W];
            if constexpr (sizeof(TI) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[2 * e] = __builtin_bit_cast(float, v[e] << 16);
                    x[2 * e + 1] = __builtin_bit_cast(float, v[e] & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned w = v[e];  // (bit_cast straight from the vector element reads element 0: copy first)
                    x[e] = __builtin_bit_cast(float, w);
                }
            }
#pragma unroll
            for (int e = 0; e < W; ++e) acc[0][e] += x[e];
#pragma unroll
            for (int j = 0; j < CC; ++j) {
#pragma unroll
                for (int e = 0; e < W; ++e) acc[1 + j][e] += x[e] * cj[j];
            }
        };
        auto conds = [&](int64_t b, float (&cj)[CC > 0 ? CC : 1]) {
#pragma unroll
            for (int j = 0; j < CC; ++j) {  // (no branch around the load: a column past the last re-reads the last, then 0)
                const float c = (float)cond[b * C + min(c0 + j, C - 1)];
                cj[j] = (c0 + j < C) ? c : 0.f;
            }
        };
        int64_t b = b0 + ry;
        for (; b + 28 < b1; b += 32) {
            u32x4 v[8];
            float cj[8][CC > 0 ? CC : 1];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const u32x4*>(dh + (b + 4 * u) * lddh + col);
#pragma unroll
            for (int u = 0; u < 8; ++u) conds(b + 4 * u, cj[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) add(v[u], cj[u]);
        }
        for (; b < b1; b += 4) {
            float cj[CC > 0 ? CC : 1];
            const u32x4 v = *reinterpret_cast<const u32x4*>(dh + b * lddh + col);
            conds(b, cj);
            add(v, cj);
        }
    }
#pragma unroll
    for (int k = 0; k <= CC; ++k)
#pragma unroll
        for (int e = 0; e < W; ++e) red[ry][k][cx * W + e] = acc[k][e];
    __syncthreads();
    float* out = ws + (int64_t)chunk * (1 + C) * H;
    for (int i = threadIdx.x; i < (1 + CC) * 64 * W; i += 256) {
        const int k = i / (64 * W), c = i - k * (64 * W);
        const int gc = blockIdx.x * 64 * W + c;
        if (gc >= H) continue;
        if (k == 0 && c0 != 0) continue;          // the bias row is written by the first condition group only
        if (k > 0 && c0 + k - 1 >= C) continue;
        out[(int64_t)(k == 0 ? 0 : c0 + k) * H + gc] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}
__global__ __launch_bounds__(256) void bias_cond_stage2(const float* __restrict__ ws, int nchunk, int C, int H,
                                                        float* __restrict__ db, float* __restrict__ dWc, int64_t lddwc,
                                                        const float* __restrict__ alpha_dev) {
    // 64 consecutive elements of the [(1+C)*H] result x 4 groups of row chunks, combined through LDS
    __shared__ float red[4][64];
    const int lx = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lx;  // over (1+C)*H
    const int64_t stride = (int64_t)(1 + C) * H;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < (1 + C) * H) {
        int c = gy;
        for (; c + 12 < nchunk; c += 16) {
            s0 += ws[(c + 0) * stride + i];
            s1 += ws[(c + 4) * stride + i];
            s2 += ws[(c + 8) * stride + i];
            s3 += ws[(c + 12) * stride + i];
        }
        for (; c < nchunk; c += 4) s0 += ws[c * stride + i];
    }
    red[gy][lx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (gy != 0 || i >= (1 + C) * H) return;
    float s = (red[0][lx] + red[1][lx]) + (red[2][lx] + red[3][lx]);
    const int k = i / H, col = i - k * H;
    const float alpha = alpha_dev ? *alpha_dev : 1.f;
    s *= alpha;
    if (k == 0) {
        if (db) db[col] = s;
    } else if (dWc) {
        dWc[(int64_t)col * lddwc + (k - 1)] = s;
    }
}

// ---------------------------------------------------------------------------------------------
template <typename TZ>
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                             const float* __restrict__ eps, int64_t B, int L,
                                                             int64_t ld, float inv_B, int use_own_kl,
                                                             float* __restrict__ z32, TZ* __restrict__ zt,
                                                             int64_t ldzt, float* __restrict__ partials) {
    __shared__ float red[32];
    const int64_t total = B * L;
    float acc = 0.f;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / L;
        const int l = (int)(i - b * L);
        const float m = mu[b * ld + l], v_l = lv[b * ld + l];
        const float var = expf(v_l) + 1e-4f;
        const float z = m + sqrtf(var) * eps[b * (int64_t)L + l];
        z32[b * (int64_t)L + l] = z;
        if (zt) st_from_f32<TZ>(zt + b * ldzt + l, z);
        if (use_own_kl)
            acc += -0.5f * (1.f + v_l - m * m - var);
        else
            acc += 0.5f * (var + m * m - 1.f - logf(var)) * inv_B;
    }
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

template <typename TO>
__global__ void reparam_kl_bwd_kernel(const float* __restrict__ dZ, int64_t lddz, int nslab, int64_t slab_stride,
                                      const float* __restrict__ extra, int64_t ldextra,
                                      const float* __restrict__ extra_scale_dev, const float* __restrict__ mu,
                                      const float* __restrict__ lv, const float* __restrict__ eps, int64_t B, int L,
                                      int64_t ld, float inv_B, int use_own_kl, const float* __restrict__ g_kl_dev,
                                      TO* __restrict__ dmu, TO* __restrict__ dlv, int64_t ldout) {
    const int64_t total = B * L;
    const float gk = g_kl_dev ? *g_kl_dev : 1.f;
    const float es = (extra && extra_scale_dev) ? *extra_scale_dev : 1.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / L;
        const int l = (int)(i - b * L);
        float dz = 0.f;
        for (int s = 0; s < nslab; ++s) dz += dZ[s * slab_stride + b * lddz + l];
        if (extra) dz += es * extra[b * ldextra + l];
        const float m = mu[b * ld + l], v_l = lv[b * ld + l];
        const float ex = expf(v_l);
        const float var = ex + 1e-4f;
        float dm_kl, dl_kl;
        if (use_own_kl) {
            dm_kl = m;
            dl_kl = -0.5f * (1.f - ex);
        } else {
            dm_kl = m * inv_B;
            dl_kl = 0.5f * (1.f - 1.f / var) * ex * inv_B;
        }
        const float gm = dz + gk * dm_kl;
        const float gl = dz * eps[b * (int64_t)L + l] * ex / (2.f * sqrtf(var)) + gk * dl_kl;
        st_from_f32<TO>(dmu + b * ldout + l, gm);
        st_from_f32<TO>(dlv + b * ldout + l, gl);
    }
}

// ---------------------------------------------------------------------------------------------
template <typename TX>
__global__ __launch_bounds__(256) void gather_rows_kernel(
    const float* __restrict__ X, const uint8_t* __restrict__ goodloc, const float* __restrict__ sizefactor,
    const int64_t* __restrict__ cond, const int64_t* __restrict__ classes, int64_t G, int C, int P,
    const int64_t* __restrict__ idx, float* __restrict__ x_out, uint8_t* __restrict__ goodloc_out,
    float* __restrict__ sf_out, int64_t* __restrict__ cond_out, int64_t* __restrict__ classes_out,
    TX* __restrict__ xin_out, int64_t ldxin, int apply_log, int vec_ok, int vec_xin,
    const float* __restrict__ rowconst, float* __restrict__ rowconst_out, int32_t* __restrict__ idx32_out,
    int64_t n, int64_t n_pad, const float* __restrict__ center, bf16_t* __restrict__ xin2_out, int64_t ldxin2,
    int vec_xin2, int64_t ldX, int64_t ldg, int64_t ldxo, int64_t ldgo, int64_t Gw) {
    // ldX / ldg / ldxo / ldgo: row pitches of X, goodloc, x_out, goodloc_out.  Gw: the columns the 16-byte walk covers -- G, or
    // G rounded up to 4 when every row involved is at least that long (the sources' pad columns, zeros, are then copied).
    // blockIdx.y: the part of the row's columns this block moves (short batches: one block per row would leave most of the
    // chip idle); the per-row scalars travel with part 0
    const int64_t r = blockIdx.x;
    const int64_t src = idx[r];
    const int part = blockIdx.y, nparts = gridDim.y;
    // The row's scalars (size factor, likelihood constant, condition and class rows) are REQUESTED here, ahead of the
    // row's columns, and stored at the end with the 32-bit index: as load - store pairs one after the other they were
    // three dependent trips to random rows in HBM and two store round trips in between (stores count in vmcnt), which
    // is all this kernel does when the consumers read the resident rows themselves.  A pointer that is absent reads
    // element 0 of `idx` instead (any valid address), and the value is dropped.
    const bool lead = part == 0;
    const bool t0 = threadIdx.x == 0;
    const int tj = (int)threadIdx.x;
    const float sfv = (sizefactor ? sizefactor : reinterpret_cast<const float*>(idx))[(lead && t0 && sizefactor) ? src : 0];
    const float rcv = (rowconst ? rowconst : reinterpret_cast<const float*>(idx))[(lead && t0 && rowconst) ? src : 0];
    const int64_t cdv = (cond ? cond : idx)[(lead && cond && tj < C) ? src * C + tj : 0];
    const int64_t clv = (classes ? classes : idx)[(lead && classes && tj < P) ? src * P + tj : 0];
    const float* xs = X + src * ldX;
    float* xd = x_out ? x_out + r * ldxo : nullptr;
    const bool copy_good = goodloc && goodloc_out;
    if (!xd && !xin_out && !xin2_out && !copy_good) {
        // nothing per gene: the consumers read the resident rows themselves
    } else if (vec_ok) {
        const int64_t gv = Gw >> 2;
        // The chunk's four centre values are fetched by the CALLER, with the chunk (a load in here, after the stores of
        // the chunk before, made every chunk wait for those stores: they count in vmcnt on this architecture).  center has
        // G entries: the chunk that holds the pad columns is handed the last four, cv = center[G-4 .. G), and picks its own.
        auto center4 = [&](int64_t c) {
            f32x4 cv = {0.f, 0.f, 0.f, 0.f};
            if (center) {  // (G >= 4 on this path; 4-byte alignment is all a global 16-byte load needs)
                const int64_t at = min(c * 4, G - 4);
                cv = f32x4{center[at], center[at + 1], center[at + 2], center[at + 3]};
            }
            return cv;
        };
        auto emit = [&](int64_t c, const f32x4& v, unsigned gm, const f32x4& cv) {
            if (xd) reinterpret_cast<f32x4*>(xd)[c] = v;
            if (xin_out || xin2_out) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = apply_log ? log1p_count(v[e]) : v[e];
                if (center) {
                    if (c * 4 + 4 <= G) {
                        t -= cv;
                    } else {  // the chunk that holds the pad columns: column c*4+e < G is entry e + s of the last four
                        const int sft = (int)(c * 4 - (G - 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = e + sft;
                            const float ce = k == 0 ? cv[0] : (k == 1 ? cv[1] : (k == 2 ? cv[2] : cv[3]));
                            if (c * 4 + e < G) t[e] -= ce;
                        }
                    }
                }
                if (xin_out) {
                    TX* dst = xin_out + r * ldxin + c * 4;
                    if constexpr (sizeof(TX) == 2) {
                        if (vec_xin) {
                            u32x2 pk;
                            pk[0] = pack2<TX>(t[0], t[1]);
                            pk[1] = pack2<TX>(t[2], t[3]);
                            *reinterpret_cast<u32x2*>(dst) = pk;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) st_from_f32<TX>(dst + e, t[e]);
                        }
                    } else {
                        if (vec_xin) {
                            *reinterpret_cast<f32x4*>(dst) = t;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) st_from_f32<TX>(dst + e, t[e]);
                        }
                    }
                }
                if (xin2_out) {
                    bf16_t* dst2 = xin2_out + r * ldxin2 + c * 4;
                    if (vec_xin2) {
                        u32x2 pk;
                        pk[0] = pack2<bf16_t>(t[0], t[1]);
                        pk[1] = pack2<bf16_t>(t[2], t[3]);
                        *reinterpret_cast<u32x2*>(dst2) = pk;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst2[e] = f2bf(t[e]);
                    }
                }
            }
            if (copy_good) reinterpret_cast<unsigned int*>(goodloc_out + r * ldgo)[c] = gm;
        };
        const unsigned int* gs = copy_good ? reinterpret_cast<const unsigned int*>(goodloc + src * ldg) : nullptr;
        const int64_t per = ((gv + nparts - 1) / nparts + 255) / 256 * 256;  // chunks per part, whole thread rounds
        const int64_t gv0 = gv;
        const int64_t cbeg = min(gv0, (int64_t)part * per), cend = min(gv0, cbeg + per);
        int64_t c = cbeg + threadIdx.x;
        for (; c + 768 < cend; c += 1024) {  // four independent 16-byte loads in flight per thread
            f32x4 v[4], cv[4];
            unsigned gm[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const f32x4*>(xs)[c + 256 * u];
            if (copy_good) {
#pragma unroll
                for (int u = 0; u < 4; ++u) gm[u] = gs[c + 256 * u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) cv[u] = center4(c + 256 * u);
            C2_VMCNT0();  // (one wait the compiler's counter sees, ahead of the stores' branches: none inside them)
#pragma unroll
            for (int u = 0; u < 4; ++u) emit(c + 256 * u, v[u], gm[u], cv[u]);
        }
        for (; c < cend; c += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(xs)[c];
            const unsigned gm = copy_good ? gs[c] : 0u;
            const f32x4 cv = center4(c);
            C2_VMCNT0();
            emit(c, v, gm, cv);
        }
    } else {
        const int64_t per = ((G + nparts - 1) / nparts + 255) / 256 * 256;
        const int64_t cbeg = min(G, (int64_t)part * per), cend = min(G, cbeg + per);
        for (int64_t c = cbeg + threadIdx.x; c < cend; c += 256) {
            const float v = xs[c];
            if (xd) xd[c] = v;
            const float t = (apply_log ? log1p_count(v) : v) - (center ? center[c] : 0.f);
            if (xin_out) st_from_f32<TX>(xin_out + r * ldxin + c, t);
            if (xin2_out) xin2_out[r * ldxin2 + c] = f2bf(t);
            if (copy_good) goodloc_out[r * ldgo + c] = goodloc[src * ldg + c];
        }
    }
    if (part != 0) return;
    if (t0 && sizefactor) sf_out[r] = sfv;
    if (t0 && rowconst) rowconst_out[r] = rcv;
    if (cond && tj < C) cond_out[r * C + tj] = cdv;
    if (classes && tj < P) classes_out[r * P + tj] = clv;
    if (idx32_out && t0) {
        idx32_out[r] = (int32_t)src;
        if (r == n - 1)
            for (int64_t q = n; q < n_pad; ++q) idx32_out[q] = (int32_t)src;  // padding: any valid row
    }
    if (cond)  // (more than 256 condition / class columns: the rest the plain way)
        for (int j = tj + 256; j < C; j += 256) cond_out[r * C + j] = cond[src * C + j];
    if (classes)
        for (int j = tj + 256; j < P; j += 256) classes_out[r * P + j] = classes[src * P + j];
}

// Dropout keep-masks for every layer of a step in one launch: Philox-4x32-10 keyed by (seed, offset), 16 mask
// bytes per counter value.  mask byte = 1 with probability 1-p.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ void dropout_mask_kernel(uint8_t* __restrict__ out, int64_t n, unsigned thresh8, unsigned long long seed,
                                    unsigned long long offset) {
    const int64_t nvec = (n + 15) / 16;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned r[4];
        const unsigned long long ctr = offset + (unsigned long long)i;
        philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0x3c6ef372u, 0xdaa66d2bu, (unsigned)seed,
                      (unsigned)(seed >> 32), r);
        u32x4 pk;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (((r[w] >> (8 * b)) & 0xffu) >= thresh8 ? 1u : 0u) << (8 * b);
            pk[w] = v;
        }
        if ((i + 1) * 16 <= n) {
            *reinterpret_cast<u32x4*>(out + i * 16) = pk;
        } else {
            for (int64_t b = i * 16; b < n; ++b) out[b] = (uint8_t)((pk[(b & 15) >> 2] >> (8 * (b & 3))) & 0xffu);
        }
    }
}

__global__ void axpy_kernel(const float* __restrict__ x, const float* __restrict__ a_dev, float* __restrict__ y,
                            int64_t n) {
    const float a = a_dev ? *a_dev : 1.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] += a * x[i];
}
// out6 = (l0, l1, l2, l3, sum_i w_i*l_i, l0+l1+l2): the step's loss vector in one launch
__global__ void combine_losses_kernel(const float* __restrict__ l, float w0, float w1, float w2, float w3,
                                      float* __restrict__ out) {
    if (threadIdx.x == 0) {
        const float a = l[0], b = l[1], c = l[2], d = l[3];
        out[0] = a; out[1] = b; out[2] = c; out[3] = d;
        out[4] = ((w0 * a + w1 * b) + w2 * c) + w3 * d;
        out[5] = (a + b) + c;
    }
}
__global__ void scale4_kernel(const float* __restrict__ g, float w0, float w1, float w2, float w3,
                              float* __restrict__ out) {
    if (threadIdx.x == 0) {
        const float s = g ? g[0] : 1.f;
        out[0] = w0 * s; out[1] = w1 * s; out[2] = w2 * s; out[3] = w3 * s;
    }
}
// The same two with the four weights read from device memory (a step replayed as a HIP graph bakes every by-value
// argument in: the KL weight changes per epoch, so it has to live on the device); accum6 (may be NULL) += out6, the
// running sums an epoch's mean losses are made of without a host read per step.
__global__ void combine_losses_dev_kernel(const float* __restrict__ l, const float* __restrict__ w,
                                          float* __restrict__ out, double* __restrict__ accum) {
    if (threadIdx.x == 0) {
        const float a = l[0], b = l[1], c = l[2], d = l[3];
        const float t = ((w[0] * a + w[1] * b) + w[2] * c) + w[3] * d, u = (a + b) + c;
        out[0] = a; out[1] = b; out[2] = c; out[3] = d; out[4] = t; out[5] = u;
        if (accum) {
            accum[0] += a; accum[1] += b; accum[2] += c; accum[3] += d; accum[4] += t; accum[5] += u;
        }
    }
}
// The pending second stages and the loss vector in one launch (256 threads).  Every sum is formed exactly as
// finalize_sum_kernel forms it (same strides, same reduction tree): bit-identical to the separate launches.  The four loss
// slots l[0..3] are taken as ZERO on entry (what they held is ignored) and receive the sums addressed to them; a sum
// addressed elsewhere is added to its target.  w_dev NULL: the weights by value.
__global__ __launch_bounds__(256) void finalize_combine_kernel(PendingSums p, float* __restrict__ l,
                                                               const float* __restrict__ w_dev, float w0, float w1, float w2,
                                                               float w3, float* __restrict__ out, double* __restrict__ accum) {
    __shared__ float red[32];
    float slot[4] = {0.f, 0.f, 0.f, 0.f};
    // The first two elements a thread adds of EVERY pending sum are requested before the first sum is formed (elements past
    // a sum's end re-read its element 0 and are not added): the sums are a few hundred partials each, so this block is a
    // chain of latencies, and one trip to memory per sum was most of it.  Same additions in the same order.
    float v0[kMaxPendingSums], v1[kMaxPendingSums];
#pragma unroll
    for (int k = 0; k < kMaxPendingSums; ++k) {
        const bool live = k < p.count && p.n[k] > 0;  // (uniform)
        const float* src = live ? p.partials[k] : l;   // (no sum in this slot: the loss vector's first element, dropped)
        const int n = live ? p.n[k] : 0;
        v0[k] = src[(int)threadIdx.x < n ? threadIdx.x : 0];
        v1[k] = src[(int)threadIdx.x + 256 < n ? threadIdx.x + 256 : 0];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kMaxPendingSums; ++k) {
        if (k >= p.count) break;
        float s = 0.f;
        if ((int)threadIdx.x < p.n[k]) s += v0[k];
        if ((int)threadIdx.x + 256 < p.n[k]) s += v1[k];
        for (int i = threadIdx.x + 512; i < p.n[k]; i += 256) s += p.partials[k][i];
        s = block_sum<256>(s, red);
        const ptrdiff_t d = p.out[k] - l;
        if (d >= 0 && d < 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (d == j) slot[j] += s;
        } else if (threadIdx.x == 0) {
            *p.out[k] += s;
        }
    }
    if (threadIdx.x == 0) {
        if (w_dev) { w0 = w_dev[0]; w1 = w_dev[1]; w2 = w_dev[2]; w3 = w_dev[3]; }
        const float a = slot[0], b = slot[1], c = slot[2], d = slot[3];
        const float t = ((w0 * a + w1 * b) + w2 * c) + w3 * d, u = (a + b) + c;
        l[0] = a; l[1] = b; l[2] = c; l[3] = d;
        out[0] = a; out[1] = b; out[2] = c; out[3] = d; out[4] = t; out[5] = u;
        if (accum) {
            accum[0] += a; accum[1] += b; accum[2] += c; accum[3] += d; accum[4] += t; accum[5] += u;
        }
    }
}
__global__ void scale4_dev_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ out) {
    if (threadIdx.x < 4) out[threadIdx.x] = w[threadIdx.x] * (g ? g[0] : 1.f);
}
// The backward seed and the two vectors it scales first, in one launch: out4 = w * g (as scale4_dev_kernel), y1 = out4[0] * x1,
// y2 = out4[0] * x2 -- the output layer's bias gradient and the dispersion gradient, which the NB loss kernel left as
// unit-upstream sums.  Every block forms out4[0] from w and g itself; block 0 writes out4.
__global__ void scale4_dev_scale2_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ out,
                                         const float* __restrict__ x1, float* __restrict__ y1, int64_t n1,
                                         const float* __restrict__ x2, float* __restrict__ y2, int64_t n2) {
    const float up = g ? g[0] : 1.f;
    const float s = w[0] * up;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n1 + n2; i += (int64_t)gridDim.x * 256) {
        if (i < n1) y1[i] = s * x1[i];
        else y2[i - n1] = s * x2[i - n1];
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) out[threadIdx.x] = w[threadIdx.x] * up;
}
// Dropout masks with the Philox counter offset in device memory; a second one-thread launch advances it after every
// block of the mask kernel has read it (stream order).
__global__ void dropout_mask_dev_kernel(uint8_t* __restrict__ out, int64_t n, unsigned thresh8, unsigned long long seed,
                                        const unsigned long long* __restrict__ offset_dev, unsigned long long lead) {
    const unsigned long long offset = *offset_dev + lead;
    const int64_t nvec = (n + 15) / 16;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned r[4];
        const unsigned long long ctr = offset + (unsigned long long)i;
        philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0x3c6ef372u, 0xdaa66d2bu, (unsigned)seed,
                      (unsigned)(seed >> 32), r);
        u32x4 pk;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (((r[w] >> (8 * b)) & 0xffu) >= thresh8 ? 1u : 0u) << (8 * b);
            pk[w] = v;
        }
        if ((i + 1) * 16 <= n) {
            *reinterpret_cast<u32x4*>(out + i * 16) = pk;
        } else {
            for (int64_t b = i * 16; b < n; ++b) out[b] = (uint8_t)((pk[(b & 15) >> 2] >> (8 * (b & 3))) & 0xffu);
        }
    }
}
__global__ void advance_counter_kernel(unsigned long long* ctr, unsigned long long by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *ctr += by;
}
// All the noise of one step from one launch: the dropout keep-masks exactly as dropout_mask_dev_kernel draws them, the
// reparameterisation's standard normals (Box-Muller on the Philox words of a second key, four normals per counter value;
// element e of the GLOBAL [ranks x B, L] matrix comes from counter value offset + e / 4, word pair e % 4, so a rank's rows
// are the rows one process would have drawn), and the counter's advance: the block that is last to take a ticket -- every
// block has read the offset by then -- adds `advance` and clears the ticket count.  state = {offset, tickets taken}.
__global__ __launch_bounds__(256) void noise_dev_kernel(uint8_t* __restrict__ masks, int64_t n_mask, unsigned thresh8,
                                                        float* __restrict__ eps, int64_t n_eps, unsigned long long eps_first,
                                                        unsigned long long seed, unsigned long long* __restrict__ state,
                                                        unsigned long long lead, unsigned long long advance) {
    const unsigned long long base = state[0];
    // Every wave of the block holds the counter before the block takes its ticket; the block with the last ticket knows
    // that all blocks have read it.  (The ticket is taken first and looked at last: its latency hides behind the draws.  A
    // fence-and-count at the END of each block was measured at 249 us for 4096 blocks, against 6 us for the draws.)
    unsigned long long ticket = 0;
    if (advance != 0) {
        __syncthreads();
        if (threadIdx.x == 0) ticket = atomicAdd(&state[1], 1ull);
    }
    const unsigned long long offset = base + lead;
    const int64_t nvec = (n_mask + 15) / 16;
    const int64_t tid = blockIdx.x * 256ll + threadIdx.x, nthr = (int64_t)gridDim.x * 256;
    for (int64_t i = tid; i < nvec; i += nthr) {
        unsigned r[4];
        const unsigned long long ctr = offset + (unsigned long long)i;
        philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0x3c6ef372u, 0xdaa66d2bu, (unsigned)seed,
                      (unsigned)(seed >> 32), r);
        u32x4 pk;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (((r[w] >> (8 * b)) & 0xffu) >= thresh8 ? 1u : 0u) << (8 * b);
            pk[w] = v;
        }
        if ((i + 1) * 16 <= n_mask) {
            *reinterpret_cast<u32x4*>(masks + i * 16) = pk;
        } else {
            for (int64_t b = i * 16; b < n_mask; ++b) masks[b] = (uint8_t)((pk[(b & 15) >> 2] >> (8 * (b & 3))) & 0xffu);
        }
    }
    // normals: one thread per counter value of the rank's element range [eps_first, eps_first + n_eps)
    if (n_eps > 0) {
        const unsigned long long v0 = eps_first >> 2, v1 = (eps_first + (unsigned long long)n_eps + 3) >> 2;
        for (unsigned long long v = v0 + (unsigned long long)tid; v < v1; v += (unsigned long long)nthr) {
            unsigned r[4];
            const unsigned long long ctr = base + v;
            philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0x9e3779b9u, 0x7f4a7c15u, (unsigned)seed ^ 0x85ebca6bu,
                          (unsigned)(seed >> 32) ^ 0xc2b2ae35u, r);
            float z[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float u1 = ((float)(r[2 * h] >> 8) + 1.f) * (1.f / 16777216.f);  // (0, 1]
                const float u2 = (float)(r[2 * h + 1] >> 8) * (1.f / 16777216.f);      // [0, 1)
                const float rad = sqrtf(-2.f * logf(u1));
                float sn, cs;
                sincospif(2.f * u2, &sn, &cs);
                z[2 * h] = rad * cs;
                z[2 * h + 1] = rad * sn;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long e = 4 * v + k;
                if (e >= eps_first && e < eps_first + (unsigned long long)n_eps) eps[e - eps_first] = z[k];
            }
        }
    }
    if (advance != 0 && threadIdx.x == 0 && ticket == (unsigned long long)gridDim.x - 1) {
        state[0] = base + advance;
        state[1] = 0ull;
    }
}

__global__ void scale_kernel(const float* __restrict__ x, const float* __restrict__ a_dev, float* __restrict__ y,
                             int64_t n) {
    const float a = *a_dev;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = a * x[i];
}
__global__ void cast_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = f2bf(s[i]);
}

__global__ void cast_f16_kernel(const float* __restrict__ s, f16_t* __restrict__ d, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = (f16_t)s[i];
}

static inline int grid_for(int64_t n, int per_block = 256, int cap = 4096) {
    int64_t b = (n + per_block - 1) / per_block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, cap));
}

// ---------------------------------------------------------------------------------------------
// Centred encoder input (bf16 mode).  The first layer computes h = t(x) W^T + b with t(x) = log(1 + x) (or x) rounded to
// bf16.  Expression matrices are dominated by a per-gene level shared by all cells, so the rounding error (2^-9 of
// |t(x)|, and 2^-9 of |W| times |t(x)|) is large against the cell-to-cell differences the gradients are made of.
// With any fixed vector c [G]:  h = (t(x) - c) W^T + (b + W c),  dW = dh^T (t(x) - c) + (sum_b dh_b) (x) c  -- exact
// algebra, and the bf16 operand is now the small centred part; W c and the rank-one term are formed in fp32.
// c = the per-gene mean of t(x) over the dataset (or over the batch, for batches handed in as tensors).
//
// column sums of t(x) over the rows in DOUBLE, two deterministic stages (no atomics).  fp32 values summed in double are
// exact for any batch this library sees, so the sum -- and the mean rounded to fp32 from it -- does not depend on how the
// rows are cut into splits, or into ranks: N ranks that all-reduce their double sums centre on the same bits as one
// process on the whole batch.
__global__ __launch_bounds__(256) void col_sum_stage1(const float* __restrict__ x, int64_t ldx, int64_t rows, int64_t cols,
                                                      int apply_log, int nsplit, double* __restrict__ part) {
    const int64_t c = blockIdx.x * 256ll + threadIdx.x;
    if (c >= cols) return;
    const int s = blockIdx.y;
    const int64_t per = (rows + nsplit - 1) / nsplit;
    const int64_t r0 = s * per, r1 = r0 + per < rows ? r0 + per : rows;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {
        float v0 = x[r * ldx + c], v1 = x[(r + 1) * ldx + c], v2 = x[(r + 2) * ldx + c], v3 = x[(r + 3) * ldx + c];
        if (apply_log) { v0 = log1p_count(v0); v1 = log1p_count(v1); v2 = log1p_count(v2); v3 = log1p_count(v3); }
        a0 += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
    }
    for (; r < r1; ++r) a0 += (double)(apply_log ? log1p_count(x[r * ldx + c]) : x[r * ldx + c]);
    part[(int64_t)s * cols + c] = (a0 + a1) + (a2 + a3);
}
__global__ __launch_bounds__(256) void col_sum_stage2(const double* __restrict__ part, int nsplit, int64_t cols,
                                                      double* __restrict__ sums) {
    const int64_t c = blockIdx.x * 256ll + threadIdx.x;
    if (c >= cols) return;
    double a = 0.0;
    for (int s = 0; s < nsplit; ++s) a += part[(int64_t)s * cols + c];
    sums[c] = a;
}
__global__ __launch_bounds__(256) void col_mean_finish(const double* __restrict__ sums, int64_t cols, double inv_rows,
                                                       float* __restrict__ out) {
    const int64_t c = blockIdx.x * 256ll + threadIdx.x;
    if (c < cols) out[c] = (float)(sums[c] * inv_rows);
}

// out[n] = bias[n] + sum_k W[n][k] c[k]: one workgroup per output row, fp32, 16-byte loads where the row allows
__global__ __launch_bounds__(256) void gemv_bias_kernel(const float* __restrict__ W, int64_t ldw, int64_t N, int64_t K,
                                                        const float* __restrict__ c, const float* __restrict__ bias,
                                                        float* __restrict__ out, int vec) {
    __shared__ float red[32];
    const int64_t n = blockIdx.x;
    const float* w = W + n * ldw;
    float a = 0.f;
    if (vec) {
        const int64_t kv = K >> 2;
        const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
        const f32x4* c4 = reinterpret_cast<const f32x4*>(c);
        // ten 16-byte loads of the row (and ten of c) in flight per thread and batch, every one issued before the first is
        // used; chunks past the row's end re-read its last chunk and count as zero (a load behind a branch is waited for
        // inside the branch: the remainder loop of the four-loads form was three more dependent trips to memory per row)
        constexpr int NB = 10;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        for (int64_t i0 = 0; i0 < kv; i0 += 256 * NB) {
            f32x4 wv[NB], cv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int64_t i = min(i0 + 256 * u + (int64_t)threadIdx.x, kv - 1);
                wv[u] = __builtin_nontemporal_load(w4 + i);
                cv[u] = c4[i];
            }
            __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler sinks each load to its use: one pair in flight)
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const float keep = (i0 + 256 * u + (int64_t)threadIdx.x < kv) ? 1.f : 0.f;
                if (u & 1) a1 += keep * (wv[u] * cv[u]);
                else a0 += keep * (wv[u] * cv[u]);
            }
        }
        const f32x4 acc = a0 + a1;
        a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    } else {
        // rows that start on 4 bytes only (a gene count that is not a multiple of 4): eight independent loads in flight
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int64_t k = threadIdx.x;
        for (; k + 7 * 256 < K; k += 8 * 256) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = __builtin_nontemporal_load(w + k + 256 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += wv[u] * c[k + 256 * u];
        }
        for (; k < K; k += 256) s[0] += w[k] * c[k];
        a = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
    a = block_sum<256>(a, red);
    if (threadIdx.x == 0) out[n] = a + (bias ? bias[n] : 0.f);
}

// G[n][k] += r[n] * c[k]
__global__ __launch_bounds__(256) void rank1_add_kernel(float* __restrict__ g, int64_t ldg, int64_t N, int64_t K,
                                                        const float* __restrict__ r, const float* __restrict__ c, int vec) {
    if (vec) {
        const int64_t kv = K >> 2, total = N * kv;
        for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int64_t n = i / kv, k = (i - n * kv) << 2;
            f32x4* p = reinterpret_cast<f32x4*>(g + n * ldg + k);
            *p = *p + r[n] * *reinterpret_cast<const f32x4*>(c + k);
        }
    } else {
        const int64_t total = N * K;
        for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int64_t n = i / K, k = i - n * K;
            g[n * ldg + k] += r[n] * c[k];
        }
    }
}

}  // namespace c2

using namespace c2;

extern "C" const char* c2_last_error(void) { return g_err.c_str(); }
extern "C" int c2_version(void) { return 1; }
extern "C" size_t c2_reduce_ws(void) { return kReduceWsFloats; }

extern "C" int c2_prep_input(const float* x, int64_t rows, int64_t cols, int64_t ldx, void* out, int out_dtype,
                             int64_t ldo, int apply_log, const float* center, void* stream) {
    C2_CHECK(x && out, "null pointer");
    C2_CHECK(out_dtype == C2_F32 || out_dtype == C2_BF16 || out_dtype == C2_F16, "bad dtype");
    C2_CHECK(ldx >= cols && ldo >= cols && rows >= 0 && cols >= 0, "bad shape");
    if (rows == 0 || cols == 0) return 0;
    const bool vec = (cols % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && aligned16(x) && aligned16(out) &&
                     (!center || aligned16(center));
    const int grid = grid_for(rows * cols / (vec ? 4 : 1));
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(TO, V) \
    hipLaunchKernelGGL((prep_input_kernel<TO, V>), dim3(grid), dim3(256), 0, st, x, rows, cols, ldx, (TO*)out, ldo, apply_log, center)
    if (out_dtype == C2_F32) {
        if (vec) LAUNCH(float, true); else LAUNCH(float, false);
    } else if (out_dtype == C2_F16) {
        if (vec) LAUNCH(f16_t, true); else LAUNCH(f16_t, false);
    } else {
        if (vec) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false);
    }
#undef LAUNCH
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_epilogue_fwd(const float* Y, int64_t ldy, int nslab, int64_t slab_stride, const float* bias,
                               const int64_t* cond, int C, const float* Wc, int64_t ldwc, const uint8_t* mask,
                               float p_drop, int use_ln, int64_t B, int64_t H, void* act, int act_dtype,
                               int64_t ldact, float* yhat, float* rstd, void* act_planes, int64_t ld_planes, void* stream) {
    C2_CHECK(Y && act, "null pointer");
    const int64_t ldp = ld_planes ? ld_planes : H;
    C2_CHECK(!act_planes || ldp >= H, "ld_planes shorter than a row");
    C2_CHECK(!act_planes || act_dtype == C2_F32, "planes accompany an fp32 activation");
    C2_CHECK(act_dtype == C2_F32 || act_dtype == C2_BF16, "bad dtype");
    C2_CHECK(nslab >= 1 && ldy >= H && ldact >= H && H < (1ll << 31), "bad shape");
    C2_CHECK(!(cond && C > 0) || Wc, "cond given without Wc");
    C2_CHECK(p_drop >= 0.f && p_drop < 1.f, "bad dropout rate");
    if (B == 0 || H == 0) return 0;
    if (C == 0) cond = nullptr;
    const float ks = 1.0f / (1.0f - p_drop);
    dim3 grid((unsigned)((B + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (H % 4 == 0) && H <= 1024 && (ldy % 4 == 0) && (slab_stride % 4 == 0) && (ldact % 4 == 0) &&
                     aligned16(Y) && (!bias || aligned16(bias)) && (!yhat || aligned16(yhat)) &&
                     ((reinterpret_cast<uintptr_t>(act) & (act_dtype == C2_F32 ? 15 : 7)) == 0) &&
                     (!mask || (reinterpret_cast<uintptr_t>(mask) & 3) == 0) &&
                     (!act_planes || ((reinterpret_cast<uintptr_t>(act_planes) & 7) == 0 && ldp % 4 == 0 && (B * ldp) % 4 == 0));
    C2_CHECK(vec || !act_planes, "act_planes needs the vector path (H % 4 == 0, H <= 1024, aligned rows)");
#define LAUNCHV(TO, E)                                                                                                 \
    hipLaunchKernelGGL((epilogue_fwd_vec_kernel<TO, E>), grid, dim3(256), 0, st, Y, ldy, nslab, slab_stride, bias, cond, \
                       C, Wc, ldwc, mask, ks, use_ln, B, (int)H, (TO*)act, ldact, yhat, rstd, (bf16_t*)act_planes, ldp)
    if (vec) {
        if (act_dtype == C2_F32) {
            if (H <= 256) LAUNCHV(float, 1); else if (H <= 512) LAUNCHV(float, 2); else LAUNCHV(float, 4);
        } else {
            if (H <= 256) LAUNCHV(bf16_t, 1); else if (H <= 512) LAUNCHV(bf16_t, 2); else LAUNCHV(bf16_t, 4);
        }
        C2_LAUNCH_CHECK();
        return 0;
    }
#undef LAUNCHV
#define LAUNCH(TO, E)                                                                                              \
    hipLaunchKernelGGL((epilogue_fwd_kernel<TO, E>), grid, dim3(256), 0, st, Y, ldy, nslab, slab_stride, bias, cond, \
                       C, Wc, ldwc, mask, ks, use_ln, B, (int)H, (TO*)act, ldact, yhat, rstd)
    if (act_dtype == C2_F32) {
        if (H <= 256) LAUNCH(float, 1); else if (H <= 1024) LAUNCH(float, 4); else LAUNCH(float, 0);
    } else {
        if (H <= 256) LAUNCH(bf16_t, 1); else if (H <= 1024) LAUNCH(bf16_t, 4); else LAUNCH(bf16_t, 0);
    }
#undef LAUNCH
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_epilogue_bwd(const float* dA, int64_t ldda, int nslab, int64_t slab_stride, const float* extra,
                               int64_t ldextra, const float* extra_scale_dev, const uint8_t* mask, float p_drop,
                               int use_ln, const float* yhat, const float* rstd, int64_t B, int64_t H, void* dh,
                               int dh_dtype, int64_t lddh, void* dh_planes, int64_t ld_planes, void* stream) {
    C2_CHECK(dA && yhat && dh, "null pointer");
    const int64_t ldp = ld_planes ? ld_planes : H;
    C2_CHECK(!dh_planes || ldp >= H, "ld_planes shorter than a row");
    C2_CHECK(!dh_planes || dh_dtype == C2_F32, "planes accompany an fp32 gradient");
    C2_CHECK(!use_ln || rstd, "rstd needed with use_ln");
    C2_CHECK(dh_dtype == C2_F32 || dh_dtype == C2_BF16, "bad dtype");
    C2_CHECK(nslab >= 1 && ldda >= H && lddh >= H && H < (1ll << 31), "bad shape");
    C2_CHECK(p_drop >= 0.f && p_drop < 1.f, "bad dropout rate");
    if (B == 0 || H == 0) return 0;
    const float ks = 1.0f / (1.0f - p_drop);
    dim3 grid((unsigned)((B + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (H % 4 == 0) && H <= 1024 && (ldda % 4 == 0) && (slab_stride % 4 == 0) && (lddh % 4 == 0) &&
                     aligned16(dA) && aligned16(yhat) && (!extra || (aligned16(extra) && ldextra % 4 == 0)) &&
                     ((reinterpret_cast<uintptr_t>(dh) & (dh_dtype == C2_F32 ? 15 : 7)) == 0) &&
                     (!mask || (reinterpret_cast<uintptr_t>(mask) & 3) == 0) &&
                     (!dh_planes || ((reinterpret_cast<uintptr_t>(dh_planes) & 7) == 0 && ldp % 4 == 0 && (B * ldp) % 4 == 0));
    C2_CHECK(vec || !dh_planes, "dh_planes needs the vector path (H % 4 == 0, H <= 1024, aligned rows)");
#define LAUNCHV(TO, E)                                                                                               \
    hipLaunchKernelGGL((epilogue_bwd_vec_kernel<TO, E>), grid, dim3(256), 0, st, dA, ldda, nslab, slab_stride, extra, \
                       ldextra, extra_scale_dev, mask, ks, use_ln, yhat, rstd, B, (int)H, (TO*)dh, lddh,            \
                       (bf16_t*)dh_planes, ldp)
    if (vec) {
        if (dh_dtype == C2_F32) {
            if (H <= 256) LAUNCHV(float, 1); else if (H <= 512) LAUNCHV(float, 2); else LAUNCHV(float, 4);
        } else {
            if (H <= 256) LAUNCHV(bf16_t, 1); else if (H <= 512) LAUNCHV(bf16_t, 2); else LAUNCHV(bf16_t, 4);
        }
        C2_LAUNCH_CHECK();
        return 0;
    }
#undef LAUNCHV
#define LAUNCH(TO, E)                                                                                            \
    hipLaunchKernelGGL((epilogue_bwd_kernel<TO, E>), grid, dim3(256), 0, st, dA, ldda, nslab, slab_stride, extra, \
                       ldextra, extra_scale_dev, mask, ks, use_ln, yhat, rstd, B, (int)H, (TO*)dh, lddh)
    if (dh_dtype == C2_F32) {
        if (H <= 256) LAUNCH(float, 1); else if (H <= 1024) LAUNCH(float, 4); else LAUNCH(float, 0);
    } else {
        if (H <= 256) LAUNCH(bf16_t, 1); else if (H <= 1024) LAUNCH(bf16_t, 4); else LAUNCH(bf16_t, 0);
    }
#undef LAUNCH
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t c2_bias_cond_grad_ws(int64_t B, int64_t H, int C) {
    const int64_t nchunk = (B + 31) / 32;  // the smallest row chunk c2_bias_cond_grad uses
    return (size_t)(nchunk * (1 + C) * H);
}

extern "C" int c2_bias_cond_grad(const void* dh, int dh_dtype, int64_t lddh, const int64_t* cond, int C, int64_t B,
                                 int64_t H, float* db, float* dWc, int64_t lddwc, const float* alpha_dev, float* ws,
                                 void* stream) {
    C2_CHECK(dh && ws, "null pointer");
    C2_CHECK(dh_dtype == C2_F32 || dh_dtype == C2_BF16, "bad dtype");
    C2_CHECK(lddh >= H && C >= 0, "bad shape");
    C2_CHECK(!dWc || (cond && lddwc >= C), "dWc needs cond");
    if (H == 0) return 0;
    if (!dWc) C = 0;
    hipStream_t st = (hipStream_t)stream;
    if (B <= kBCG_DIRECT_B) {  // one launch, no second stage
        dim3 g1((unsigned)((H + 31) / 32));
#define BCD_L(TI, CC, c0)                                                                                              \
    hipLaunchKernelGGL((bias_cond_stage1<TI, CC, true, 16>), g1, dim3(512), 0, st, (const TI*)dh, lddh, cond, C, c0, B, (int)H, \
                       ws, db, dWc, lddwc, alpha_dev)
        for (int c0 = 0; c0 == 0 || c0 < C; c0 += kBCG_MAXC) {
            const int left = C - c0;
            if (dh_dtype == C2_F32) {
                if (left <= 0) BCD_L(float, 0, c0);
                else if (left <= 4) BCD_L(float, 4, c0);
                else BCD_L(float, 8, c0);
            } else {
                if (left <= 0) BCD_L(bf16_t, 0, c0);
                else if (left <= 4) BCD_L(bf16_t, 4, c0);
                else BCD_L(bf16_t, 8, c0);
            }
        }
#undef BCD_L
        C2_LAUNCH_CHECK();
        return 0;
    }
    const int W = dh_dtype == C2_F32 ? 4 : 8;
    // (a width that is not a multiple of W with rows that are: the last chunk of a row reads into the row's pad, whose sums
    // are not written -- the 2-byte activations of widths like 820 have rows padded to 8 elements)
    const int64_t HW = (H + W - 1) / W * W;
    const bool vecp = lddh % W == 0 && lddh >= HW && aligned16(dh) && H >= 64 * W;
    // rows per chunk: 32 for the narrow matrices (more workgroups than 128-row chunks give), 128 otherwise / scalar path
    const int rb = (vecp && H * (int64_t)((B + 127) / 128) < 256ll * 64 * W) ? 32 : kBCG_RB;
    const int nchunk = (int)std::max<int64_t>(1, (B + rb - 1) / rb);
    if (vecp) {
        dim3 gv((unsigned)((HW / W + 63) / 64), nchunk);
#define BCV_L(TI, CC, c0) \
    hipLaunchKernelGGL((bias_stage1_vec<TI, CC>), gv, dim3(256), 0, st, (const TI*)dh, lddh, cond, C, c0, B, (int)H, rb, ws)
        for (int c0 = 0; c0 == 0 || c0 < C; c0 += 4) {
            const int left = C - c0;
            if (dh_dtype == C2_F32) {
                if (left <= 0) BCV_L(float, 0, c0); else BCV_L(float, 4, c0);
            } else {
                if (left <= 0) BCV_L(bf16_t, 0, c0); else BCV_L(bf16_t, 4, c0);
            }
        }
#undef BCV_L
    } else {
    dim3 g1((unsigned)((H + 127) / 128), nchunk);
#define BCG_L(TI, CC, c0) \
    hipLaunchKernelGGL((bias_cond_stage1<TI, CC>), g1, dim3(512), 0, st, (const TI*)dh, lddh, cond, C, c0, B, (int)H, ws)
    for (int c0 = 0; c0 == 0 || c0 < C; c0 += kBCG_MAXC) {
        const int left = C - c0;
        if (dh_dtype == C2_F32) {
            if (left <= 0) BCG_L(float, 0, c0);
            else if (left <= 4) BCG_L(float, 4, c0);
            else BCG_L(float, 8, c0);
        } else {
            if (left <= 0) BCG_L(bf16_t, 0, c0);
            else if (left <= 4) BCG_L(bf16_t, 4, c0);
            else BCG_L(bf16_t, 8, c0);
        }
    }
#undef BCG_L
    }
    const int n2 = (int)((1 + C) * H);
    hipLaunchKernelGGL(bias_cond_stage2, dim3((n2 + 63) / 64), dim3(256), 0, st, ws, nchunk, C, (int)H, db, dWc,
                       lddwc, alpha_dev);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_reparam_kl_fwd(const float* mu, const float* lv, const float* eps, int64_t B, int64_t L, int64_t ld,
                                 float inv_B, int use_own_kl, float* z32, void* zt, int zt_dtype, int64_t ldzt,
                                 float* kl_out, float* ws, void* stream) {
    C2_CHECK(mu && lv && eps && z32 && kl_out && ws, "null pointer");
    C2_CHECK(ld >= L && (!zt || ldzt >= L), "bad leading dimension");
    if (B == 0 || L == 0) return 0;
    const int grid = grid_for(B * L, 256, 1024);
    hipStream_t st = (hipStream_t)stream;
    if (zt && zt_dtype == C2_BF16)
        hipLaunchKernelGGL((reparam_kl_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, mu, lv, eps, B, (int)L, ld,
                           inv_B, use_own_kl, z32, (bf16_t*)zt, ldzt, ws);
    else
        hipLaunchKernelGGL((reparam_kl_fwd_kernel<float>), dim3(grid), dim3(256), 0, st, mu, lv, eps, B, (int)L, ld,
                           inv_B, use_own_kl, z32, (float*)zt, ldzt, ws);
    launch_finalize_sum(ws, grid, kl_out, st);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_reparam_kl_bwd(const float* dZ, int64_t lddz, int nslab, int64_t slab_stride, const float* extra,
                                 int64_t ldextra, const float* extra_scale_dev, const float* mu, const float* lv,
                                 const float* eps, int64_t B, int64_t L, int64_t ld, float inv_B, int use_own_kl,
                                 const float* g_kl_dev, void* dmu, void* dlv, int out_dtype, int64_t ldout,
                                 void* stream) {
    C2_CHECK(dZ && mu && lv && eps && dmu && dlv, "null pointer");
    C2_CHECK(out_dtype == C2_F32 || out_dtype == C2_BF16, "bad dtype");
    C2_CHECK(ld >= L && ldout >= L && lddz >= L && nslab >= 1, "bad shape");
    if (B == 0 || L == 0) return 0;
    const int grid = grid_for(B * L);
    hipStream_t st = (hipStream_t)stream;
    if (out_dtype == C2_BF16)
        hipLaunchKernelGGL((reparam_kl_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, dZ, lddz, nslab, slab_stride,
                           extra, ldextra, extra_scale_dev, mu, lv, eps, B, (int)L, ld, inv_B, use_own_kl, g_kl_dev,
                           (bf16_t*)dmu, (bf16_t*)dlv, ldout);
    else
        hipLaunchKernelGGL((reparam_kl_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, dZ, lddz, nslab, slab_stride,
                           extra, ldextra, extra_scale_dev, mu, lv, eps, B, (int)L, ld, inv_B, use_own_kl, g_kl_dev,
                           (float*)dmu, (float*)dlv, ldout);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_gather_rows_ld(const float* X, int64_t ldX, const uint8_t* goodloc, int64_t ldg, const float* sizefactor,
                                 const int64_t* cond, const int64_t* classes, int64_t G, int C, int P, const int64_t* idx,
                                 int64_t n, float* x_out, int64_t ldx_out, uint8_t* goodloc_out, int64_t ldg_out,
                                 float* sf_out, int64_t* cond_out, int64_t* classes_out, void* xin_out, int xin_dtype,
                                 int64_t ldxin, int apply_log, const float* rowconst, float* rowconst_out,
                                 int32_t* idx32_out, int64_t idx32_len, const float* center, void* xin2_out,
                                 int64_t ldxin2, void* stream) {
    C2_CHECK(X && idx, "null pointer");
    C2_CHECK(ldX >= G && (!goodloc || ldg >= G) && (!x_out || ldx_out >= G) && (!goodloc_out || ldg_out >= G),
             "a row pitch is shorter than the gene count");
    C2_CHECK(!center || xin_out || xin2_out, "center given without xin_out");
    C2_CHECK(!xin2_out || ldxin2 >= G, "ldxin2 too small");
    C2_CHECK(!xin_out || xin_dtype == C2_F32 || xin_dtype == C2_BF16 || xin_dtype == C2_F16, "bad xin dtype");
    C2_CHECK(!rowconst || rowconst_out, "rowconst_out missing");
    C2_CHECK(!idx32_out || idx32_len >= n, "idx32_out shorter than the batch");
    C2_CHECK(!sizefactor || sf_out, "sf_out missing");
    C2_CHECK(!cond || cond_out, "cond_out missing");
    C2_CHECK(!classes || classes_out, "classes_out missing");
    C2_CHECK(!xin_out || ldxin >= G, "ldxin too small");
    if (n == 0 || G == 0) return 0;
    const bool copy_good = goodloc && goodloc_out;
    // the 16-byte walk: over G when that is a multiple of 4, else over G rounded up when every row involved is that long
    const int64_t G4 = (G + 3) / 4 * 4;
    const bool rows_cover = ldX >= G4 && (!x_out || ldx_out >= G4) && (!copy_good || (ldg >= G4 && ldg_out >= G4)) &&
                            (!xin_out || ldxin >= G4) && (!xin2_out || ldxin2 >= G4);
    const int64_t Gw = (G % 4 == 0) ? G : (rows_cover ? G4 : 0);
    const int vec_ok = Gw > 0 && aligned16(X) && ldX % 4 == 0 && (!x_out || (aligned16(x_out) && ldx_out % 4 == 0)) &&
                       (!center || aligned16(center)) &&
                       (!copy_good || (((uintptr_t)goodloc & 3) == 0 && ((uintptr_t)goodloc_out & 3) == 0 && ldg % 4 == 0 &&
                                       ldg_out % 4 == 0));
    const int vec_xin = xin_out && aligned16(xin_out) && (ldxin % 8 == 0);
    const int vec_xin2 = xin2_out && aligned16(xin2_out) && (ldxin2 % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    // short batches with per-gene work: several blocks per row, so that the launch is ~1024 blocks (each at least 1024 genes)
    const bool per_gene = x_out || xin_out || xin2_out || copy_good;
    unsigned parts = 1;
    if (per_gene && n < 1024) parts = (unsigned)std::max<int64_t>(1, std::min<int64_t>((1024 + n - 1) / n, G / 1024));
    const dim3 grid((unsigned)n, parts);
    if (xin_out && xin_dtype == C2_F16)
        hipLaunchKernelGGL((gather_rows_kernel<f16_t>), grid, dim3(256), 0, st, X, goodloc, sizefactor,
                           cond, classes, G, C, P, idx, x_out, goodloc_out, sf_out, cond_out, classes_out,
                           (f16_t*)xin_out, ldxin, apply_log, vec_ok, vec_xin, rowconst, rowconst_out, idx32_out, n, idx32_len,
                           center, (bf16_t*)xin2_out, ldxin2, vec_xin2, ldX, ldg, ldx_out, ldg_out, Gw);
    else if (xin_out && xin_dtype == C2_BF16)
        hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), grid, dim3(256), 0, st, X, goodloc, sizefactor,
                           cond, classes, G, C, P, idx, x_out, goodloc_out, sf_out, cond_out, classes_out,
                           (bf16_t*)xin_out, ldxin, apply_log, vec_ok, vec_xin, rowconst, rowconst_out, idx32_out, n, idx32_len, center,
                           (bf16_t*)xin2_out, ldxin2, vec_xin2, ldX, ldg, ldx_out, ldg_out, Gw);
    else
        hipLaunchKernelGGL((gather_rows_kernel<float>), grid, dim3(256), 0, st, X, goodloc, sizefactor,
                           cond, classes, G, C, P, idx, x_out, goodloc_out, sf_out, cond_out, classes_out,
                           (float*)xin_out, ldxin, apply_log, vec_ok, vec_xin, rowconst, rowconst_out, idx32_out, n, idx32_len, center,
                           (bf16_t*)xin2_out, ldxin2, vec_xin2, ldX, ldg, ldx_out, ldg_out, Gw);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_gather_rows(const float* X, const uint8_t* goodloc, const float* sizefactor, const int64_t* cond,
                              const int64_t* classes, int64_t G, int C, int P, const int64_t* idx, int64_t n,
                              float* x_out, uint8_t* goodloc_out, float* sf_out, int64_t* cond_out,
                              int64_t* classes_out, void* xin_out, int xin_dtype, int64_t ldxin, int apply_log,
                              const float* rowconst, float* rowconst_out, int32_t* idx32_out, int64_t idx32_len,
                              const float* center, void* xin2_out, int64_t ldxin2, void* stream) {
    return c2_gather_rows_ld(X, G, goodloc, G, sizefactor, cond, classes, G, C, P, idx, n, x_out, G, goodloc_out, G, sf_out,
                             cond_out, classes_out, xin_out, xin_dtype, ldxin, apply_log, rowconst, rowconst_out, idx32_out,
                             idx32_len, center, xin2_out, ldxin2, stream);
}

extern "C" int c2_dropout_masks(uint8_t* out, int64_t n, float p_drop, uint64_t seed, uint64_t offset, void* stream) {
    C2_CHECK(out && n >= 0 && aligned16(out), "out must be a 16-byte aligned buffer");
    C2_CHECK(p_drop >= 0.f && p_drop < 1.f, "bad dropout rate");
    if (n == 0) return 0;
    // keep iff byte >= round(256 p): p is quantised to 1/256 (0.05 -> 13/256 = 0.0508); the scale stays 1/(1-p)
    const unsigned thresh8 = (unsigned)lrintf(p_drop * 256.f);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, out, n,
                       thresh8, (unsigned long long)seed, (unsigned long long)offset);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_axpy(const float* x, const float* a_dev, float* y, int64_t n, void* stream) {
    C2_CHECK(x && y && n >= 0, "bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, a_dev, y, n);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_defer_loss_sums(int on) {
    const int was = g_defer_sums ? 1 : 0;
    if (on && !g_defer_sums) g_pending_sums.count = 0;  // nothing left over from a forward that never reached its combine
    g_defer_sums = on != 0;
    return was;
}

// the pending second stages, if any, in the launch that forms the loss vector
static bool flush_pending_sums(const float* losses4, const float* w_dev, float w0, float w1, float w2, float w3, float* out6,
                               double* accum6, hipStream_t st) {
    if (g_pending_sums.count == 0) return false;
    hipLaunchKernelGGL(finalize_combine_kernel, dim3(1), dim3(256), 0, st, g_pending_sums, const_cast<float*>(losses4), w_dev,
                       w0, w1, w2, w3, out6, accum6);
    g_pending_sums.count = 0;
    return true;
}

extern "C" int c2_combine_losses(const float* losses4, float w0, float w1, float w2, float w3, float* out6,
                                 void* stream) {
    C2_CHECK(losses4 && out6, "null pointer");
    if (flush_pending_sums(losses4, nullptr, w0, w1, w2, w3, out6, nullptr, (hipStream_t)stream)) {
        C2_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(combine_losses_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, losses4, w0, w1, w2, w3, out6);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_scale4(const float* g_dev, float w0, float w1, float w2, float w3, float* out4, void* stream) {
    C2_CHECK(out4, "null pointer");
    hipLaunchKernelGGL(scale4_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_dev, w0, w1, w2, w3, out4);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_combine_losses_dev(const float* losses4, const float* w_dev, float* out6, double* accum6,
                                     void* stream) {
    C2_CHECK(losses4 && w_dev && out6, "null pointer");
    if (flush_pending_sums(losses4, w_dev, 0.f, 0.f, 0.f, 0.f, out6, accum6, (hipStream_t)stream)) {
        C2_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(combine_losses_dev_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, losses4, w_dev, out6, accum6);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_scale4_dev(const float* g_dev, const float* w_dev, float* out4, void* stream) {
    C2_CHECK(w_dev && out4, "null pointer");
    hipLaunchKernelGGL(scale4_dev_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_dev, w_dev, out4);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_scale4_dev_scale2(const float* g_dev, const float* w_dev, float* out4, const float* x1, float* y1,
                                    int64_t n1, const float* x2, float* y2, int64_t n2, void* stream) {
    C2_CHECK(w_dev && out4 && n1 >= 0 && n2 >= 0 && (n1 == 0 || (x1 && y1)) && (n2 == 0 || (x2 && y2)), "null pointer");
    hipLaunchKernelGGL(scale4_dev_scale2_kernel, dim3(grid_for(std::max<int64_t>(n1 + n2, 1))), dim3(256), 0,
                       (hipStream_t)stream, g_dev, w_dev, out4, x1, y1, n1, x2, y2, n2);
    C2_LAUNCH_CHECK();
    return 0;
}

static int noise_grid_cap() {  // C2_NOISE_GRID: blocks of the noise launch at most (each takes one ticket: same-address atomics serialise)
    static const int v = [] { const char* e = getenv("C2_NOISE_GRID"); return e ? std::max(1, atoi(e)) : 256; }();
    return v;
}

extern "C" int c2_noise_dev(uint8_t* masks, int64_t n_mask, float p_drop, float* eps, int64_t n_eps, uint64_t eps_first,
                            uint64_t seed, uint64_t* state, uint64_t lead, uint64_t advance, void* stream) {
    C2_CHECK(state && n_mask >= 0 && n_eps >= 0 && (n_mask == 0 || masks) && (n_eps == 0 || eps), "null pointer or negative count");
    C2_CHECK(n_mask == 0 || aligned16(masks), "masks must be a 16-byte aligned buffer");
    C2_CHECK(n_mask == 0 || (p_drop >= 0.f && p_drop < 1.f), "bad dropout rate");
    if (n_mask == 0 && n_eps == 0 && advance == 0) return 0;
    const unsigned thresh8 = (unsigned)lrintf(p_drop * 256.f);
    const int64_t work = std::max<int64_t>((n_mask + 15) / 16, (n_eps + 3) / 4 + 1);
    hipLaunchKernelGGL(noise_dev_kernel, dim3(grid_for(work, 256, noise_grid_cap())), dim3(256), 0, (hipStream_t)stream, masks, n_mask, thresh8, eps,
                       n_eps, (unsigned long long)eps_first, (unsigned long long)seed, (unsigned long long*)state,
                       (unsigned long long)lead, (unsigned long long)advance);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_dropout_masks_dev(uint8_t* out, int64_t n, float p_drop, uint64_t seed, uint64_t* offset_dev,
                                    uint64_t lead, uint64_t advance, void* stream) {
    C2_CHECK(out && offset_dev && n >= 0 && aligned16(out), "out must be a 16-byte aligned buffer");
    C2_CHECK(p_drop >= 0.f && p_drop < 1.f, "bad dropout rate");
    if (n > 0) {
        const unsigned thresh8 = (unsigned)lrintf(p_drop * 256.f);
        hipLaunchKernelGGL(dropout_mask_dev_kernel, dim3(grid_for((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, out, n,
                           thresh8, (unsigned long long)seed, (const unsigned long long*)offset_dev,
                           (unsigned long long)lead);
    }
    hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)offset_dev,
                       (unsigned long long)advance);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_graph_launch(void* graph_exec, void* stream) {
    C2_CHECK(graph_exec, "null graph");
    const hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) {
        c2::set_error(std::string("c2_graph_launch: hipGraphLaunch: ") + hipGetErrorString(e));
        return 2;
    }
    return 0;
}

extern "C" int c2_scale(const float* x, const float* a_dev, float* y, int64_t n, void* stream) {
    C2_CHECK(x && a_dev && y, "null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, a_dev, y, n);
    C2_LAUNCH_CHECK();
    return 0;
}

// dst[c][r] = src[r][c] for 2-byte elements: 64 x 64 tiles through LDS.  VEC (16-byte aligned rows on both sides, whole
// tiles): a thread loads two 16-byte chunks of a source row and stores two 16-byte chunks of a destination row (eight lanes
// cover 128 contiguous bytes); the tile's rows are 33 words apart, so the eight row chunks x four column words a wave reads
// per step fall on distinct banks.
template <bool VEC>
__global__ __launch_bounds__(256) void transpose16_kernel(const bf16_t* __restrict__ src, int64_t lds_, int64_t rows,
                                                          int64_t cols, bf16_t* __restrict__ dst, int64_t ldd) {
    __shared__ bf16_t tile[64][66];
    const int64_t r0 = blockIdx.y * 64ll, c0 = blockIdx.x * 64ll;
    if (VEC) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = threadIdx.x + 256 * h, r = q >> 3, ch = q & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(src + (r0 + r) * lds_ + c0 + 8 * ch);
            unsigned* t = reinterpret_cast<unsigned*>(&tile[r][8 * ch]);
            t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = threadIdx.x + 256 * h, c = q >> 3, rc = q & 7;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = (unsigned)tile[8 * rc + 2 * e][c] | ((unsigned)tile[8 * rc + 2 * e + 1][c] << 16);
            *reinterpret_cast<u32x4*>(dst + (c0 + c) * ldd + r0 + 8 * rc) = o;
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < rows && c < cols) ? src[r * lds_ + c] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[c * ldd + r] = tile[tx][ty + 4 * i];
    }
}

extern "C" int c2_transpose_bf16(const void* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, int64_t ld_dst,
                                 void* stream) {
    C2_CHECK(src && dst && ld_src >= cols && ld_dst >= rows && rows >= 0 && cols >= 0, "bad arguments");
    if (rows == 0 || cols == 0) return 0;
    C2_CHECK((rows + 63) / 64 < 65536, "too many rows");
    const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    const bool vec = rows % 64 == 0 && cols % 64 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && aligned16(src) && aligned16(dst);
    if (vec)
        hipLaunchKernelGGL(transpose16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, rows,
                           cols, (bf16_t*)dst, ld_dst);
    else
        hipLaunchKernelGGL(transpose16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src,
                           rows, cols, (bf16_t*)dst, ld_dst);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
    C2_CHECK(src && dst && n >= 0, "bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_cast_f16(const float* src, void* dst, int64_t n, void* stream) {
    C2_CHECK(src && dst && n >= 0, "bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_f16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (f16_t*)dst, n);
    C2_LAUNCH_CHECK();
    return 0;
}

static int col_splits(int64_t rows) { return (int)std::max<int64_t>(1, std::min<int64_t>(64, rows / 64)); }

extern "C" size_t c2_col_sum_ws(int64_t rows, int64_t cols) { return (size_t)col_splits(rows) * (size_t)cols; }

extern "C" int c2_col_sum(const float* x, int64_t ldx, int64_t rows, int64_t cols, int apply_log, double* sums, double* ws,
                          void* stream) {
    C2_CHECK(rows >= 0 && cols >= 0, "negative size");
    if (cols == 0) return 0;
    C2_CHECK(sums && ws, "null pointer");
    C2_CHECK(rows == 0 || (x && ldx >= cols), "bad arguments");
    const int ns = col_splits(rows);
    hipStream_t st = (hipStream_t)stream;
    const unsigned gx = (unsigned)((cols + 255) / 256);
    hipLaunchKernelGGL(col_sum_stage1, dim3(gx, ns), dim3(256), 0, st, x, ldx, rows, cols, apply_log, ns, ws);
    hipLaunchKernelGGL(col_sum_stage2, dim3(gx), dim3(256), 0, st, (const double*)ws, ns, cols, sums);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_col_mean_from_sums(const double* sums, int64_t cols, int64_t total_rows, float* out, void* stream) {
    C2_CHECK(cols >= 0 && total_rows > 0, "bad sizes (the mean of no rows is undefined)");
    if (cols == 0) return 0;
    C2_CHECK(sums && out, "null pointer");
    hipLaunchKernelGGL(col_mean_finish, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sums, cols,
                       1.0 / (double)total_rows, out);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_gemv_bias(const float* W, int64_t ldw, int64_t N, int64_t K, const float* c, const float* bias, float* out,
                            void* stream) {
    C2_CHECK(N >= 0 && K >= 0, "negative size");
    if (N == 0) return 0;
    C2_CHECK(W && c && out && ldw >= K, "bad arguments");
    const int vec = (K % 4 == 0) && (ldw % 4 == 0) && aligned16(W) && aligned16(c);
    hipLaunchKernelGGL(gemv_bias_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, c, bias, out, vec);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_rank1_add(float* G, int64_t ldg, int64_t N, int64_t K, const float* r, const float* c, void* stream) {
    C2_CHECK(N >= 0 && K >= 0, "negative size");
    if (N == 0 || K == 0) return 0;
    C2_CHECK(G && r && c && ldg >= K, "bad arguments");
    const int vec = (K % 4 == 0) && (ldg % 4 == 0) && aligned16(G) && aligned16(c);
    hipLaunchKernelGGL(rank1_add_kernel, dim3(grid_for(N * K / (vec ? 4 : 1))), dim3(256), 0, (hipStream_t)stream, G, ldg, N, K,
                       r, c, vec);
    C2_LAUNCH_CHECK();
    return 0;
}

// (a forward that raised between its loss kernels and its combine call: forget the second stages it left pending)
extern "C" int c2_drop_pending_sums(void) {
    g_pending_sums.count = 0;
    return 0;
}
