// Fused multi-tensor Adam (torch.optim.Adam semantics as celligner2/trainers/celligner2/trainer.py:262-266,316-323
// sets it up: L2 weight decay folded into the gradient, optional clip_grad_value_, bias correction) for gfx950.
// One launch updates up to kMaxT tensors: 28 B/param of HBM traffic (+2 B when a bf16 shadow copy of the
// weights is refreshed for the bf16-operand GEMMs), 16-byte accesses, no atomics.
#include "common.h"

namespace c2 {

constexpr int kMaxT = 48;
constexpr int kIter = 4;                          // 16-byte chunks per thread and array
constexpr int kElemsPerBlock = 256 * 4 * kIter;

struct AdamTable {
    c2_adam_tensor t[kMaxT];
    int first_block[kMaxT + 1];
    int n;
};

struct AdamHyper {
    float step_size, beta1, beta2, omb1, omb2, eps, wd, clip, inv_bc2_sqrt;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamHyper& h) {
    if (h.clip > 0.f) g = fminf(fmaxf(g, -h.clip), h.clip);
    g = g + h.wd * p;
    m = h.beta1 * m + h.omb1 * g;
    v = h.beta2 * v + h.omb2 * g * g;
    const float denom = sqrtf(v) * h.inv_bc2_sqrt + h.eps;
    p = p - h.step_size * (m / denom);
}

// operand copies the GEMMs read: bf16(p), and bf16(p - bf16(p)) for the narrow layers' planes
__device__ __forceinline__ int64_t shadow_index(const c2_adam_tensor& t, int64_t i) {
    if (t.shadow_cols == 0) return i;
    // (rows shadow_ld apart: planes in padded storage.  32-bit division where the index allows: the 64-bit one costs more
    // than the rest of the update of a chunk)
    const int64_t r = i < (1ll << 32) ? (int64_t)((unsigned)i / (unsigned)t.shadow_cols) : i / t.shadow_cols;
    return r * t.shadow_ld + (i - r * t.shadow_cols);
}
__device__ __forceinline__ void store_shadows(const c2_adam_tensor& t, int64_t i, const f32x4& p) {
    u32x2 pk;
    if (t.shadow_cols & 3) {  // a width that is not a multiple of 4: a chunk of four may straddle rows -- element by element
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t si = shadow_index(t, i + e);
            if (t.shadow_f16) {
                reinterpret_cast<f16_t*>(t.shadow_bf16)[si] = (f16_t)p[e];
            } else {
                const bf16_t hb = f2bf(p[e]);
                reinterpret_cast<bf16_t*>(t.shadow_bf16)[si] = hb;
                if (t.shadow_lo) reinterpret_cast<bf16_t*>(t.shadow_lo)[si] = f2bf(p[e] - bf2f(hb));
            }
        }
        return;
    }
    i = shadow_index(t, i);  // (a chunk of four never straddles rows: shadow_cols % 4 == 0)
    if (t.shadow_f16) {  // an IEEE-half copy (the encoder's first layer in bf16 mode): no low plane
        pk[0] = pack2<f16_t>(p[0], p[1]);
        pk[1] = pack2<f16_t>(p[2], p[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(t.shadow_bf16) + i) = pk;
        return;
    }
    pk[0] = (unsigned)f2bf(p[0]) | ((unsigned)f2bf(p[1]) << 16);
    pk[1] = (unsigned)f2bf(p[2]) | ((unsigned)f2bf(p[3]) << 16);
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(t.shadow_bf16) + i) = pk;
    if (t.shadow_lo) {
        bf16_t l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) l[e] = f2bf(p[e] - bf2f(f2bf(p[e])));
        u32x2 pl;
        pl[0] = (unsigned)l[0] | ((unsigned)l[1] << 16);
        pl[1] = (unsigned)l[2] | ((unsigned)l[3] << 16);
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(t.shadow_lo) + i) = pl;
    }
}

// Step state on the device (c2_adam_state of c2mi.h): one thread advances the step count and turns the hyper-parameters
// into the kernel's fp32 constants exactly as the host does for c2_adam_step (double arithmetic, rounded once).
__global__ void adam_prepare_kernel(c2_adam_state* st, AdamHyper* h) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->step += 1;
    const double bc1 = 1.0 - pow(st->beta1, (double)st->step);
    const double bc2 = 1.0 - pow(st->beta2, (double)st->step);
    h->step_size = (float)(st->lr / bc1);
    h->beta1 = (float)st->beta1; h->beta2 = (float)st->beta2;
    h->omb1 = (float)(1.0 - st->beta1); h->omb2 = (float)(1.0 - st->beta2);
    h->eps = (float)st->eps; h->wd = (float)st->weight_decay; h->clip = (float)st->clip_value;
    h->inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void adam_kernel(AdamTable tab, AdamHyper h);
__global__ __launch_bounds__(256) void adam_dev_kernel(AdamTable tab, const AdamHyper* hp);

__device__ __forceinline__ void adam_block(const AdamTable& tab, const AdamHyper& h);

__global__ __launch_bounds__(256) void adam_kernel(AdamTable tab, AdamHyper h) { adam_block(tab, h); }
__global__ __launch_bounds__(256) void adam_dev_kernel(AdamTable tab, const AdamHyper* hp) {
    const AdamHyper h = *hp;  // (uniform address: scalar loads)
    adam_block(tab, h);
}

__device__ __forceinline__ void adam_block(const AdamTable& tab, const AdamHyper& h) {
    // locate the tensor of this block (tab.n <= 48: linear scan on scalar registers)
    int ti = 0;
    while (ti + 1 < tab.n && (int)blockIdx.x >= tab.first_block[ti + 1]) ++ti;
    const c2_adam_tensor t = tab.t[ti];
    const int64_t base = (int64_t)(blockIdx.x - tab.first_block[ti]) * kElemsPerBlock;
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) |
                       reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 15) == 0 &&
                     (!t.shadow_bf16 || (reinterpret_cast<uintptr_t>(t.shadow_bf16) & 7) == 0) &&
                     (!t.shadow_lo || (reinterpret_cast<uintptr_t>(t.shadow_lo) & 7) == 0);
    // All of the block's loads are issued before the first store: p, g, m and v come through pointers the compiler must
    // assume to alias, so a load placed after a store in the source stays after it -- with load / update / store per
    // chunk only four 16-byte loads per thread were ever in flight.
    if (vec && base + kElemsPerBlock <= t.n) {
        f32x4 p[kIter], g[kIter], m[kIter], v[kIter];
#pragma unroll
        for (int k = 0; k < kIter; ++k) {
            const int64_t i = base + (int64_t)(threadIdx.x + 256 * k) * 4;
#ifndef C2_ADAM_NO_NT  // streamed once per step: keep them out of the caches (measured 5.2 -> 6.1 TB/s)
            p[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(t.p + i));
            g[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(t.g + i));
            m[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(t.m + i));
            v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(t.v + i));
#else
            p[k] = *reinterpret_cast<const f32x4*>(t.p + i);
            g[k] = *reinterpret_cast<const f32x4*>(t.g + i);
            m[k] = *reinterpret_cast<const f32x4*>(t.m + i);
            v[k] = *reinterpret_cast<const f32x4*>(t.v + i);
#endif
        }
#pragma unroll
        for (int k = 0; k < kIter; ++k) {
            const int64_t i = base + (int64_t)(threadIdx.x + 256 * k) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[k][e], me = m[k][e], ve = v[k][e];
                adam_one(pe, g[k][e], me, ve, h);
                p[k][e] = pe; m[k][e] = me; v[k][e] = ve;
            }
#ifndef C2_ADAM_NO_NT
            __builtin_nontemporal_store(p[k], reinterpret_cast<f32x4*>(t.p + i));
            __builtin_nontemporal_store(m[k], reinterpret_cast<f32x4*>(t.m + i));
            __builtin_nontemporal_store(v[k], reinterpret_cast<f32x4*>(t.v + i));
#else
            *reinterpret_cast<f32x4*>(t.p + i) = p[k];
            *reinterpret_cast<f32x4*>(t.m + i) = m[k];
            *reinterpret_cast<f32x4*>(t.v + i) = v[k];
#endif
            if (t.shadow_bf16) store_shadows(t, i, p[k]);
        }
        return;
    }
    for (int k = 0; k < kIter; ++k) {  // a tensor's last block, unaligned tensors
        const int64_t i = base + (int64_t)(threadIdx.x + 256 * k) * 4;
        if (i >= t.n) continue;
        if (vec && i + 4 <= t.n) {
            f32x4 p = *reinterpret_cast<f32x4*>(t.p + i);
            const f32x4 g = *reinterpret_cast<const f32x4*>(t.g + i);
            f32x4 m = *reinterpret_cast<f32x4*>(t.m + i);
            f32x4 v = *reinterpret_cast<f32x4*>(t.v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e];
                adam_one(pe, g[e], me, ve, h);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
            *reinterpret_cast<f32x4*>(t.p + i) = p;
            *reinterpret_cast<f32x4*>(t.m + i) = m;
            *reinterpret_cast<f32x4*>(t.v + i) = v;
            if (t.shadow_bf16) store_shadows(t, i, p);
        } else {
            for (int e = 0; e < 4 && i + e < t.n; ++e) {
                float pe = t.p[i + e], me = t.m[i + e], ve = t.v[i + e];
                adam_one(pe, t.g[i + e], me, ve, h);
                t.p[i + e] = pe; t.m[i + e] = me; t.v[i + e] = ve;
                const int64_t si = t.shadow_bf16 ? shadow_index(t, i + e) : 0;
                if (t.shadow_bf16 && t.shadow_f16) {
                    reinterpret_cast<f16_t*>(t.shadow_bf16)[si] = (f16_t)pe;
                } else if (t.shadow_bf16) {
                    const bf16_t hb = f2bf(pe);
                    reinterpret_cast<bf16_t*>(t.shadow_bf16)[si] = hb;
                    if (t.shadow_lo) reinterpret_cast<bf16_t*>(t.shadow_lo)[si] = f2bf(pe - bf2f(hb));
                }
            }
        }
    }
}

}  // namespace c2

using namespace c2;

// hp_dev == NULL: hyper-parameters by value (h); else read from device memory by every block
static int adam_launch(const c2_adam_tensor* desc_host, int n_tensors, const AdamHyper& h, const AdamHyper* hp_dev,
                       void* stream) {
    for (int i = 0; i < n_tensors;) {  // one launch per kMaxT NON-EMPTY descriptors; i = descriptors consumed so far
        AdamTable tab;
        tab.n = 0;
        int blocks = 0;
        for (; i < n_tensors && tab.n < kMaxT; ++i) {
            const c2_adam_tensor& d = desc_host[i];
            if (d.n == 0) continue;
            C2_CHECK(d.p && d.g && d.m && d.v && d.n > 0, "bad tensor descriptor");
            C2_CHECK(!d.shadow_lo || d.shadow_bf16, "shadow_lo without the high plane");
            C2_CHECK(d.shadow_cols == 0 || (d.shadow_cols > 0 && d.shadow_ld >= d.shadow_cols && d.shadow_ld % 4 == 0 &&
                                            d.n % d.shadow_cols == 0),
                     "pitched shadows need shadow_ld % 4 == 0 >= shadow_cols and n a whole number of rows");
            C2_CHECK(!(d.shadow_f16 && d.shadow_lo), "an f16 shadow has no low plane");
            tab.t[tab.n] = d;
            tab.first_block[tab.n] = blocks;
            blocks += (int)((d.n + kElemsPerBlock - 1) / kElemsPerBlock);
            ++tab.n;
        }
        tab.first_block[tab.n] = blocks;
        if (blocks == 0) continue;
        if (hp_dev) hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tab, hp_dev);
        else hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tab, h);
    }
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_adam_step(const c2_adam_tensor* desc_host, int n_tensors, double lr, double beta1, double beta2,
                            double eps, double weight_decay, double clip_value, int64_t step_t, void* stream) {
    C2_CHECK(desc_host || n_tensors == 0, "null descriptor table");
    C2_CHECK(step_t >= 1, "step_t must be >= 1");
    const double bc1 = 1.0 - pow(beta1, (double)step_t);
    const double bc2 = 1.0 - pow(beta2, (double)step_t);
    AdamHyper h;
    h.step_size = (float)(lr / bc1);
    h.beta1 = (float)beta1; h.beta2 = (float)beta2;
    h.omb1 = (float)(1.0 - beta1); h.omb2 = (float)(1.0 - beta2);  // as torch: 1-beta in double
    h.eps = (float)eps; h.wd = (float)weight_decay; h.clip = (float)clip_value;
    h.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    return adam_launch(desc_host, n_tensors, h, nullptr, stream);
}

extern "C" size_t c2_adam_hyper_bytes(void) { return sizeof(AdamHyper); }

extern "C" int c2_adam_prepare(c2_adam_state* state_dev, void* hyper_dev, void* stream) {
    C2_CHECK(state_dev && hyper_dev, "null pointer");
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state_dev, (AdamHyper*)hyper_dev);
    C2_LAUNCH_CHECK();
    return 0;
}

extern "C" int c2_adam_step_dev(const c2_adam_tensor* desc_host, int n_tensors, const void* hyper_dev, void* stream) {
    C2_CHECK(desc_host || n_tensors == 0, "null descriptor table");
    C2_CHECK(hyper_dev, "null hyper-parameter block");
    AdamHyper unused{};
    return adam_launch(desc_host, n_tensors, unused, (const AdamHyper*)hyper_dev, stream);
}
