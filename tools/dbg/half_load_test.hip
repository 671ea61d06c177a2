// debugging aid: what do a b64 raw buffer load of four halves and two ways of widening them return?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const _Float16* in, float* a, float* b, int n) {
    const int tid = threadIdx.x;
    auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in), 0, n * 2, 0x00020000u);
    u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, tid * 8, 0, 0));
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 lo = __builtin_bit_cast(h2, v[0]), hi = __builtin_bit_cast(h2, v[1]);
    f32x4 w = f32x4{(float)lo[0], (float)lo[1], (float)hi[0], (float)hi[1]};
    for (int e = 0; e < 4; ++e) a[4 * tid + e] = w[e];
    for (int e = 0; e < 4; ++e) {
        unsigned short bits = (unsigned short)((v[e >> 1] >> (16 * (e & 1))) & 0xffffu);
        b[4 * tid + e] = (float)__builtin_bit_cast(_Float16, bits);
    }
}
int main() {
    const int n = 256;
    _Float16 h[n]; for (int i = 0; i < n; ++i) h[i] = (_Float16)(float)i;
    _Float16* d; float *a, *b; hipMalloc(&d, n * 2); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, b, n);
    float ha[n], hb[n]; hipMemcpy(ha, a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("%d: vec %g  int %g\n", i, ha[i], hb[i]);
    int bad_a = 0, bad_b = 0; for (int i = 0; i < n; ++i) { bad_a += ha[i] != (float)i; bad_b += hb[i] != (float)i; }
    printf("bad vec %d bad int %d\n", bad_a, bad_b);
    return 0;
}
