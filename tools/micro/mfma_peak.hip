// Sustained rate of v_mfma_f32_16x16x4_f32 with operands in registers: what the matrix pipe gives a
// kernel that does nothing else.  ./mfma_peak [waves per SIMD = 1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(1024) void mfma_loop(float *out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 16; ++i) s += acc[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = s[0];
}

int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    int n_cu = 256;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    float *out;
    hipMalloc(&out, 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(n_cu), dim3(256 * wps), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)n_cu * 4 * wps * iters * 64 * 2048.0;
        printf("%d CUs x %d waves/SIMD: %.3f ms  %.1f TFLOP/s  (%.2f cycles per MFMA per SIMD at 2.4 GHz)\n", n_cu, wps, ms, flops / ms * 1e-9,
               ms * 1e-3 * 2.4e9 / ((double)wps * iters * 64));
    }
    return 0;
}
