// Does an f32 matrix instruction stream leave room for another wave's vector instructions on the same SIMD?
// Block of 8 waves (two per SIMD): waves 0-3 run v_mfma_f32_16x16x4_f32 back to back, waves 4-7 run a stream of
// plain / packed f32 adds or integer adds.  Prints shader cycles of the block for each side alone and for both.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu.cpp -o tools/micro/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int KIND, int MK = 0>
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, int n_mfma, int n_valu, int swap, int prio) {
    const int wave = (threadIdx.x >> 6) ^ (swap ? 4 : 0);
    if (prio == 1 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    if (prio == 2 && wave < 4) __builtin_amdgcn_s_setprio(3);
    const long long t0 = clock64();
    if (wave < 4) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
        if (MK == 0) {
            for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
        } else {                             // bf16 operands: v_mfma_f32_16x16x32_bf16, eight values per lane and operand
            using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
            bf16x8 va, vb;
            for (int i = 0; i < 8; ++i) { va[i] = (__bf16)(a + i); vb[i] = (__bf16)(b - i); }
            for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[i], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
        const float y = 1.0001f;
        for (int it = 0; it < n_valu; ++it) {
            if (KIND == 0) {
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
            } else if (KIND == 1) {
                f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, yy = {y, y};
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(yy));
                x0 = p0[0]; x1 = p0[1]; x2 = p1[0]; x3 = p1[1]; x4 = p2[0]; x5 = p2[1]; x6 = p3[0]; x7 = p3[1];
            } else {
                int i0 = __float_as_int(x0), i1 = __float_as_int(x1), i2 = __float_as_int(x2), i3 = __float_as_int(x3);
                int i4 = __float_as_int(x4), i5 = __float_as_int(x5), i6 = __float_as_int(x6), i7 = __float_as_int(x7);
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(3));
                x0 = __int_as_float(i0); x1 = __int_as_float(i1); x2 = __int_as_float(i2); x3 = __int_as_float(i3);
                x4 = __int_as_float(i4); x5 = __int_as_float(i5); x6 = __int_as_float(i6); x7 = __int_as_float(i7);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    if ((threadIdx.x & 63) == 0) cyc[256 + blockIdx.x * 8 + wave] = clock64() - t0;   // (logical wave: 0-3 matrix, 4-7 vector)   // when this wave was done
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
}

template <int KIND, int MK = 0>
void run(const char *name, float *out, long long *cyc, int swap = 0, int prio = 0) {
    const int NM = 4096, NV = 4096;   // 32768 MFMA (x32 cycles = 1.05 M cycles), 32768 vector instructions per wave
    for (int mode = 0; mode < 3; ++mode) {
        const int nm = mode != 1 ? NM : 0, nv = mode != 0 ? NV : 0;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<KIND, MK>), dim3(256), dim3(512), 0, 0, out, cyc, nm, nv, swap, prio);
        hipDeviceSynchronize();
        static long long h[256 * 9];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        long long s = 0, sm = 0, sv = 0;
        for (int i = 0; i < 256; ++i) { s += h[i]; sm += h[256 + i * 8 + 1]; sv += h[256 + i * 8 + 5]; }
        printf("   (wave 1, matrix side, done after %.0f cycles; wave 5, vector side, after %.0f)\n", sm / 256.0, sv / 256.0);
        printf("%-12s %-10s %9.0f cycles per block  (%.1f per vector instruction, %.1f per MFMA)\n", name,
               mode == 0 ? "mfma only" : mode == 1 ? "valu only" : "both", s / 256.0, nv ? s / 256.0 / (8.0 * nv) : 0.0,
               nm ? s / 256.0 / (8.0 * nm) : 0.0);
    }
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 9 * 8);
    run<0>("v_add_f32", out, cyc);
    run<0>("add swapped", out, cyc, 1, 0);
    run<0>("add prio-v", out, cyc, 0, 1);
    run<0>("add prio-m", out, cyc, 0, 2);
    run<0>("add sw+prio-v", out, cyc, 1, 1);
    run<1>("v_pk_add_f32", out, cyc, 1, 1);
    run<0, 1>("bf16 + add", out, cyc);
    run<0, 1>("bf16 + add sw", out, cyc, 1, 0);
    return 0;
}
