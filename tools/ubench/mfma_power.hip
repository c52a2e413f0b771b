// Micro-benchmark 3: does the operand DATA change MFMA throughput (power management)?  256 CUs x 8 waves run
// back-to-back v_mfma_f32_32x32x16_f16 on (a) zeros, (b) ones, (c) random fp16 operands that change every step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

__global__ __launch_bounds__(512, 1) void mfma_kernel(const half8* __restrict__ src, int iters, float* sink) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 0xffff]; b[i] = src[(tid * 8 + 4 + i) & 0xffff]; }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x16{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m)
            acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[(m >> 2) & 3], acc[m & 7], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    const int n = 0x10000;
    std::vector<_Float16> h(n * 8);
    half8* src; float* sink;
    (void)hipMalloc(&src, n * 16); (void)hipMalloc(&sink, 64);
    const char* names[3] = {"zeros", "ones", "random N(0,1)"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < n * 8; ++i) {
            float u = (rand() / (float)RAND_MAX + rand() / (float)RAND_MAX + rand() / (float)RAND_MAX - 1.5f) * 2.f;
            h[i] = (_Float16)(mode == 0 ? 0.f : mode == 1 ? 1.f : u);
        }
        (void)hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
        const int iters = 20000;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(512), 0, 0, src, iters, sink);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(512), 0, 0, src, iters, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = 256.0 * 8 * iters * 24 * 2.0 * 32 * 32 * 16;
        printf("%-14s %8.3f ms  %7.1f TFLOP/s  -> %5.3f us per 24-MFMA slab per wave pair (%.2f GHz-equivalent)\n", names[mode],
               ms, flops / ms / 1e9, ms * 1e3 / iters, 2 * 24 * 32 / (ms * 1e3 / iters) / 1e3);
    }
    return 0;
}
