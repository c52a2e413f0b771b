// Micro-benchmark 5 (r04): in the POWER-limited regime (random operands, tools/ubench/mfma_power.hip: 1.70 of 2.47 PFLOP/s), does
// the ORDER in which a wave cycles through its operand fragments, or the MFMA shape, change the sustained rate?  256 CUs x 8 waves,
// back-to-back MFMAs, accumulator reuse distance 8, random N(0,1)-like fp16 operands.
//   v0  32x32x16: A changes every MFMA, B every 4th            (mfma_power.hip's pattern)
//   v1  32x32x16: A and B constant (one fragment pair for all MFMAs)
//   v2  32x32x16: B changes every MFMA, A every 4th
//   v3  32x32x16: Gray order -- exactly one operand changes between consecutive MFMAs
//   v4  32x32x16: the conv main loop's order: (a_i, b_j) for all i, j of hi x hi, then hi x lo, then lo x hi (2 x 2 blocks)
//   v5  32x32x16: per block pair the three products back to back: (ah_i, bh_j), (ah_i, bl_j), (al_i, bh_j)
//   v6  16x16x32: A changes every MFMA, B every 4th
//   v7  16x16x32: A and B constant
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

template <int V>
__global__ __launch_bounds__(512, 1) void k32(const half8* __restrict__ src, int iters, float* sink) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    half8 a[4], b[4];                      // v4 / v5: a = {ah0, ah1, al0, al1}, b = {bh0, bh1, bl0, bl1}
    for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 0xffff]; b[i] = src[(tid * 8 + 4 + i) & 0xffff]; }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x16{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            int ia, ib, ic = m & 7;
            if (V == 0) { ia = m & 3; ib = (m >> 2) & 3; }
            if (V == 1) { ia = 0; ib = 0; }
            if (V == 2) { ia = (m >> 2) & 3; ib = m & 3; }
            if (V == 3) { const int g = m ^ (m >> 1); ia = (g >> 1) & 3; ib = ((g >> 3) & 1) * 2 + (g & 1); ia = ((m + 1) >> 1) & 3; ib = (m >> 1) & 3; }
            if (V == 4) {      // 12 MFMAs per k-step: hi x hi (i, j), hi x lo, lo x hi; two k-steps
                const int q = m % 12, p = q >> 2, i = (q >> 1) & 1, j = q & 1;
                ia = (p == 2 ? 2 : 0) + i; ib = (p == 1 ? 2 : 0) + j; ic = (p == 0 ? 0 : 4) + 2 * i + j;
            }
            if (V == 5) {      // per (i, j): hh, hl, lh back to back
                const int q = m % 12, blk = q / 3, p = q % 3, i = blk >> 1, j = blk & 1;
                ia = (p == 2 ? 2 : 0) + i; ib = (p == 1 ? 2 : 0) + j; ic = (p == 0 ? 0 : 4) + 2 * i + j;
            }
            acc[ic] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc[ic], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) sink[0] = s;
}

template <int V>
__global__ __launch_bounds__(512, 1) void k16(const half8* __restrict__ src, int iters, float* sink) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 0xffff]; b[i] = src[(tid * 8 + 4 + i) & 0xffff]; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            const int ia = V == 6 ? (m & 3) : 0, ib = V == 6 ? ((m >> 2) & 3) : 0;
            acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ia], b[ib], acc[m & 7], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
}

template <class K>
void run(const char* name, K kern, const half8* src, float* sink) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, src, iters, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, src, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * iters * 24 * 2.0 * 32 * 32 * 16;
    printf("%-52s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}

int main() {
    const int n = 0x10000;
    std::vector<_Float16> h(n * 8);
    half8* src; float* sink;
    (void)hipMalloc(&src, n * 16); (void)hipMalloc(&sink, 64);
    for (int i = 0; i < n * 8; ++i)
        h[i] = (_Float16)((rand() / (float)RAND_MAX + rand() / (float)RAND_MAX + rand() / (float)RAND_MAX - 1.5f) * 2.f);
    (void)hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run("v0 32x32x16  A every MFMA, B every 4th", k32<0>, src, sink);
        run("v1 32x32x16  A, B constant", k32<1>, src, sink);
        run("v2 32x32x16  B every MFMA, A every 4th", k32<2>, src, sink);
        run("v3 32x32x16  one operand changes per MFMA", k32<3>, src, sink);
        run("v4 32x32x16  conv order (hh all, hl all, lh all)", k32<4>, src, sink);
        run("v5 32x32x16  per block: hh, hl, lh", k32<5>, src, sink);
        run("v6 16x16x32  A every MFMA, B every 4th", k16<6>, src, sink);
        run("v7 16x16x32  A, B constant", k16<7>, src, sink);
    }
    return 0;
}
