// Reproducer for the fine_match SLP-build deviations (r06; tools/studies/fine_bisect.py found the instruction):
//
//     v_pk_mul_f32 v[152:153], v[202:203], v[152:153] op_sel:[0,1]
//
// -- a packed fp32 multiply whose LOW result takes the HIGH half of src1 (op_sel crossing) and whose destination IS src1.  With one wave
// per SIMD the kernel that contains it is bit-reproducible; with two workgroups per CU (two waves per SIMD) ~3e-4 of its executions
// deliver a wrong low half.  This file runs that instruction in isolation, self-checking, at two waves per SIMD, next to the variants
// the bisect showed to be clean:
//   mode 0  in place, crossing           v_pk_mul_f32 D, S, D op_sel:[0,1]           lo = S.lo * D.hi, hi = S.hi * D.hi
//   mode 1  crossing, NOT in place       v_pk_mul_f32 T, S, D op_sel:[0,1]
//   mode 2  in place, no crossing        v_mov_b32 D.lo, D.hi ; v_pk_mul_f32 D, S, D
//   mode 3  two scalar multiplies        v_mul_f32 D.lo, S.lo, D.hi ; v_mul_f32 D.hi, S.hi, D.hi
// Every wave also runs what the real kernel runs around it (LDS reads, v_exp, a few MFMAs) so that the two waves of a SIMD contend for
// the same issue ports.  Operands are small integers x powers of two: products are exact, a mismatch is a wrong operand, not rounding.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 pk_opsel_inplace.hip -o pk_opsel_inplace && ./pk_opsel_inplace [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Sample {
    float lo_got, lo_exp, hi_got, hi_exp, s_lo, s_hi, d_lo, d_hi;
};

template <int MODE, bool MFMA>
__global__ __launch_bounds__(256, 2) void probe(unsigned long long* bad, Sample* samples, unsigned* nsamp, int iters, float* sink) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) lds[i] = (float)((i * 37 + 11) % 61 - 30);          // small integers
    __syncthreads();
    f32x16 acc = {0};
    half8 ha, hb;
    for (int k = 0; k < 8; ++k) { ha[k] = (_Float16)(0.01f * (lane + k)); hb[k] = (_Float16)(0.02f * (k + 1)); }
    unsigned long long nbad = 0;
    float keep = 0.f;
    const int skew = (blockIdx.x & 1) * 3 + (tid >> 6);                                       // de-phase the co-resident waves
    for (int it = 0; it < iters; ++it) {
        const float gx = lds[(tid * 7 + it * 5 + skew) & 2047], gy = lds[(tid * 13 + it * 3 + 1) & 2047];
        const float e = __builtin_amdgcn_exp2f(-(float)((it + lane + skew) & 7));             // 2^-k, exact
        const float x0 = lds[(tid + it) & 2047];
        f2 S = {gx * gx, gy};
        f2 D = {x0, e};
        const float s_lo = S.x, s_hi = S.y, d_lo = D.x, d_hi = D.y;
        float exp_lo, exp_hi;
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(exp_lo), "=&v"(exp_hi) : "v"(s_lo), "v"(s_hi), "v"(d_hi));
        if (MODE == 0) {
            asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %0 op_sel:[0,1]\n\ts_nop 0" : "+v"(D) : "v"(S));
        } else if (MODE == 1) {
            f2 T;
            asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 0" : "=&v"(T) : "v"(S), "v"(D));
            D = T;
        } else if (MODE == 2) {
            D.x = D.y;
            asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %0\n\ts_nop 0" : "+v"(D) : "v"(S));
        } else {
            float a, b;
            asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(a), "=&v"(b) : "v"(S.x), "v"(S.y), "v"(D.y));
            D = f2{a, b};
        }
        const bool wrong = D.x != exp_lo || D.y != exp_hi;
        if (wrong) {
            ++nbad;
            const unsigned k = atomicAdd(nsamp, 1u);
            if (k < 64) samples[k] = Sample{D.x, exp_lo, D.y, exp_hi, s_lo, s_hi, d_lo, d_hi};
        }
        keep += D.x + D.y;
        if (MFMA) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    float s = keep;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 123456.789f) sink[0] = s;
}

template <int MODE, bool MFMA>
void run(const char* what, int iters) {
    unsigned long long* bad;
    Sample* samples;
    unsigned* nsamp;
    float* sink;
    hipMalloc(&bad, 8); hipMalloc(&samples, 64 * sizeof(Sample)); hipMalloc(&nsamp, 4); hipMalloc(&sink, 4);
    hipMemset(bad, 0, 8); hipMemset(nsamp, 0, 4);
    const int smem = 79 * 1024;                                    // two workgroups per CU, like fine_match
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int grid = 512 * 4;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, MFMA>), dim3(grid), dim3(256), smem, 0, bad, samples, nsamp, iters, sink);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long hb = 0;
    unsigned hn = 0;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hn, nsamp, 4, hipMemcpyDeviceToHost);
    const double execs = (double)grid * 256 * iters;
    printf("%-52s %s: %llu wrong of %.3g lane-executions (%.2e), %.1f ms\n", what, MFMA ? "beside MFMAs" : "VALU / LDS only", hb, execs,
           hb / execs, ms);
    if (hn) {
        std::vector<Sample> hs(64);
        hipMemcpy(hs.data(), samples, 64 * sizeof(Sample), hipMemcpyDeviceToHost);
        for (unsigned i = 0; i < hn && i < 6; ++i) {
            const Sample& s = hs[i];
            const char* kind = s.lo_got == s.s_lo * s.hi_got ? "lo = S.lo * NEW D.hi (read after the high half was written)"
                               : s.lo_got == s.s_lo * s.d_lo ? "lo = S.lo * D.lo (op_sel ignored)"
                               : s.hi_got != s.hi_exp        ? "high half wrong"
                                                             : "other";
            printf("    got (%g, %g) expected (%g, %g); S = (%g, %g), D = (%g, %g): %s\n", s.lo_got, s.hi_got, s.lo_exp, s.hi_exp, s.s_lo,
                   s.s_hi, s.d_lo, s.d_hi, kind);
        }
    }
    hipFree(bad); hipFree(samples); hipFree(nsamp); hipFree(sink);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    run<0, true>("mode 0: in place + op_sel crossing", iters);
    run<0, false>("mode 0: in place + op_sel crossing", iters);
    run<1, true>("mode 1: crossing, destination != source", iters);
    run<2, true>("mode 2: in place, no crossing (v_mov first)", iters);
    run<3, true>("mode 3: two scalar multiplies", iters);
    return 0;
}
