// Can the VALU work of one wave run under the MFMAs of ANOTHER wave of the same SIMD?  (The question behind every "two workgroups per
// CU hide each other's phases" design in this library: fine_match, linear_gemm_sf, s2d_front.)
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run back-to-back MFMAs, waves 4-7 (their SIMD partners) run a VALU loop.
// Reported: shader cycles (s_memtime) of each loop alone and of both together, per VALU flavour and MFMA shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int VALU_KIND, int MFMA_KIND>      // VALU: 0 v_fma_f32, 1 v_pk_fma_f32, 2 mixed int/cvt ; MFMA: 0 16x16x32, 1 32x32x16
__global__ __launch_bounds__(512, 1) void probe(int run_mfma, int run_valu, int n, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    half8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (lane + k)); b[k] = (_Float16)(0.002f * (k + 1)); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float keep = 0.f;
    if (wave < 4) {
        if (run_mfma) {
            if (MFMA_KIND == 0) {
                f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
                for (int i = 0; i < n; ++i) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
                }
                keep = c0[0] + c1[1] + c2[2] + c3[3];
            } else {
                f32x16 c0 = {0}, c1 = {0};
                for (int i = 0; i < n; ++i) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                }
                keep = c0[0] + c1[1];
            }
        }
    } else if (run_valu) {
        if (VALU_KIND == 0) {
            float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = 1, x5 = 2, x6 = 3, x7 = 4;
            for (int i = 0; i < n; ++i) {
                x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
                x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
                x4 = __builtin_fmaf(x4, 1.0001f, 0.5f); x5 = __builtin_fmaf(x5, 1.0001f, 0.5f);
                x6 = __builtin_fmaf(x6, 1.0001f, 0.5f); x7 = __builtin_fmaf(x7, 1.0001f, 0.5f);
            }
            keep = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        } else if (VALU_KIND == 1) {
            f2 x0 = {(float)lane, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
            const f2 m = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
            for (int i = 0; i < n; ++i) {
                x0 = x0 * m + c; x1 = x1 * m + c; x2 = x2 * m + c; x3 = x3 * m + c;
                x0 = x0 * m + c; x1 = x1 * m + c; x2 = x2 * m + c; x3 = x3 * m + c;
            }
            keep = x0.x + x1.y + x2.x + x3.y;
        } else {
            int y0 = lane, y1 = lane * 3, y2 = 7, y3 = 11;
            float z = lane;
            for (int i = 0; i < n; ++i) {
                y0 = y0 * 5 + 1; y1 = (y1 >> 1) ^ y0; y2 = y2 + (y1 & 15); y3 = max(y3, y2 & 1023);
                z = fmaxf(z * 0.999f, (float)(y3 & 255)); y0 ^= y2; y1 += y3; y2 = (y2 << 1) | (y0 & 1);
            }
            keep = z + (float)(y0 + y1 + y2 + y3);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (keep == 123456.789f) sink[0] = keep;
}

template <int VK, int MK>
void run(const char* what, int n) {
    unsigned long long* out;
    float* sink;
    (void)hipMalloc(&out, 256 * 8 * 8); (void)hipMalloc(&sink, 4);
    double res[3][2];
    for (int mode = 0; mode < 3; ++mode) {
        const int rm = mode != 1, rv = mode != 0;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<VK, MK>), dim3(256), dim3(512), 0, 0, rm, rv, n, out, sink);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        (void)hipMemcpy(h.data(), out, 256 * 8 * 8, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
        res[mode][0] = m / 1024; res[mode][1] = v / 1024;
    }
    const int mf = MK == 0 ? 4 : 2, vi = 8;
    printf("%-44s MFMA alone %.1f ticks/MFMA | VALU alone %.2f ticks/instr | together: MFMA %.1f (x%.2f), VALU %.2f (x%.2f)\n", what,
           res[0][0] / n / mf, res[1][1] / n / vi, res[2][0] / n / mf, res[2][0] / res[0][0], res[2][1] / n / vi, res[2][1] / res[1][1]);
    (void)hipFree(out); (void)hipFree(sink);
}

int main() {
    const int n = 20000;
    run<0, 0>("v_fma_f32 beside 16x16x32 MFMAs", n);
    run<1, 0>("v_pk_fma_f32 beside 16x16x32 MFMAs", n);
    run<2, 0>("integer / compare mix beside 16x16x32 MFMAs", n);
    run<0, 1>("v_fma_f32 beside 32x32x16 MFMAs", n);
    run<1, 1>("v_pk_fma_f32 beside 32x32x16 MFMAs", n);
    run<2, 1>("integer / compare mix beside 32x32x16 MFMAs", n);
    return 0;
}
