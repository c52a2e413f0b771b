// What does a "weight fragments from LDS -> MFMA" stream cost with ONE wave per SIMD (the fused encoder kernels' main loops)?
// 256-thread workgroups, one per CU (LDS-limited); per "slab" a wave reads sixteen 1-KB fragments (lane-linear ds_read_b128, the four
// waves read the SAME addresses, as the kernels do) and issues 24 v_mfma_f32_32x32x16_f16 on 8 accumulators.
//   mode 0: no LDS reads (register operands)                       -- the MFMA floor
//   mode 1: C++ loads, the compiler's schedule (reads of a k-step, counted waits, MFMAs)
//   mode 2: inline-asm reads one k-step ahead of the MFMAs that use them
//   active: number of waves of the workgroup that run the loop (LDS contention)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_lds_stream.hip -o mfma_lds_stream && ./mfma_lds_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(int n, int active, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 2; i += 256) reinterpret_cast<_Float16*>(smem)[i] = (_Float16)(0.001f * ((i * 37) & 255));
    __syncthreads();
    half8 bh, bl;
    for (int k = 0; k < 8; ++k) { bh[k] = (_Float16)(0.002f * (k + 1 + lane)); bl[k] = (_Float16)(0.0001f * (k + 3)); }
    f32x16 am[4], ax[4];
    for (int b = 0; b < 4; ++b) am[b] = ax[b] = f32x16{0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < active) {
        if (MODE == 0) {
            half8 a = bh;
            for (int i = 0; i < n; ++i) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) am[b] = mfma(a, bh, am[b]);
#pragma unroll
                    for (int b = 0; b < 4; ++b) ax[b] = mfma(a, bl, ax[b]);
#pragma unroll
                    for (int b = 0; b < 4; ++b) ax[b] = mfma(bl, bh, ax[b]);
                }
            }
        } else if (MODE == 1) {
            for (int i = 0; i < n; ++i) {
                const char* slab = smem + (i & 3) * 16384;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    half8 wh[4], wl[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        wh[b] = *reinterpret_cast<const half8*>(slab + ((ks * 4 + b) * 2 + 0) * 1024 + lane * 16);
                        wl[b] = *reinterpret_cast<const half8*>(slab + ((ks * 4 + b) * 2 + 1) * 1024 + lane * 16);
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) am[b] = mfma(wh[b], bh, am[b]);
#pragma unroll
                    for (int b = 0; b < 4; ++b) ax[b] = mfma(wl[b], bh, ax[b]);
#pragma unroll
                    for (int b = 0; b < 4; ++b) ax[b] = mfma(wh[b], bl, ax[b]);
                }
                asm volatile("" ::: "memory");
            }
        } else {
            const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + lane * 16;
            u32x4 cur[8], nxt[8];
#define RD(D, A, O) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(D) : "v"(A), "n"(O))
#define WT8(N, A) asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]) : "n"(N))
#pragma unroll
            for (int j = 0; j < 8; ++j) RD(cur[j], base, j * 1024);
            for (int i = 0; i < n; ++i) {
                const unsigned sa = base + (i & 3) * 16384, sn = base + ((i + 1) & 3) * 16384;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (ks == 0) RD(nxt[j], sa, (8 + j) * 1024);
                        else RD(nxt[j], sn, j * 1024);
                    }
                    WT8(8, cur);
#pragma unroll
                    for (int b = 0; b < 4; ++b) am[b] = mfma(__builtin_bit_cast(half8, cur[2 * b]), bh, am[b]);
#pragma unroll
                    for (int b = 0; b < 4; ++b) ax[b] = mfma(__builtin_bit_cast(half8, cur[2 * b + 1]), bh, ax[b]);
#pragma unroll
                    for (int b = 0; b < 4; ++b) ax[b] = mfma(__builtin_bit_cast(half8, cur[2 * b]), bl, ax[b]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
                }
            }
            WT8(0, cur);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float keep = 0.f;
    for (int b = 0; b < 4; ++b) keep += am[b][0] + ax[b][1];
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
    if (keep == 12345.678f) sink[0] = keep;
}

template <int MODE>
static void run(const char* name, int active) {
    unsigned long long* d; float* s;
    hipMalloc(&d, 256 * 4 * 8); hipMalloc(&s, 4);
    const int n = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 100 * 1024, 0, n, active, d, s);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost);
    double sum = 0; int cnt = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < active; ++w) { sum += (double)h[b * 4 + w]; ++cnt; }
    printf("%-58s active waves %d: %.1f ticks per MFMA (%.0f per 24-MFMA slab)\n", name, active, sum / cnt / (n * 24.0), sum / cnt / n);
    hipFree(d); hipFree(s);
}

int main() {
    run<0>("register operands", 4);
    for (int a : {1, 2, 4}) run<1>("fragments from LDS, compiler schedule", a);
    for (int a : {1, 2, 4}) run<2>("fragments from LDS, asm reads one k-step ahead", a);
    return 0;
}
