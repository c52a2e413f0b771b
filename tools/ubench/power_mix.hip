// Micro-benchmark 4: how much MFMA throughput do the other activities of the conv main loop cost under the
// power cap?  Random fp16 operands everywhere.  Per "slab" a wave runs 24 v_mfma_f32_32x32x16_f16 and optionally
//   R: 16 ds_read_b128 fragment reads (their data feed the MFMAs),
//   D: P 1-KB global->LDS DMA pieces (16 rows x 64 B from an L2-resident buffer) with a 2-slab lead and a barrier.
// 8 waves per CU, 256 CUs, like the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int P, bool READS>
__global__ __launch_bounds__(512, 1) void k(const char* __restrict__ src, unsigned bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    // fill this wave's 16 KB of LDS with random halves
    for (int i = 0; i < 16; ++i)
        *reinterpret_cast<half8*>(smem + (wave * 16 + i) * 1024 + lane * 16) =
            *reinterpret_cast<const half8*>(src + ((blockIdx.x * 128 + wave * 16 + i) * 1024 + lane * 16) % bytes);
    __syncthreads();
    const unsigned lane_off = (unsigned)(lane >> 2) * 256u + (lane & 3) * 16u;
    const unsigned npieces = bytes / 4096 - 1;
    unsigned pos = (unsigned)((blockIdx.x * 8 + wave) * 7919u) % npieces;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x16{0};
    half8 f[16];
    for (int i = 0; i < 16; ++i) f[i] = *reinterpret_cast<const half8*>(smem + (wave * 16 + i) * 1024 + lane * 16);
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + 131072 + (wave * 3 * (P ? P : 1) + slot) * 1024), 16,
                                                     pos * 4096 + lane_off + (it & 3) * 64, 0, 0, 0);
            pos = pos + 1 >= npieces ? 0 : pos + 1;
            slot = slot + 1 >= 3 * P ? 0 : slot + 1;
        }
        if (P) { wait_vmcnt<2 * P>(); __builtin_amdgcn_s_barrier(); }
        if (READS) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                f[i] = *reinterpret_cast<const half8*>(smem + ((wave * 16 + ((i + it) & 15)) * 1024 + lane * 16));
        }
#pragma unroll
        for (int m = 0; m < 24; ++m)
            acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[m & 7], f[8 + ((m >> 1) & 7)], acc[m & 7], 0, 0, 0);
    }
    wait_vmcnt<0>();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][9];
    if (s == 12345.678f) sink[0] = s;
}

template <int P, bool READS>
void run(const char* name, const char* src, unsigned bytes, float* sink) {
    const int iters = 4000;
    auto kk = k<P, READS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kk, dim3(256), dim3(512), 160 * 1024, 0, src, bytes, iters, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kk, dim3(256), dim3(512), 160 * 1024, 0, src, bytes, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * iters * 24 * 2.0 * 32 * 32 * 16;
    printf("%-46s %8.3f ms  %7.1f TFLOP/s MFMA   %5.3f us per slab\n", name, ms, flops / ms / 1e9, ms * 1e3 / iters);
}

int main() {
    const unsigned bytes = 2u << 20;
    std::vector<_Float16> h(bytes / 2);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX + rand() / (float)RAND_MAX + rand() / (float)RAND_MAX - 1.5f) * 2.f);
    char* src; float* sink;
    (void)hipMalloc(&src, bytes); (void)hipMalloc(&sink, 64);
    (void)hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, false>("MFMA only (random operands)", src, bytes, sink);
        run<0, true>("MFMA + 16 fragment reads / slab", src, bytes, sink);
        run<4, false>("MFMA + 4 KB/wave DMA / slab", src, bytes, sink);
        run<4, true>("MFMA + reads + DMA (the conv loop's mix)", src, bytes, sink);
        run<2, true>("MFMA + reads + 2 KB/wave DMA", src, bytes, sink);
    }
    return 0;
}
