// Micro-benchmark 2: the conv kernel's slab loop reduced to its resource skeleton, to find what makes real
// L2 -> LDS traffic expensive next to MFMA work.  One iteration = one "slab": every wave issues P 1-KB LDS-DMA
// pieces (16 rows x 64 B, or 8 rows x 128 B), waits until the pieces of LEAD iterations ago have landed, optionally
// barriers, optionally reads LDS fragments, optionally runs NMFMA 32x32x16 f16 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) void lds_void;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ROWS, int P, int LEAD, bool BAR, int NREAD, int NMFMA>
__global__ __launch_bounds__(512, 1) void slab_kernel(const char* __restrict__ src, unsigned bytes, int iters,
                                                       int row_stride, float* sink, int same_addr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    constexpr int LPR = 64 / ROWS;
    const unsigned lane_off = (unsigned)(lane / LPR) * row_stride + (lane % LPR) * 16u;
    const unsigned piece_stride = (unsigned)ROWS * row_stride;
    const unsigned npieces = bytes / piece_stride - 1;
    // same_addr: every workgroup streams the same addresses in lockstep (like the weight slabs of a GEMM)
    unsigned pos = (unsigned)(((same_addr ? 0 : blockIdx.x) * 8 + wave) * 7919u) % npieces;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x16{0};
    half8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    constexpr int RING = (LEAD + 1) * P;                      // 1-KB slots per wave
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const unsigned off = pos * piece_stride + lane_off + ((it & 3) * (LPR * 16u)) % row_stride;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (wave * RING + slot) * 1024), 16, off, 0, 0, 0);
            pos = pos + 1 >= npieces ? 0 : pos + 1;
            slot = slot + 1 >= RING ? 0 : slot + 1;
        }
        wait_vmcnt<LEAD * P>();
        if (BAR) __builtin_amdgcn_s_barrier();
        // NREAD independent, conflict-free 1-KB fragment reads in batches of 8 (sunk by an empty asm)
#pragma unroll
        for (int r0 = 0; r0 < NREAD; r0 += 8) {
            half8 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r)
                v[r] = *reinterpret_cast<const half8*>(smem + (((wave * 16 + r0 + r) & 127) * 1024 + lane * 16));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("" ::"v"(v[r]));
#endif
        }
#pragma unroll
        for (int m = 0; m < NMFMA; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
    }
    wait_vmcnt<0>();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
    if (s == 12345.678f) sink[0] = s;
}

template <int ROWS, int P, int LEAD, bool BAR, int NREAD, int NMFMA>
void run(const char* src, unsigned bytes, int row_stride, float* sink, int same_addr = 0) {
    const int iters = 2000, nwg = 256;
    auto k = slab_kernel<ROWS, P, LEAD, BAR, NREAD, NMFMA>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 150 * 1024, 0, src, bytes, iters, row_stride, sink, same_addr);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 150 * 1024, 0, src, bytes, iters, row_stride, sink, same_addr);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double tot = (double)nwg * 8 * iters * P * 1024.0;
    printf("%s rows %2d x %3d B  P %d lead %d bar %d reads %2d mfma %2d: %7.3f us/slab  %6.1f GB/s per CU  (mfma floor %5.3f us)\n",
           same_addr ? "same-addr" : "own-addr ", ROWS, 1024 / ROWS, P, LEAD, (int)BAR, NREAD, NMFMA, ms * 1e3 / iters, tot / ms / 1e6 / nwg,
           NMFMA * 2 * 32 / 2.4e3);
}

int main() {
    const unsigned bytes = 2u << 20;
    char* src; float* sink;
    (void)hipMalloc(&src, bytes); (void)hipMemset(src, 0, bytes); (void)hipMalloc(&sink, 64);
    run<16, 4, 2, true, 16, 24>(src, bytes, 256, sink, 0);
    run<16, 4, 2, true, 16, 24>(src, bytes, 256, sink, 1);
    run<16, 4, 2, true, 16, 24>(src, bytes, 2304, sink, 0);
    run<16, 4, 2, true, 16, 24>(src, bytes, 2304, sink, 1);
    run<8, 4, 2, true, 16, 24>(src, bytes, 2304, sink, 1);
    run<16, 4, 2, true, 0, 0>(src, bytes, 2304, sink, 1);
    run<16, 4, 2, true, 0, 0>(src, bytes, 256, sink, 1);
    run<8, 4, 2, true, 0, 0>(src, bytes, 256, sink, 1);
    run<16, 2, 2, true, 16, 24>(src, bytes, 2304, sink, 1);
    return 0;
}
