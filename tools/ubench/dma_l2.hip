// Micro-benchmark: L2-resident global -> LDS (buffer_load ... lds, 16 B/lane) and global -> VGPR throughput per CU
// as a function of the access shape of one wave instruction.  Build: hipcc --offload-arch=gfx950 -O3 dma_l2.hip -o dma_l2
//   shape 0: 1 KB contiguous                     (8 full 128-B lines)
//   shape 1: 16 rows x 64 B,  row stride RS      (the conv kernels' K=32 fp16 slab rows)
//   shape 2:  8 rows x 128 B, row stride RS      (K=64 fp16 rows)
//   shape 3:  4 rows x 256 B, row stride RS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int SHAPE, int DEPTH, bool TO_LDS>
__global__ __launch_bounds__(512, 1) void stream_kernel(const char* __restrict__ src, unsigned bytes, int iters,
                                                         int row_stride, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    constexpr int ROWS = SHAPE == 0 ? 1 : SHAPE == 1 ? 16 : SHAPE == 2 ? 8 : 4;
    constexpr int LPR = 64 / ROWS;                                // lanes per row
    const unsigned lane_off = SHAPE == 0 ? lane * 16u : (unsigned)(lane / LPR) * row_stride + (lane % LPR) * 16u;
    const unsigned piece_stride = SHAPE == 0 ? 1024u : (unsigned)ROWS * row_stride;
    // every wave walks its own region, wrapping inside the (L2-resident) buffer
    unsigned pos = (unsigned)((blockIdx.x * 8 + wave) * 7919u) % (bytes / piece_stride);
    const unsigned npieces = bytes / piece_stride - 1;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned off = pos * piece_stride + lane_off + (SHAPE == 0 ? 0u : ((it & 3) * (LPR * 16u)) % row_stride);
            if (TO_LDS) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (wave * DEPTH + d) * 1024), 16, off, 0, 0, 0);
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + off);
                acc += v;
            }
            pos = pos + 1 >= npieces ? 0 : pos + 1;
        }
        if (TO_LDS) wait_vmcnt<DEPTH / 2>();
    }
    if (TO_LDS) {
        wait_vmcnt<0>();
        __syncthreads();
        acc[0] = *reinterpret_cast<float*>(smem + threadIdx.x * 4);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int SHAPE, int DEPTH, bool TO_LDS>
void run(const char* name, const char* src, unsigned bytes, int row_stride, float* sink, int nwg) {
    const int iters = 2000 / DEPTH * 8;
    auto k = stream_kernel<SHAPE, DEPTH, TO_LDS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 100 * 1024, 0, src, bytes, iters, row_stride, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 100 * 1024, 0, src, bytes, iters, row_stride, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double tot = (double)nwg * 8 * iters * DEPTH * 1024.0;
    printf("%-44s stride %5d depth %2d  %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip\n", name, row_stride, DEPTH, ms,
           tot / ms / 1e6 / nwg, tot / ms / 1e9);
}

int main() {
    const unsigned bytes = 2u << 20;                              // 2 MiB: resident in every XCD's 4 MiB L2
    char* src; float* sink;
    (void)hipMalloc(&src, bytes); (void)hipMemset(src, 1, bytes); (void)hipMalloc(&sink, 64);
    const int nwg = 256;
    run<0, 8, true>("lds-dma contiguous 1 KB", src, bytes, 0, sink, nwg);
    run<0, 16, true>("lds-dma contiguous 1 KB", src, bytes, 0, sink, nwg);
    for (int rs : {256, 512, 2304, 4608}) {
        run<1, 8, true>("lds-dma 16 rows x 64 B", src, bytes, rs, sink, nwg);
        run<2, 8, true>("lds-dma  8 rows x 128 B", src, bytes, rs, sink, nwg);
        run<3, 8, true>("lds-dma  4 rows x 256 B", src, bytes, rs, sink, nwg);
    }
    run<1, 16, true>("lds-dma 16 rows x 64 B", src, bytes, 256, sink, nwg);
    run<2, 16, true>("lds-dma  8 rows x 128 B", src, bytes, 256, sink, nwg);
    run<0, 8, false>("global_load->vgpr contiguous 1 KB", src, bytes, 0, sink, nwg);
    run<1, 8, false>("global_load->vgpr 16 rows x 64 B", src, bytes, 256, sink, nwg);
    run<2, 8, false>("global_load->vgpr  8 rows x 128 B", src, bytes, 256, sink, nwg);
    run<1, 8, false>("global_load->vgpr 16 rows x 64 B", src, bytes, 2304, sink, nwg);
    return 0;
}
