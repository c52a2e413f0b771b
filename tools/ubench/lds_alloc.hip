// What LDS range does the hardware give two co-resident workgroups that each ask for ~79 KB of dynamic LDS?
// Every workgroup records HW_REG_LDS_ALLOC (base / size granules) and HW_REG_HW_ID (CU / SE / XCC) of its first wave, then checks
// that LDS it wrote at its START and at its END is still intact after a delay in which the co-resident workgroup writes its own.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_alloc.hip -o tools/ubench/lds_alloc && ./lds_alloc [bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* w = reinterpret_cast<unsigned*>(smem);
    const int n = bytes / 4;
    const unsigned tag = 0x10000u + blockIdx.x;
    for (int i = threadIdx.x; i < n; i += 256) w[i] = tag ^ (unsigned)i;
    __syncthreads();
    for (int k = 0; k < 200; ++k) __builtin_amdgcn_s_sleep(127);          // let the neighbour write its region
    __syncthreads();
    int bad = 0;
    for (int i = threadIdx.x; i < n; i += 256) bad += (w[i] != (tag ^ (unsigned)i));
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    atomicAdd(&s_bad, bad);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg((6) | (0 << 6) | (31 << 11));     // HW_REG_LDS_ALLOC, all 32 bits
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
        out[blockIdx.x * 4 + 2] = (unsigned)s_bad;
        out[blockIdx.x * 4 + 3] = (unsigned)(__builtin_amdgcn_s_memtime() >> 10);
    }
}

int main(int argc, char** argv) {
    const int bytes = argc > 1 ? atoi(argv[1]) : 80896;
    const int nblk = 1024;
    unsigned* d;
    hipMalloc(&d, nblk * 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), bytes, 0, d, bytes);
    std::vector<unsigned> h(nblk * 4);
    hipMemcpy(h.data(), d, nblk * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> allocs;
    int bad_blocks = 0;
    for (int b = 0; b < nblk; ++b) {
        allocs[h[b * 4]]++;
        bad_blocks += h[b * 4 + 2] != 0;
    }
    printf("dynamic LDS %d B per workgroup, %d workgroups: %d saw their LDS overwritten\n", bytes, nblk, bad_blocks);
    for (auto& kv : allocs)
        printf("  HW_REG_LDS_ALLOC 0x%08x: base field %u, size field %u  x %d workgroups\n", kv.first, kv.first & 0xff, (kv.first >> 12) & 0x1ff, kv.second);
    return 0;
}
