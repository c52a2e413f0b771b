// Residency census of s2d_front_kernel (csrc/s2d_front.hip compiled into this file with -DS2D_CENSUS): what the occupancy API says,
// and which workgroups were actually alive together on a CU (HW_ID + start / end stamps of every workgroup).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../include -I../../detectorfreesfm_amd/csrc s2d_census.hip -o s2d_census
#define S2D_CENSUS 1
#include "../../detectorfreesfm_amd/csrc/s2d_front.hip"
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
namespace dfsfm { void set_last_error(const char*, hipError_t) {} }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2048;
    int nb = -1;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&s2d_front_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, s2d_front_kernel, 256, SMEM);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&s2d_front_kernel));
    printf("dynamic LDS %d B; occupancy API: %d workgroups per CU; numRegs %d, static LDS %zu, maxDynamicShared %d\n", SMEM, nb, fa.numRegs,
           fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes);
    float *x, *w1g, *b1, *b2;
    _Float16 *w2h, *w2l, *crh, *crl, *poh, *pol;
    unsigned long long* census;
    (void)hipMalloc(&x, (size_t)n * 35 * 35 * 3 * 4); (void)hipMemset(x, 0, (size_t)n * 35 * 35 * 3 * 4);
    (void)hipMalloc(&w1g, 8 * 216 * 4); (void)hipMemset(w1g, 0, 8 * 216 * 4);
    (void)hipMalloc(&b1, 256); (void)hipMemset(b1, 0, 256);
    (void)hipMalloc(&b2, 256); (void)hipMemset(b2, 0, 256);
    (void)hipMalloc(&w2h, 128 * 576 * 2); (void)hipMemset(w2h, 0, 128 * 576 * 2);
    (void)hipMalloc(&w2l, 128 * 576 * 2); (void)hipMemset(w2l, 0, 128 * 576 * 2);
    (void)hipMalloc(&crh, (size_t)n * 19 * 19 * 64 * 2); (void)hipMalloc(&crl, (size_t)n * 19 * 19 * 64 * 2);
    (void)hipMalloc(&poh, (size_t)n * 18 * 18 * 64 * 2); (void)hipMalloc(&pol, (size_t)n * 18 * 18 * 64 * 2);
    (void)hipMalloc(&census, (size_t)n * 32);
    FrontArgs g{};
    g.x = x; g.w1g = w1g; g.b1 = b1; g.w2h = w2h; g.w2l = w2l; g.b2 = b2; g.crh = crh; g.crl = crl; g.poh = poh; g.pol = pol;
    g.c0 = 8; g.c1 = 27; g.kpad = 576; g.w2bytes = 64 * 576 * 2; g.first_gen = 0; g.skew = 0; g.census = census;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(s2d_front_kernel, dim3(n), dim3(256), SMEM, 0, g);
        (void)hipDeviceSynchronize();
    }
    printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
    std::vector<unsigned long long> h((size_t)n * 4);
    (void)hipMemcpy(h.data(), census, (size_t)n * 32, hipMemcpyDeviceToHost);
    // per (XCC / SE / CU): maximum number of workgroups alive at once
    std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
    std::map<unsigned long long, int> lds;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = (unsigned)h[i * 4];
        const unsigned key = hw & 0xFFFFFF00u & ~0x3Fu;           // drop wave / SIMD id bits (kept coarse: CU + SE + ...)
        ev[hw >> 8].push_back({h[i * 4 + 2], +1});
        ev[hw >> 8].push_back({h[i * 4 + 3], -1});
        lds[h[i * 4 + 1]]++;
        (void)key;
    }
    std::map<int, int> hist;
    for (auto& kv : ev) {
        auto v = kv.second;
        std::sort(v.begin(), v.end());
        int cur = 0, mx = 0;
        for (auto& e : v) { cur += e.second; mx = std::max(mx, cur); }
        hist[mx]++;
    }
    printf("%zu distinct HW_ID[31:8] groups; max workgroups alive at once per group: ", ev.size());
    for (auto& kv : hist) printf("%d x%d  ", kv.first, kv.second);
    printf("\nHW_REG_LDS_ALLOC values: ");
    for (auto& kv : lds) printf("0x%08llx x%d  ", kv.first, kv.second);
    unsigned long long dmin = ~0ull, dmax = 0;
    for (int i = 0; i < n; ++i) { const unsigned long long d = h[i * 4 + 3] - h[i * 4 + 2]; dmin = std::min(dmin, d); dmax = std::max(dmax, d); }
    printf("\nworkgroup duration (s_memtime ticks): min %llu max %llu\n", dmin, dmax);
    return 0;
}
