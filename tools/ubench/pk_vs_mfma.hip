// pk_vs_mfma -- does a packed-fp32 VALU instruction (v_pk_fma_f32) return wrong results while ANOTHER wave of the same SIMD has MFMAs
// in flight?  (r04 found that fine_match.hip's run-to-run deviations with two co-resident workgroups vanish when the
// SLP vectoriser's packed-fp32 instructions do, and concluded exactly that -- from a correlation.  This is the minimal reproducer.)
//
// Every experiment is SELF-CHECKING and EXACT: all operands are small integers held in fp32 / fp16, every product and every partial sum
// is an integer below 2^24, so any evaluation order gives the same bits and the expected value is tracked in the integer ALU of the
// same lane.  A mismatch is a hardware (or hazard) fault, never rounding.
//
//   mode A  "alone"     : 256-thread workgroups, one per CU (LDS-padded), every wave loops v_pk_fma_f32 on register data.
//   mode B  "beside"    : 512-thread workgroups = two waves per SIMD (wave w and w + 4 share SIMD w % 4): waves 0-3 loop v_pk_fma_f32,
//                         waves 4-7 loop back-to-back v_mfma_f32_32x32x16_f16 on random-ish operands (results discarded).
//   mode C  "scalar"    : as B with v_fma_f32 pairs instead of the packed instruction (control).
//   mode D  "consumer"  : fine_match's shape -- every wave runs MFMA -> (wait states) -> packed-fp32 second-moment updates that READ the
//                         MFMA result, two workgroups co-resident per CU (256 threads, small LDS), i.e. beside the other workgroup's
//                         MFMAs; the MFMA result itself is checked as well (D[i][j] = 16 a(i) b(j) by construction).
//   mode E  "consumer1" : as D with ONE workgroup per CU (LDS-padded): the control r04 found bit-exact.
//   mode F  "consumer-s": as D with scalar v_fma_f32 consumers.
// Each mode runs ITER iterations per lane on every CU and reports the number of lanes whose final (or any intermediate) value differed.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/pk_vs_mfma.hip -o tools/ubench/pk_vs_mfma
// run  : tools/ubench/pk_vs_mfma [iterations per lane, default 1000000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

enum { MODE_ALONE = 0, MODE_BESIDE = 1, MODE_SCALAR = 2, MODE_CONS2 = 3, MODE_CONS1 = 4, MODE_CONS_S = 5 };

// acc += x * y, packed (one v_pk_fma_f32 on a 64-bit register pair) or as two scalar FMAs
__device__ __forceinline__ void fma_pk(f2& acc, f2 x, f2 y) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y)); }
__device__ __forceinline__ void fma_sc(f2& acc, f2 x, f2 y) {
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc.x) : "v"(x.x), "v"(y.x));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc.y) : "v"(x.y), "v"(y.y));
}

// register-only packed / scalar loop; returns the number of check points (every 4096 iterations) at which a lane was wrong
template <bool PACKED>
__device__ __forceinline__ unsigned valu_loop(int iters, int lane) {
    unsigned bad = 0;
    f2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    int exp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int xa = (it & 3) + 1, xb = ((it >> 2) & 3) + 1, ya = (lane & 7) + 1, yb = ((lane >> 3) & 7) + 1;     // 1 .. 8
        const f2 x = {(float)xa, (float)xb}, y = {(float)ya, (float)yb};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f2 yk = {y.x + (float)k, y.y + (float)(3 - k)};
            if (PACKED) fma_pk(acc[k], x, yk); else fma_sc(acc[k], x, yk);
            exp_[2 * k] += xa * (ya + k);
            exp_[2 * k + 1] += xb * (yb + 3 - k);
        }
        if ((it & 4095) == 4095) {                    // sums stay below 4096 * 8 * 11 < 2^24: exact; check and restart
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) ok = ok && acc[k].x == (float)exp_[2 * k] && acc[k].y == (float)exp_[2 * k + 1];
            bad += ok ? 0u : 1u;
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[k] = f2{0.f, 0.f}; exp_[2 * k] = exp_[2 * k + 1] = 0; }
        }
    }
    return bad;
}

// back-to-back MFMAs on changing operands (the partner wave of modes B / C); the result feeds a dummy store so nothing is removed
__device__ __forceinline__ float mfma_loop(int iters, int lane) {
    f32x16 c0 = {}, c1 = {};
    half8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.37f * (float)((lane * 7 + k * 3) % 11) - 1.5f); b[k] = (_Float16)(0.21f * (float)((lane * 5 + k) % 13) - 1.1f); }
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        if ((it & 255) == 255) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] *= 1e-3f; c1[r] *= 1e-3f; }        // keep the values finite
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    return s;
}

// fine_match's shape: an MFMA whose result is known exactly, then packed (or scalar) moment updates that read it
template <bool PACKED>
__device__ __forceinline__ unsigned consumer_loop(int iters, int lane, unsigned* bad_mfma) {
    unsigned bad = 0, badm = 0;
    const int col = lane & 31, half = lane >> 5;
    f2 m[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    int exp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        // A[i][k] = alpha(i) for every k, B[k][j] = beta(j): D[i][j] = 16 alpha(i) beta(j).  A operand: lane = row i (lane & 31);
        // B operand: lane = column j; D register r of a lane: row 8 (r / 4) + 4 half + (r % 4), column lane & 31
        const int al = ((col + it) & 3) + 1, be = ((col * 3 + it) & 3) + 1;
        half8 a, b;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(float)al; b[k] = (_Float16)(float)be; }
        f32x16 d = {};
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 7" : "=v"(d) : "v"(a), "v"(b));   // operands settled; 24 wait states after: > the 16-pass rule
        bool okm = true;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + 4 * half + (r & 3);
            okm = okm && d[r] == (float)(16 * (((row + it) & 3) + 1) * be);
        }
        badm += okm ? 0u : 1u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                 // second-moment style updates: m += d * w with small integer weights
            const f2 w = {(float)(k + 1), (float)(4 - k)};
            const f2 dv0 = {d[4 * k], d[4 * k + 1]}, dv1 = {d[4 * k + 2], d[4 * k + 3]};
            if (PACKED) { fma_pk(m[k], dv0, w); fma_pk(m[k], dv1, w); } else { fma_sc(m[k], dv0, w); fma_sc(m[k], dv1, w); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * k + q, row = 8 * (r >> 2) + 4 * half + (r & 3);
                exp_[2 * k + (q & 1)] += 16 * (((row + it) & 3) + 1) * be * ((q & 1) ? (4 - k) : (k + 1));
            }
        }
        if ((it & 1023) == 1023) {                    // 1024 * 2 * 256 * 4 < 2^24
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) ok = ok && m[k].x == (float)exp_[2 * k] && m[k].y == (float)exp_[2 * k + 1];
            bad += ok ? 0u : 1u;
#pragma unroll
            for (int k = 0; k < 4; ++k) { m[k] = f2{0.f, 0.f}; exp_[2 * k] = exp_[2 * k + 1] = 0; }
        }
    }
    *bad_mfma = badm;
    return bad;
}

template <int MODE>
__global__ __launch_bounds__(MODE == MODE_BESIDE || MODE == MODE_SCALAR ? 512 : 256) void kern(int iters, unsigned* bad_valu,
                                                                                                  unsigned* bad_mfma, float* sink) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (smem[tid & 63] == 77) sink[0] = 1.f;          // keep the dynamic LDS allocation alive (it sets the residency)
    unsigned bv = 0, bm = 0;
    if (MODE == MODE_ALONE) {
        bv = valu_loop<true>(iters, lane);
    } else if (MODE == MODE_BESIDE || MODE == MODE_SCALAR) {
        if (wave < 4) bv = MODE == MODE_BESIDE ? valu_loop<true>(iters, lane) : valu_loop<false>(iters, lane);
        else sink[1 + (blockIdx.x * 512 + tid) % 1024] = mfma_loop(iters / 2, lane);      // ~ the same wall time as the VALU waves
    } else {
        bv = MODE == MODE_CONS_S ? consumer_loop<false>(iters, lane, &bm) : consumer_loop<true>(iters, lane, &bm);
    }
    if (bv) atomicAdd(bad_valu, 1u);
    if (bm) atomicAdd(bad_mfma, 1u);
}

template <int MODE>
void run(const char* name, int iters, int threads, int smem, int wgs_per_cu, int cus) {
    unsigned *bv, *bm;
    float* sink;
    CHECK(hipMalloc(&bv, 4)); CHECK(hipMalloc(&bm, 4)); CHECK(hipMalloc(&sink, 4096 * 4));
    CHECK(hipMemset(bv, 0, 4)); CHECK(hipMemset(bm, 0, 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&kern<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern<MODE>, threads, smem));
    hipEvent_t s, e;
    CHECK(hipEventCreate(&s)); CHECK(hipEventCreate(&e));
    CHECK(hipEventRecord(s));
    hipLaunchKernelGGL(kern<MODE>, dim3(cus * wgs_per_cu), dim3(threads), smem, 0, iters, bv, bm, sink);
    CHECK(hipEventRecord(e)); CHECK(hipEventSynchronize(e));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, s, e));
    unsigned hv = 0, hm = 0;
    CHECK(hipMemcpy(&hv, bv, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hm, bm, 4, hipMemcpyDeviceToHost));
    const long lanes = (long)cus * wgs_per_cu * (MODE == MODE_BESIDE || MODE == MODE_SCALAR ? 256 : threads);
    printf("%-44s workgroups/CU (API) %d  lanes checked %8ld x %d iterations  %8.1f ms   lanes with a wrong VALU result: %u   with a wrong MFMA result: %u\n",
           name, occ, lanes, iters, ms, hv, hm);
    CHECK(hipFree(bv)); CHECK(hipFree(bm)); CHECK(hipFree(sink));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000000;
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("pk_vs_mfma: %d CUs, %d iterations per lane; every value is an exact integer < 2^24, expectation tracked in the integer ALU\n", cus, iters);
    for (int rep = 0; rep < 2; ++rep) {
        run<MODE_ALONE>("A packed alone (1 wave/SIMD)", iters, 256, 100 * 1024, 1, cus);
        run<MODE_BESIDE>("B packed beside an MFMA wave (same SIMD)", iters, 512, 100 * 1024, 1, cus);
        run<MODE_SCALAR>("C scalar beside an MFMA wave (same SIMD)", iters, 512, 100 * 1024, 1, cus);
        run<MODE_CONS2>("D MFMA -> packed consumer, 2 workgroups/CU", iters / 4, 256, 60 * 1024, 2, cus);
        run<MODE_CONS1>("E MFMA -> packed consumer, 1 workgroup/CU", iters / 4, 256, 100 * 1024, 1, cus);
        run<MODE_CONS_S>("F MFMA -> scalar consumer, 2 workgroups/CU", iters / 4, 256, 60 * 1024, 2, cus);
    }
    return 0;
}
