// Reproducer for the fine_match SLP-build deviations (r06; tools/studies/fine_bisect.py found the instruction):
//
//     v_pk_mul_f32 v[152:153], v[202:203], v[152:153] op_sel:[0,1]
//
// -- a packed fp32 multiply whose LOW result takes the HIGH half of a source pair (an op_sel bit is set).  With one wave per SIMD the
// kernel that contains it is bit-reproducible; with two workgroups per CU ~3e-4 of its executions deliver a wrong low half.  This file
// runs that instruction class in isolation, self-checking, two waves per SIMD, and maps what triggers it:
//   form   which packed instruction / modifier (A is the culprit of fine_match; C is the broadcast form SLP code is full of)
//   env    own   = the wave itself has MFMAs in flight (4 x v_mfma_f32_32x32x16_f16 per iteration, as fine_match has between its
//                  softmax updates);  none = no MFMA anywhere;  partner = only the OTHER workgroup of the CU runs MFMAs
//   pos    far = ~40 VALU / LDS instructions after the wave's last MFMA was issued; near = directly behind it
// Operands are small integers x powers of two: every product is exact, a mismatch is a wrong operand, not rounding.
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

enum Form { A_MUL_SRC1HI_TO_LO, A_INPLACE, B_MUL_SRC0HI_TO_LO, C_MUL_SRC1LO_TO_HI, D_ADD_SRC1HI_TO_LO, E_FMA_SRC1HI_TO_LO, F_NOSEL_AFTER_MOV,
            G_TWO_SCALAR, H_FMA_BROADCAST_SRC0, P_PLAIN_FMA };
enum Env { OWN, NONE, PARTNER };

template <int FORM>
__device__ __forceinline__ void apply(f2& D, const f2 S, float& exp_lo, float& exp_hi) {
    // expected values by scalar instructions the compiler cannot fold with the packed one
    const float sl = S.x, sh = S.y, dl = D.x, dh = D.y;
    if (FORM == A_MUL_SRC1HI_TO_LO || FORM == A_INPLACE || FORM == F_NOSEL_AFTER_MOV || FORM == G_TWO_SCALAR) {
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(exp_lo), "=&v"(exp_hi) : "v"(sl), "v"(sh), "v"(dh));
    } else if (FORM == B_MUL_SRC0HI_TO_LO) {
        asm volatile("v_mul_f32 %0, %3, %4\n\tv_mul_f32 %1, %3, %5" : "=&v"(exp_lo), "=&v"(exp_hi) : "v"(sl), "v"(sh), "v"(dl), "v"(dh));
    } else if (FORM == C_MUL_SRC1LO_TO_HI) {
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(exp_lo), "=&v"(exp_hi) : "v"(sl), "v"(sh), "v"(dl));
    } else if (FORM == D_ADD_SRC1HI_TO_LO) {
        asm volatile("v_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %4" : "=&v"(exp_lo), "=&v"(exp_hi) : "v"(sl), "v"(sh), "v"(dh));
    } else if (FORM == H_FMA_BROADCAST_SRC0) {      // fma(D.lo (both lanes), S, D): direct_conv's / SLP code's broadcast form
        asm volatile("v_fma_f32 %0, %4, %2, %4\n\tv_fma_f32 %1, %4, %3, %5" : "=&v"(exp_lo), "=&v"(exp_hi) : "v"(sl), "v"(sh), "v"(dl), "v"(dh));
    } else if (FORM == P_PLAIN_FMA) {               // fma(S, D, S), no modifier
        asm volatile("v_fma_f32 %0, %2, %4, %2\n\tv_fma_f32 %1, %3, %5, %3" : "=&v"(exp_lo), "=&v"(exp_hi) : "v"(sl), "v"(sh), "v"(dl), "v"(dh));
    } else {      // E: fma(S, D.hi-crossed, S)
        asm volatile("v_fma_f32 %0, %2, %4, %2\n\tv_fma_f32 %1, %3, %4, %3" : "=&v"(exp_lo), "=&v"(exp_hi) : "v"(sl), "v"(sh), "v"(dh));
    }
    f2 T;
    if (FORM == A_MUL_SRC1HI_TO_LO) {
        asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 0" : "=&v"(T) : "v"(S), "v"(D));
        D = T;
    } else if (FORM == A_INPLACE) {
        asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %0 op_sel:[0,1]\n\ts_nop 0" : "+v"(D) : "v"(S));
    } else if (FORM == B_MUL_SRC0HI_TO_LO) {
        asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[1,0]\n\ts_nop 0" : "=&v"(T) : "v"(S), "v"(D));
        D = T;
    } else if (FORM == C_MUL_SRC1LO_TO_HI) {
        asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\ts_nop 0" : "=&v"(T) : "v"(S), "v"(D));
        D = T;
    } else if (FORM == D_ADD_SRC1HI_TO_LO) {
        asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 0" : "=&v"(T) : "v"(S), "v"(D));
        D = T;
    } else if (FORM == E_FMA_SRC1HI_TO_LO) {
        asm volatile("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]\n\ts_nop 0" : "=&v"(T) : "v"(S), "v"(D));
        D = T;
    } else if (FORM == H_FMA_BROADCAST_SRC0) {
        asm volatile("s_nop 0\n\tv_pk_fma_f32 %0, %2, %1, %2 op_sel_hi:[0,1,1]\n\ts_nop 0" : "=&v"(T) : "v"(S), "v"(D));
        D = T;
    } else if (FORM == P_PLAIN_FMA) {
        asm volatile("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %1\n\ts_nop 0" : "=&v"(T) : "v"(S), "v"(D));
        D = T;
    } else if (FORM == F_NOSEL_AFTER_MOV) {
        D.x = D.y;
        asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %0\n\ts_nop 0" : "+v"(D) : "v"(S));
    } else {
        float a, b;
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(a), "=&v"(b) : "v"(sl), "v"(sh), "v"(dh));
        D = f2{a, b};
    }
}

template <int FORM, int ENV, bool NEAR>
__global__ __launch_bounds__(256, 2) void probe(unsigned long long* bad, Sample* samples, unsigned* nsamp, int iters, float* sink) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) lds[i] = (float)((i * 37 + 11) % 61 - 30);          // small integers
    __syncthreads();
    f32x16 acc = {0};
    half8 ha, hb;
    for (int k = 0; k < 8; ++k) { ha[k] = (_Float16)(0.01f * (lane + k)); hb[k] = (_Float16)(0.02f * (k + 1)); }
    float keep = 0.f;
    if (ENV == PARTNER && (blockIdx.x & 1)) {          // the partner workgroups: MFMAs only (odd blocks share CUs with even ones)
        for (int it = 0; it < iters * 6; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += acc[r];
        if (s == 123456.789f) sink[0] = s;
        return;
    }
    const bool mfma = ENV == OWN;
    unsigned long long nbad = 0;
    const int skew = (blockIdx.x & 1) * 3 + (tid >> 6);                                       // de-phase the co-resident waves
    f2 Sn, Dn;                                                                                // operands of the NEXT iteration (NEAR)
    auto operands = [&](int it, f2& S, f2& D) __attribute__((always_inline)) {
        const float gx = lds[(tid * 7 + it * 5 + skew) & 2047], gy = lds[(tid * 13 + it * 3 + 1) & 2047];
        const float e = __builtin_amdgcn_exp2f(-(float)((it + lane + skew) & 7));             // 2^-k, exact
        S = f2{gx * gx, gy};
        D = f2{lds[(tid + it) & 2047], e};
    };
    operands(0, Sn, Dn);
    for (int it = 0; it < iters; ++it) {
        f2 S, D;
        if (NEAR) { S = Sn; D = Dn; } else operands(it, S, D);
        const float s_lo = S.x, s_hi = S.y, d_lo = D.x, d_hi = D.y;
        float exp_lo, exp_hi;
        if (NEAR && mfma) {                             // the packed op directly behind the MFMAs
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
            asm volatile("" : "+v"(S), "+v"(D));
        }
        apply<FORM>(D, S, exp_lo, exp_hi);
        if (D.x != exp_lo || D.y != exp_hi) {
            ++nbad;
            const unsigned k = atomicAdd(nsamp, 1u);
            if (k < 64) samples[k] = Sample{D.x, exp_lo, D.y, exp_hi, s_lo, s_hi, d_lo, d_hi};
        }
        keep += D.x + D.y;
        if (NEAR) operands(it + 1, Sn, Dn);
        if (!NEAR && mfma) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    float s = keep;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 123456.789f) sink[0] = s;
}

template <int FORM, int ENV, bool NEAR>
void run(const char* what, int iters) {
    unsigned long long* bad;
    Sample* samples;
    unsigned* nsamp;
    float* sink;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&samples, 64 * sizeof(Sample)); (void)hipMalloc(&nsamp, 4); (void)hipMalloc(&sink, 4);
    (void)hipMemset(bad, 0, 8); (void)hipMemset(nsamp, 0, 4);
    const int smem = 79 * 1024;                                    // two workgroups per CU, like fine_match
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<FORM, ENV, NEAR>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int grid = 512 * 4;
    hipLaunchKernelGGL((probe<FORM, ENV, NEAR>), dim3(grid), dim3(256), smem, 0, bad, samples, nsamp, iters, sink);
    (void)hipDeviceSynchronize();
    unsigned long long hb = 0;
    unsigned hn = 0;
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&hn, nsamp, 4, hipMemcpyDeviceToHost);
    const double execs = (double)(ENV == PARTNER ? grid / 2 : grid) * 256 * iters;
    printf("%-62s MFMAs: %-8s %-5s %12llu wrong of %.3g lane-executions (%.2e)\n", what, ENV == OWN ? "own wave" : ENV == NONE ? "none" : "partner",
           NEAR ? "near" : "far", hb, execs, hb / execs);
    if (hn) {
        std::vector<Sample> hs(64);
        (void)hipMemcpy(hs.data(), samples, 64 * sizeof(Sample), hipMemcpyDeviceToHost);
        for (unsigned i = 0; i < hn && i < 3; ++i) {
            const Sample& s = hs[i];
            printf("      got (%g, %g) expected (%g, %g); S = (%g, %g), D = (%g, %g)%s\n", s.lo_got, s.hi_got, s.lo_exp, s.hi_exp, s.s_lo, s.s_hi,
                   s.d_lo, s.d_hi, s.lo_got == 0.f && s.hi_got == s.hi_exp ? ": the LOW result is 0, the high one right" : "");
        }
    }
    (void)hipFree(bad); (void)hipFree(samples); (void)hipFree(nsamp); (void)hipFree(sink);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    run<A_MUL_SRC1HI_TO_LO, OWN, false>("A  v_pk_mul_f32 T, S, D op_sel:[0,1]   (lo = S.lo * D.HI)", iters);
    run<A_MUL_SRC1HI_TO_LO, OWN, true>("A  v_pk_mul_f32 T, S, D op_sel:[0,1]", iters);
    run<A_MUL_SRC1HI_TO_LO, NONE, false>("A  v_pk_mul_f32 T, S, D op_sel:[0,1]", iters);
    run<A_MUL_SRC1HI_TO_LO, PARTNER, false>("A  v_pk_mul_f32 T, S, D op_sel:[0,1]", iters);
    run<A_INPLACE, OWN, false>("A' v_pk_mul_f32 D, S, D op_sel:[0,1]   (in place, as in fine_match)", iters);
    run<A_INPLACE, OWN, true>("A' v_pk_mul_f32 D, S, D op_sel:[0,1]", iters);
    run<A_INPLACE, PARTNER, false>("A' v_pk_mul_f32 D, S, D op_sel:[0,1]", iters);
    run<B_MUL_SRC0HI_TO_LO, OWN, false>("B  v_pk_mul_f32 T, S, D op_sel:[1,0]   (lo = S.HI * D.lo)", iters);
    run<B_MUL_SRC0HI_TO_LO, PARTNER, false>("B  v_pk_mul_f32 T, S, D op_sel:[1,0]", iters);
    run<C_MUL_SRC1LO_TO_HI, OWN, false>("C  v_pk_mul_f32 T, S, D op_sel_hi:[1,0] (hi = S.hi * D.LO: the broadcast form)", iters);
    run<C_MUL_SRC1LO_TO_HI, OWN, true>("C  v_pk_mul_f32 T, S, D op_sel_hi:[1,0]", iters);
    run<C_MUL_SRC1LO_TO_HI, PARTNER, false>("C  v_pk_mul_f32 T, S, D op_sel_hi:[1,0]", iters);
    run<D_ADD_SRC1HI_TO_LO, OWN, false>("D  v_pk_add_f32 T, S, D op_sel:[0,1]", iters);
    run<E_FMA_SRC1HI_TO_LO, OWN, false>("E  v_pk_fma_f32 T, S, D, S op_sel:[0,1,0]", iters);
    run<F_NOSEL_AFTER_MOV, OWN, false>("F  v_mov_b32 D.lo, D.hi ; v_pk_mul_f32 D, S, D   (no modifier)", iters);
    run<F_NOSEL_AFTER_MOV, OWN, true>("F  v_mov_b32 D.lo, D.hi ; v_pk_mul_f32 D, S, D", iters);
    run<G_TWO_SCALAR, OWN, false>("G  v_mul_f32 x 2", iters);
    // the forms the library's kernels DO contain beside MFMAs (encoder_fused / encoder256: plain and broadcast), 25 x as long
    run<P_PLAIN_FMA, OWN, false>("P  v_pk_fma_f32 T, S, D, S                 (no modifier), long run", iters * 25);
    run<P_PLAIN_FMA, OWN, true>("P  v_pk_fma_f32 T, S, D, S, long run", iters * 25);
    run<H_FMA_BROADCAST_SRC0, OWN, false>("H  v_pk_fma_f32 T, D, S, D op_sel_hi:[0,1,1] (D.lo in both lanes), long run", iters * 25);
    run<H_FMA_BROADCAST_SRC0, OWN, true>("H  v_pk_fma_f32 T, D, S, D op_sel_hi:[0,1,1], long run", iters * 25);
    run<C_MUL_SRC1LO_TO_HI, OWN, false>("C  v_pk_mul_f32 T, S, D op_sel_hi:[1,0], long run", iters * 25);
    run<B_MUL_SRC0HI_TO_LO, OWN, false>("B  v_pk_mul_f32 T, S, D op_sel:[1,0], long run", iters * 25);
    return 0;
}
