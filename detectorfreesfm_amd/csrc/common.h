// Shared helpers for the gfx950 kernels of libdfsfm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "dfsfm_hip.h"

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace dfsfm {

// Records the text of a runtime error for dfsfm_last_error_string().
void set_last_error(const char* what, hipError_t e);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(what, e);
        return DFSFM_E_LAUNCH;
    }
    return DFSFM_OK;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// > 64 KiB of dynamic LDS needs the opt-in attribute, per kernel instance AND per device: one static instance of
// this per launch site remembers the devices it has been set on (a process may drive several GPUs).
struct SmemAttr {
    unsigned long long done = 0;
    void ensure(const void* fn, int bytes) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done & bit)) {
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            done |= bit;
        }
    }
};

// C/D fragment row of a 32x32 MFMA accumulator register (MI355X guide, section 3):
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ int mfma32_row(int reg, int half) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

// phi(x) = elu(x) + 1 with torch's arithmetic: expm1(x) for x <= 0, then + 1
// (third_party/LoFTR/src/loftr/loftr_module/linear_attention.py:10-11).
__device__ __forceinline__ float elu_plus_one(float x) {
    return (x > 0.f ? x : expm1f(x)) + 1.f;
}

// phi(x) = elu(x) + 1 = exp(x) for x <= 0, x + 1 otherwise.  The reference evaluates expm1(x) + 1 in fp32; here exp(x) comes
// from the hardware exp2 with the argument's rounding error compensated (x * log2(e) carried as hi + lo), which stays within
// ~2 ulp of it -- below the 2^-22 of the split operands this value is converted to next -- at a fifth of the instructions.
__device__ __forceinline__ float phi_fast(float x) {
    const float L2E = 1.4426950408889634f;
    const float hi = x * L2E;
    const float lo = __builtin_fmaf(x, L2E, -hi) + x * 1.925963033500853e-8f;      // rounding of x * L2E + (log2(e) - L2E)
    const float e = __builtin_amdgcn_exp2f(hi);
    const float r = __builtin_fmaf(e * 0.6931471805599453f, lo, e);                 // 2^(hi + lo) ~ 2^hi (1 + lo ln 2)
    return x > 0.f ? x + 1.f : r;
}

// exp(x) for x <= 0 (-inf allowed: 0) on the same compensated exp2: ~2 ulp at a fifth of expf's instructions.  exp(-104) is
// below the smallest fp32 denormal, so the clamp changes no result and keeps -inf out of the compensation term.
__device__ __forceinline__ float exp_neg(float x) {
    const float L2E = 1.4426950408889634f;
    x = fmaxf(x, -104.f);
    const float hi = x * L2E;
    const float lo = __builtin_fmaf(x, L2E, -hi) + x * 1.925963033500853e-8f;
    const float e = __builtin_amdgcn_exp2f(hi);
    return __builtin_fmaf(e * 0.6931471805599453f, lo, e);
}

// fp16x2 split of an fp32 value: v = hi + lo/2048 (22 significant bits).
// The value saturates at the largest finite fp16 (65504) instead of overflowing to inf: an out-of-range activation
// yields a finite, wrong value (hi = +-65504, lo = 0) that `hi == +-65504` flags (ops.check_split_range), never an
// inf - inf = NaN that would spread through the matches.  One v_med3 on the input covers both planes: with
// |xc| <= 65504 the remainder (xc - hi) * 2048 is at most 2^15, so lo needs no clamp of its own (a second one
// measured 0.7 % of both steps together with this one).  In-range values split exactly as before.
// Values below the fp16 normal range simply get a subnormal hi: gfx950's MFMA keeps fp16 subnormal operands exactly
// (tools/probe_mfma_denorm.py: 5.96e-8 in, 5.96e-8 out), so the compare-and-select that used to route them to lo alone
// was two VALU operations per output element for nothing (the GEMM epilogue is VALU-bound).
__device__ __forceinline__ void split_f32(float x, _Float16& hi, _Float16& lo) {
    const float xc = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    const _Float16 h = (_Float16)xc;
    hi = h;
    lo = (_Float16)((xc - (float)h) * 2048.f);
}

typedef _Float16 dfsfm_half4 __attribute__((ext_vector_type(4)));

// Store 4 consecutive values as fp32 (if out) and/or as split fp16 planes (if outh).
__device__ __forceinline__ void store4(float* out, _Float16* outh, _Float16* outl, int64_t o32, int64_t o16,
                                       const f32x4 v) {
    if (out) *reinterpret_cast<f32x4*>(out + o32) = v;
    if (outh) {
        dfsfm_half4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 a, b;
            split_f32(v[e], a, b);
            h[e] = a;
            l[e] = b;
        }
        *reinterpret_cast<dfsfm_half4*>(outh + o16) = h;
        *reinterpret_cast<dfsfm_half4*>(outl + o16) = l;
    }
}

// x + (the same variable of lane ^ 16 / lane ^ 32) without the LDS crossbar that __shfl_xor goes through (ds_bpermute_b32 and
// an lgkmcnt wait): gfx950's v_permlane16_swap / v_permlane32_swap hand both lanes of a pair the two values in the same order,
// and the addition is commutative, so the sum has the bits of x + __shfl_xor(x, 16 / 32) on both.
__device__ __forceinline__ float add_xor16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the variable of lane ^ 32
__device__ __forceinline__ float from_xor32(float x, int lane) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(lane & 32 ? r[0] : r[1]);
}

}  // namespace dfsfm
