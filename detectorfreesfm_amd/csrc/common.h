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

}  // namespace dfsfm
