// K6/K9 first layers -- direct convolution for tiny input depth on gfx950 (MI355X).
//
// The stem of ResNetFPN_8_2 (7x7 stride 2, 1 -> 128 channels; third_party/LoFTR/src/loftr/backbone/
// resnet_fpn.py:100-104) and conv1_1 of the S2DNet VGG encoder (3x3, 3 -> 64 channels;
// src/MultiviewMatcher/backbone/S2DNet/s2dnet.py:127-175) have K = kh*kw*Cin = 49 / 27: far too shallow for
// the matrix cores (a 32-deep K slab would be mostly padding and the implicit-GEMM gather dominates).  They
// are HBM/VALU work: one thread owns one output pixel, keeps its kh*kw*Cin input taps and all Cout
// accumulators in registers and runs an fp32 FMA chain (v_pk_fma_f32, two channels per instruction) against
// weights that every lane reads from the same address -- scalar loads through the constant cache, no LDS.
// Bias (folded BN) + ReLU + the fp16 hi/lo split for the next layer are fused; each thread writes its pixel's
// channels as contiguous 16-byte pieces.  Arithmetic is exact fp32 (no operand splitting needed here).
//
// Algorithmic bytes per output pixel: Cin*stride^2*4 read + Cout*4 written (split planes) -> HBM-bound at
// ~0.2 ms (stem, 16 x 480x640) / ~0.7 ms (conv1_1, 10000 35x35 patches); FMA floor 0.15 / 0.3 ms.
#include "common.h"

namespace {

using namespace dfsfm;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct DirectArgs {
    const float* x;            // element (n,y,x,c) at n*sxn + y*sxh + x*ldx + c
    const float* w;            // [kh*kw*Cin][Cout] fp32, taps in (ky,kx,ci) order
    const float* bias;         // [Cout] or null
    float* out;                // [M][ldo] fp32 or null
    _Float16* outh;            // split planes [M][ldo_s] or null
    _Float16* outl;
    int64_t sxn, sxh, ldx, ldo, ldo_s, M;
    int H, W, Ho, Wo, relu;
};

template <int CIN, int KS, int STRIDE, int PAD, int COUT>
__global__ __launch_bounds__(256) void direct_conv_kernel(DirectArgs g) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = m < g.M;
    const int64_t mc = valid ? m : g.M - 1;
    const int ox = (int)(mc % g.Wo);
    const int64_t t = mc / g.Wo;
    const int oy = (int)(t % g.Ho);
    const int64_t n = t / g.Ho;
    const float* img = g.x + n * g.sxn;

    // Cout in passes of 64 channels (32 packed accumulators).  The tap loop is a real loop: only one tap's weights
    // (64 SGPRs, four s_load_dwordx16) are live at a time, the next tap's input is fetched while the current tap's
    // 32 v_pk_fma_f32 run, and 4-6 waves per SIMD cover the scalar-load latency.
    constexpr int PASS = 64;
    const f32x2* __restrict__ b2 = reinterpret_cast<const f32x2*>(g.bias);
    const f32x2* __restrict__ w2 = reinterpret_cast<const f32x2*>(g.w);      // uniform addresses: scalar loads
    auto tap_load = [&](int kk, float (&v)[CIN]) __attribute__((always_inline)) {
        const int ky = kk / KS, kx = kk - ky * KS;
        const int iy = oy * STRIDE + ky - PAD, ix = ox * STRIDE + kx - PAD;
        const bool ok = valid && kk < KS * KS && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        const float* p = img + (int64_t)(ok ? iy : 0) * g.sxh + (int64_t)(ok ? ix : 0) * g.ldx;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            const float t = p[c];
            v[c] = ok ? t : 0.f;
        }
    };
#pragma unroll
    for (int c0 = 0; c0 < COUT; c0 += PASS) {
        f32x2 acc[PASS / 2];
#pragma unroll
        for (int c = 0; c < PASS / 2; ++c) acc[c] = b2 ? b2[c0 / 2 + c] : f32x2{0.f, 0.f};
        float cur[CIN], nxt[CIN];
        tap_load(0, cur);
#pragma unroll 1
        for (int kk = 0; kk < KS * KS; ++kk) {
            tap_load(kk + 1, nxt);
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const f32x2 xv = {cur[ci], cur[ci]};
                const f32x2* __restrict__ wk = reinterpret_cast<const f32x2*>(
                    __builtin_assume_aligned(w2 + (int64_t)(kk * CIN + ci) * (COUT / 2) + c0 / 2, 256));
#pragma unroll
                for (int c = 0; c < PASS / 2; ++c) acc[c] += xv * wk[c];
            }
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) cur[ci] = nxt[ci];
        }
        // bias is in the accumulators; ReLU, split, and each lane writes its pixel's 64 channels as 16-byte pieces
        // (measured: routing the stores through an LDS transpose for full-line writes is not faster -- the layer
        // is bound by the bytes it writes)
#pragma unroll
        for (int c8 = 0; c8 < PASS / 8; ++c8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[2 * q] = acc[c8 * 4 + q][0];
                v[2 * q + 1] = acc[c8 * 4 + q][1];
            }
            if (g.relu) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            const int ch = c0 + c8 * 8;
            if (!valid) continue;
            if (g.out) {
                *reinterpret_cast<f32x4*>(g.out + m * g.ldo + ch) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(g.out + m * g.ldo + ch + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
            if (g.outh) {
                half8 h, l;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    _Float16 a, b;
                    split_f32(v[q], a, b);
                    h[q] = a;
                    l[q] = b;
                }
                *reinterpret_cast<half8*>(g.outh + m * g.ldo_s + ch) = h;
                *reinterpret_cast<half8*>(g.outl + m * g.ldo_s + ch) = l;
            }
        }
    }
}

}  // namespace

extern "C" int dfsfm_conv2d_direct_f32(const float* x, int64_t sxn, int64_t sxh, int64_t ldx, int Nimg, int H, int W,
                                       int Cin, const float* w, int Cout, int kh, int kw, int stride, int pad,
                                       const float* bias, int relu, float* out, int64_t ldo, void* out_hi,
                                       void* out_lo, int64_t ldo_s, void* stream_) {
    if (!x || !w || (!out && !out_hi) || ((out_hi == nullptr) != (out_lo == nullptr))) return DFSFM_E_BADARG;
    if (Nimg <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0)
        return DFSFM_E_BADARG;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || ldx < Cin || (out && ldo < Cout) || (out_hi && ldo_s < Cout)) return DFSFM_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(w) & 255) || (bias && (reinterpret_cast<uintptr_t>(bias) & 7))) return DFSFM_E_UNSUPPORTED;
    if (out && ((ldo & 3) || (reinterpret_cast<uintptr_t>(out) & 15))) return DFSFM_E_UNSUPPORTED;
    if (out_hi && ((ldo_s & 7) || (reinterpret_cast<uintptr_t>(out_hi) & 15) || (reinterpret_cast<uintptr_t>(out_lo) & 15)))
        return DFSFM_E_UNSUPPORTED;
    DirectArgs g{};
    g.x = x; g.w = w; g.bias = bias; g.out = out;
    g.outh = static_cast<_Float16*>(out_hi); g.outl = static_cast<_Float16*>(out_lo);
    g.sxn = sxn; g.sxh = sxh; g.ldx = ldx; g.ldo = ldo; g.ldo_s = ldo_s;
    g.M = (int64_t)Nimg * Ho * Wo; g.H = H; g.W = W; g.Ho = Ho; g.Wo = Wo; g.relu = relu;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid((unsigned)((g.M + 255) / 256)), blk(256);
    if (Cin == 1 && kh == 7 && kw == 7 && stride == 2 && pad == 3 && Cout == 128)
        hipLaunchKernelGGL((direct_conv_kernel<1, 7, 2, 3, 128>), grid, blk, 0, stream, g);
    else if (Cin == 3 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && Cout == 64)
        hipLaunchKernelGGL((direct_conv_kernel<3, 3, 1, 1, 64>), grid, blk, 0, stream, g);
    else
        return DFSFM_E_UNSUPPORTED;        // callers fall back to dfsfm_conv2d_nhwc_f32 (implicit GEMM)
    return dfsfm::check_launch("dfsfm_conv2d_direct_f32");
}
