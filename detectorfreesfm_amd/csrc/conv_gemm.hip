// K2 / K6 / K9 -- implicit-GEMM convolution and linear layers on the fp16 matrix cores with an
// exact-class fp16x2 operand split, for gfx950 (MI355X).
//
// Replaces the dense layers the reference runs through cuDNN / cuBLAS:
//   nn.Conv2d + eval BatchNorm + ReLU (+ residual)   third_party/LoFTR/src/loftr/backbone/resnet_fpn.py:15-40,100-118
//   VGG / adaptation convs of S2DNet                 src/MultiviewMatcher/backbone/S2DNet/s2dnet.py:24-52,127-175
//   nn.Linear q/k/v/merge/mlp                        third_party/LoFTR/src/loftr/loftr_module/transformer.py:21-31
//
// Why not fp32 MFMA: v_mfma_f32_32x32x2_f32 runs at 1/16 of the fp16 rate (157 TF vs ~2.2 PF).
// Every fp32 operand x is split into two fp16 numbers, hi = fp16(x) and lo = fp16((x-hi)*2^11)
// (22 significant bits; |x| < 2^-14 goes entirely to lo so no fp16 subnormal is ever fed to the
// matrix core); the product uses three MFMAs
//        acc_m += A_hi B_hi          acc_x += A_hi B_lo + A_lo B_hi          (lo*lo dropped, 2^-22)
// with fp32 accumulation, and the result is acc_m + acc_x * 2^-11.  fp16*fp16 products are exact
// in fp32, so the only error beyond plain fp32 arithmetic is the dropped 2^-22 term: measured GEMM
// error vs fp64 is 5.9e-7 for both this scheme and fp32 (K=1152), and the end-to-end outputs of
// both plugins are unchanged.  Operand range: |x| < 65504.
//
// One kernel family: out[m, co] = sum_{ky,kx,ci} in[pix(m,ky,kx), ci] * w[co, ky, kx, ci], NHWC,
// M = Nimg*Ho*Wo output pixels, K = kh*kw*Cin flattened (slabs may straddle taps, 4-channel
// granularity), weights pre-split to fp16 [Npad][Kpad] hi / lo.  A linear layer is the 1x1 case
// with a row stride.  128x128 tile, BK=32, 4 waves x (2x2 MFMA 32x32x16 tiles), fragments read with
// conflict-free swizzled ds_read_b128, next slab prefetched into registers and split while the
// MFMAs run; epilogue fuses bias (folded BN), residual, ReLU.
#include "common.h"

namespace {

using namespace dfsfm;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE_B = BM * BK * 2;                 // bytes of one fp16 tile (8 KB)
constexpr int STAGE_B = 4 * TILE_B;                 // A_hi, A_lo, B_hi, B_lo
constexpr int SMEM_BYTES = 2 * STAGE_B;             // double buffered: 64 KB

struct ConvArgs {
    const float* x;          // NHWC input: element (n,y,x,c) at n*sxn + y*sxh + x*ldx + c (floats)
    const _Float16* wh;      // [Npad][Kpad]
    const _Float16* wl;
    const float* bias;       // [Cout] or null
    const float* res;        // [M][ldr] or null
    float* out;              // [M][ldo]
    int64_t ldx, sxh, sxn, ldr, ldo, M;
    int H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, K, Kpad, relu;
};

__device__ __forceinline__ void split4(const f32x4 v, half4& hi, half4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e];
        const _Float16 h = fabsf(x) >= 6.103515625e-05f ? (_Float16)x : (_Float16)0.f;
        hi[e] = h;
        lo[e] = (_Float16)((x - (float)h) * 2048.f);
    }
}

// byte offset of (row, 16-byte k-slot) inside a [128][32] fp16 tile; XOR swizzle makes the
// ds_read_b128 fragment reads of 16 different rows conflict-free (64-byte rows, 4 rows per bank row)
__device__ __forceinline__ int tile_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

struct RowGeom {           // one activation row handled by this thread
    int iy0, ix0;
    int64_t base;
    bool ok;
};

template <bool VEC_A>
__device__ __forceinline__ void load_slab(const ConvArgs& g, int k0, int c4, int tid, int n0, const RowGeom& r0,
                                          const RowGeom& r1, const RowGeom& r2, const RowGeom& r3, f32x4& a0,
                                          f32x4& a1, f32x4& a2, f32x4& a3, uint4& bh0, uint4& bh1, uint4& bl0,
                                          uint4& bl1) {
    const int k = k0 + c4 * 4;
    auto one = [&](const RowGeom& rg) __attribute__((always_inline)) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (VEC_A) {
            const bool kok = k < g.K;
            const int tap = kok ? k / g.Cin : 0;
            const int ci = k - tap * g.Cin;
            const int ky = tap / g.kw, kx = tap - ky * g.kw;
            const int iy = rg.iy0 + ky, ix = rg.ix0 + kx;
            if (kok && rg.ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                v = *reinterpret_cast<const f32x4*>(g.x + rg.base + (int64_t)iy * g.sxh + (int64_t)ix * g.ldx + ci);
        } else {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ke = k + e;
                if (ke < g.K && rg.ok) {
                    const int tap = ke / g.Cin, ci = ke - tap * g.Cin;
                    const int ky = tap / g.kw, kx = tap - ky * g.kw;
                    const int iy = rg.iy0 + ky, ix = rg.ix0 + kx;
                    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                        t[e] = g.x[rg.base + (int64_t)iy * g.sxh + (int64_t)ix * g.ldx + ci];
                }
            }
            v = f32x4{t[0], t[1], t[2], t[3]};
        }
        return v;
    };
    a0 = one(r0);
    a1 = one(r1);
    a2 = one(r2);
    a3 = one(r3);
    {
        const int r = tid >> 2, c = tid & 3;
        const int64_t off = (int64_t)(n0 + r) * g.Kpad + k0 + c * 8;
        bh0 = *reinterpret_cast<const uint4*>(g.wh + off);
        bl0 = *reinterpret_cast<const uint4*>(g.wl + off);
        const int64_t off2 = off + (int64_t)64 * g.Kpad;
        bh1 = *reinterpret_cast<const uint4*>(g.wh + off2);
        bl1 = *reinterpret_cast<const uint4*>(g.wl + off2);
    }
}

__device__ __forceinline__ void store_slab(char* st, int tid, int c4, const f32x4& a0, const f32x4& a1,
                                           const f32x4& a2, const f32x4& a3, const uint4& bh0, const uint4& bh1,
                                           const uint4& bl0, const uint4& bl1) {
    auto put = [&](const f32x4& a, int r) __attribute__((always_inline)) {
        half4 hi, lo;
        split4(a, hi, lo);
        const int off = tile_off(r, c4 >> 1) + (c4 & 1) * 8;
        *reinterpret_cast<half4*>(st + off) = hi;
        *reinterpret_cast<half4*>(st + TILE_B + off) = lo;
    };
    const int r = tid >> 3;
    put(a0, r);
    put(a1, r + 32);
    put(a2, r + 64);
    put(a3, r + 96);
    const int rb = tid >> 2, c = tid & 3;
    *reinterpret_cast<uint4*>(st + 2 * TILE_B + tile_off(rb, c)) = bh0;
    *reinterpret_cast<uint4*>(st + 3 * TILE_B + tile_off(rb, c)) = bl0;
    *reinterpret_cast<uint4*>(st + 2 * TILE_B + tile_off(rb + 64, c)) = bh1;
    *reinterpret_cast<uint4*>(st + 3 * TILE_B + tile_off(rb + 64, c)) = bl1;
}

template <bool VEC_A>   // VEC_A: Cin % 4 == 0 -> 16-byte activation loads; else scalar gathers
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kgrp = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread geometry of its 4 activation rows (rows (tid>>3) + 32 j, k-group c4) ----------
    const int c4 = tid & 7;
    auto geom = [&](int j) __attribute__((always_inline)) -> RowGeom {
        RowGeom rg;
        const int64_t m = m0 + (tid >> 3) + 32 * j;
        rg.ok = m < g.M;
        const int64_t mm = rg.ok ? m : 0;
        const int ox = (int)(mm % g.Wo);
        const int64_t t = mm / g.Wo;
        const int oy = (int)(t % g.Ho);
        const int64_t n = t / g.Ho;
        rg.iy0 = oy * g.stride - g.pad;
        rg.ix0 = ox * g.stride - g.pad;
        rg.base = n * g.sxn;
        return rg;
    };
    const RowGeom g0 = geom(0), g1 = geom(1), g2 = geom(2), g3 = geom(3);
    f32x4 a0, a1, a2, a3;
    uint4 bh0, bh1, bl0, bl1;
#define GLOAD(k0) load_slab<VEC_A>(g, (k0), c4, tid, n0, g0, g1, g2, g3, a0, a1, a2, a3, bh0, bh1, bl0, bl1)
#define LSTORE(buf) store_slab(smem + (buf) * STAGE_B, tid, c4, a0, a1, a2, a3, bh0, bh1, bl0, bl1)

    f32x16 accm[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            accm[i][j] = f32x16{0};
            accx[i][j] = f32x16{0};
        }

    const int nk = g.Kpad / BK;
    GLOAD(0);
    LSTORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) GLOAD((kt + 1) * BK);
        const char* st = smem + buf * STAGE_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wr * 64 + i * 32 + col;
                const int off = tile_off(r, ks * 2 + kgrp);
                ah[i] = *reinterpret_cast<const half8*>(st + off);
                al[i] = *reinterpret_cast<const half8*>(st + TILE_B + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wc * 64 + j * 32 + col;
                const int off = tile_off(r, ks * 2 + kgrp);
                bh[j] = *reinterpret_cast<const half8*>(st + 2 * TILE_B + off);
                bl[j] = *reinterpret_cast<const half8*>(st + 3 * TILE_B + off);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], accm[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) LSTORE(buf ^ 1);
        __syncthreads();
    }

#undef GLOAD
#undef LSTORE
    // ---- epilogue: combine, bias (folded BN), residual, ReLU; 128-byte row segments per store ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wc * 64 + j * 32 + col;
            if (n >= g.Cout) continue;
            const float b = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wr * 64 + i * 32 + mfma32_row(r, kgrp);
                if (m < g.M) {
                    float v = accm[i][j][r] + accx[i][j][r] * (1.f / 2048.f) + b;
                    if (g.res) v += g.res[m * g.ldr + n];
                    if (g.relu) v = fmaxf(v, 0.f);
                    g.out[m * g.ldo + n] = v;
                }
            }
        }
}

// 3x3 / stride 2 / pad 1 max pooling, NHWC (nn.MaxPool2d(3, 2, 1), s2dnet.py:89-92)
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                int H, int W, int C, int Ho, int Wo, int64_t total4) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total4) return;
    const int c4 = (int)(e % (C / 4));
    int64_t t = e / (C / 4);
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int64_t n = t / Ho;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            if (ix < 0 || ix >= W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((n * H + iy) * W + ix) * C + c4 * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], v[q]);
        }
    }
    *reinterpret_cast<f32x4*>(y + ((n * Ho + oy) * Wo + ox) * C + c4 * 4) = m;
}

}  // namespace

extern "C" int dfsfm_conv2d_nhwc_f32(const float* x, int64_t sxn, int64_t sxh, int64_t ldx, int Nimg, int H, int W,
                                     int Cin,
                                     const void* w_hi, const void* w_lo, int Cout, int Kpad, int kh, int kw,
                                     int stride, int pad, const float* bias, const float* residual, int64_t ldr,
                                     int relu, float* out, int64_t ldo, void* stream_) {
    if (!x || !w_hi || !w_lo || !out) return DFSFM_E_BADARG;
    if (Nimg <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0)
        return DFSFM_E_BADARG;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return DFSFM_E_BADARG;
    const int K = kh * kw * Cin;
    if (Kpad < K || Kpad % BK != 0 || ldx < Cin || ldo < Cout || (residual && ldr < Cout)) return DFSFM_E_BADARG;
    const bool vec = (Cin % 4 == 0) && (ldx % 4 == 0) && (sxh % 4 == 0) && (sxn % 4 == 0) &&
                     !(reinterpret_cast<uintptr_t>(x) & 15);
    if ((reinterpret_cast<uintptr_t>(w_hi) & 15) || (reinterpret_cast<uintptr_t>(w_lo) & 15)) return DFSFM_E_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ConvArgs g;
    g.x = x; g.wh = static_cast<const _Float16*>(w_hi); g.wl = static_cast<const _Float16*>(w_lo);
    g.bias = bias; g.res = residual; g.out = out; g.ldx = ldx; g.sxh = sxh; g.sxn = sxn; g.ldr = ldr; g.ldo = ldo;
    g.M = (int64_t)Nimg * Ho * Wo; g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.Cout = Cout;
    g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad; g.K = K; g.Kpad = Kpad; g.relu = relu;
    const dim3 grid((unsigned)((g.M + BM - 1) / BM), (unsigned)((Cout + BN - 1) / BN)), blk(256);
    static bool attr_set[2] = {false, false};
    if (vec) {
        if (!attr_set[0]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
            attr_set[0] = true;
        }
        hipLaunchKernelGGL(conv_gemm_kernel<true>, grid, blk, SMEM_BYTES, stream, g);
    } else {
        if (!attr_set[1]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
            attr_set[1] = true;
        }
        hipLaunchKernelGGL(conv_gemm_kernel<false>, grid, blk, SMEM_BYTES, stream, g);
    }
    return dfsfm::check_launch("dfsfm_conv2d_nhwc_f32");
}

extern "C" int dfsfm_maxpool3x3s2_nhwc_f32(const float* x, int Nimg, int H, int W, int C, float* out, void* stream_) {
    if (!x || !out || Nimg <= 0 || H <= 0 || W <= 0 || C <= 0) return DFSFM_E_BADARG;
    if (C % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return DFSFM_E_UNSUPPORTED;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total4 = (int64_t)Nimg * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), x, out, H, W, C, Ho, Wo, total4);
    return dfsfm::check_launch("dfsfm_maxpool3x3s2_nhwc_f32");
}
