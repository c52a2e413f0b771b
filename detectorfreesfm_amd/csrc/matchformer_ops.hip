// Element-wise / stencil kernels the MatchFormer-LA coarse matcher needs beside the GEMM, attention and matching
// kernels it shares with LoFTR -- gfx950 (MI355X).  All HBM-bound: every element is read
// once and written once.
//
//  * dwconv3x3: depth-wise 3x3 convolution (groups = C, pad 1, bias) on NHWC fp32 maps with the consumer fused:
//        mode 1  x * sigmoid(dw(x))   Positional            third_party/MatchFormer/model/backbone/match_LA_large.py:108-116
//        mode 2  GELU(dw(x))          DWConv + act of Mlp   :15-27, :39-41
//    result as fp32 and / or fp16x2-split planes (the operand format of the next linear layer).
//  * bilinear_up: F.interpolate(mode='bilinear', align_corners=True) of the FPN top-down path (:232-236).
#include "common.h"

namespace {

using namespace dfsfm;

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// One thread = 4 consecutive channels of a vertical strip of DW_R output pixels: the DW_R + 2 input rows are read once
// per strip (3 * (DW_R + 2) 16-B loads for DW_R outputs instead of 9 each).  Per output the taps are still added in
// (ky, kx) order onto the bias.  Workgroups are dealt to the XCDs in contiguous bands (blockIdx round-robins over the
// 8 XCDs), so the x / y neighbours a strip shares with the next workgroup are in the same L2.
constexpr int DW_R = 4;
template <int MODE>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w9c,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        _Float16* __restrict__ outh, _Float16* __restrict__ outl,
                                                        int H, int W, int C, int nstrips, int64_t total, int64_t nblk) {
    const int64_t per_xcd = (nblk + 7) >> 3;
    const int64_t blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t e = blk * 256 + threadIdx.x;
    if (blk >= nblk || e >= total) return;
    const int c4 = (int)(e % (C / 4));
    int64_t t = e / (C / 4);
    const int ox = (int)(t % W);
    t /= W;
    const int oy0 = (int)(t % nstrips) * DW_R;
    const int64_t n = t / nstrips;
    f32x4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(w9c + k * C + c4 * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
    const bool has_l = ox > 0, has_r = ox + 1 < W;
    const int xl = has_l ? ox - 1 : ox, xr = has_r ? ox + 1 : ox;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[DW_R + 2][3];
#pragma unroll
    for (int r = 0; r < DW_R + 2; ++r) {            // input row oy0 + r - 1; rows / columns outside the map read as zero
        const int iy = oy0 + r - 1;
        const bool ok = iy >= 0 && iy < H;
        const float* row = x + ((n * H + (ok ? iy : oy0)) * W) * C + c4 * 4;
        const f32x4 l = *reinterpret_cast<const f32x4*>(row + (int64_t)xl * C);
        const f32x4 m = *reinterpret_cast<const f32x4*>(row + (int64_t)ox * C);
        const f32x4 rr = *reinterpret_cast<const f32x4*>(row + (int64_t)xr * C);
        v[r][0] = ok && has_l ? l : zero;
        v[r][1] = ok ? m : zero;
        v[r][2] = ok && has_r ? rr : zero;
    }
#pragma unroll
    for (int o = 0; o < DW_R; ++o) {
        if (oy0 + o >= H) break;
        f32x4 acc = b;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(v[o + ky][kx][q], wv[ky * 3 + kx][q], acc[q]);
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == 1) r[q] = v[o + 1][1][q] * (1.f / (1.f + expf(-acc[q])));
            else if (MODE == 2) r[q] = gelu_erf(acc[q]);
            else r[q] = acc[q];
        }
        const int64_t oo = ((n * H + oy0 + o) * W + ox) * C + c4 * 4;
        store4(out, outh, outl, oo, oo, r);
    }
}

// align_corners=True: src = dst * (in - 1) / (out - 1); 4-tap lerp in the operation order of ATen's CPU kernel
__global__ __launch_bounds__(256) void bilinear_up_kernel(const float* __restrict__ x, float* __restrict__ out, int hin,
                                                          int win, int hout, int wout, int C, float sy, float sx,
                                                          int64_t total4) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total4) return;
    const int c4 = (int)(e % (C / 4));
    int64_t t = e / (C / 4);
    const int ox = (int)(t % wout);
    t /= wout;
    const int oy = (int)(t % hout);
    const int64_t n = t / hout;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* b = x + n * hin * win * C + c4 * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((int64_t)y0 * win + x0) * C);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((int64_t)y0 * win + x1) * C);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((int64_t)y1 * win + x0) * C);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((int64_t)y1 * win + x1) * C);
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = ly0 * (lx0 * v00[q] + lx1 * v01[q]) + ly1 * (lx0 * v10[q] + lx1 * v11[q]);
    *reinterpret_cast<f32x4*>(out + ((n * hout + oy) * wout + ox) * C + c4 * 4) = r;
}

}  // namespace

extern "C" int dfsfm_dwconv3x3_nhwc_f32(const float* x, int N, int H, int W, int C, const float* w9c, const float* bias,
                                        int mode, float* out, void* out_hi, void* out_lo, void* stream_) {
    if (N == 0) return DFSFM_OK;
    if (!x || !w9c || !bias || (!out && !out_hi) || N < 0 || H <= 0 || W <= 0 || C <= 0) return DFSFM_E_BADARG;
    if ((out_hi == nullptr) != (out_lo == nullptr) || mode < 0 || mode > 2) return DFSFM_E_BADARG;
    if (C % 8 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w9c) & 15) ||
        (reinterpret_cast<uintptr_t>(bias) & 15) || (out && (reinterpret_cast<uintptr_t>(out) & 15)) ||
        (out_hi && ((reinterpret_cast<uintptr_t>(out_hi) & 7) || (reinterpret_cast<uintptr_t>(out_lo) & 7))))
        return DFSFM_E_UNSUPPORTED;
    const int nstrips = (H + DW_R - 1) / DW_R;
    const int64_t total = (int64_t)N * nstrips * W * (C / 4), nblk = (total + 255) / 256;
    if (((nblk + 7) >> 3) * 8 > 0x7fffffff) return DFSFM_E_UNSUPPORTED;
    const dim3 grid((unsigned)(((nblk + 7) >> 3) * 8)), blk(256);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    _Float16* oh = static_cast<_Float16*>(out_hi);
    _Float16* ol = static_cast<_Float16*>(out_lo);
    if (mode == 0) hipLaunchKernelGGL(dwconv3x3_kernel<0>, grid, blk, 0, stream, x, w9c, bias, out, oh, ol, H, W, C, nstrips, total, nblk);
    else if (mode == 1) hipLaunchKernelGGL(dwconv3x3_kernel<1>, grid, blk, 0, stream, x, w9c, bias, out, oh, ol, H, W, C, nstrips, total, nblk);
    else hipLaunchKernelGGL(dwconv3x3_kernel<2>, grid, blk, 0, stream, x, w9c, bias, out, oh, ol, H, W, C, nstrips, total, nblk);
    return dfsfm::check_launch("dfsfm_dwconv3x3_nhwc_f32");
}

extern "C" int dfsfm_bilinear_up_nhwc_f32(const float* x, int N, int hin, int win, int C, int hout, int wout, float* out,
                                          void* stream_) {
    if (N == 0) return DFSFM_OK;
    if (!x || !out || N < 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0 || C <= 0) return DFSFM_E_BADARG;
    if (C % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return DFSFM_E_UNSUPPORTED;
    // area_pixel_compute_scale(align_corners=True): (in - 1) / (out - 1), 0 when out == 1
    const float sy = hout > 1 ? (float)(hin - 1) / (float)(hout - 1) : 0.f;
    const float sx = wout > 1 ? (float)(win - 1) / (float)(wout - 1) : 0.f;
    const int64_t total4 = (int64_t)N * hout * wout * (C / 4);
    hipLaunchKernelGGL(bilinear_up_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), x, out, hin, win, hout, wout, C, sy, sx, total4);
    return dfsfm::check_launch("dfsfm_bilinear_up_nhwc_f32");
}
