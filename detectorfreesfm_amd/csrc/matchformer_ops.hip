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

// one thread = 4 consecutive channels of one pixel; neighbours come from L1/L2 (each input element is used 9 times)
template <int MODE>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w9c,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        _Float16* __restrict__ outh, _Float16* __restrict__ outl,
                                                        int H, int W, int C, int64_t total4) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total4) return;
    const int c4 = (int)(e % (C / 4));
    int64_t t = e / (C / 4);
    const int ox = (int)(t % W);
    t /= W;
    const int oy = (int)(t % H);
    const int64_t n = t / H;
    f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
    f32x4 centre = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + ky - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox + kx - 1;
            if (ix < 0 || ix >= W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((n * H + iy) * W + ix) * C + c4 * 4);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w9c + (ky * 3 + kx) * C + c4 * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(v[q], wv[q], acc[q]);
            if (ky == 1 && kx == 1) centre = v;
        }
    }
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (MODE == 1) r[q] = centre[q] * (1.f / (1.f + expf(-acc[q])));
        else if (MODE == 2) r[q] = gelu_erf(acc[q]);
        else r[q] = acc[q];
    }
    const int64_t o = ((n * H + oy) * W + ox) * C + c4 * 4;
    store4(out, outh, outl, o, o, r);
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
    const int64_t total4 = (int64_t)N * H * W * (C / 4);
    const dim3 grid((unsigned)((total4 + 255) / 256)), blk(256);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    _Float16* oh = static_cast<_Float16*>(out_hi);
    _Float16* ol = static_cast<_Float16*>(out_lo);
    if (mode == 0) hipLaunchKernelGGL(dwconv3x3_kernel<0>, grid, blk, 0, stream, x, w9c, bias, out, oh, ol, H, W, C, total4);
    else if (mode == 1) hipLaunchKernelGGL(dwconv3x3_kernel<1>, grid, blk, 0, stream, x, w9c, bias, out, oh, ol, H, W, C, total4);
    else hipLaunchKernelGGL(dwconv3x3_kernel<2>, grid, blk, 0, stream, x, w9c, bias, out, oh, ol, H, W, C, total4);
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
