// K8 -- RoIAlign patch extraction (TensorFlow crop_and_resize semantics) for gfx950 (MI355X).
//
// Replaces roi_align.RoIAlign(crop, crop, transform_fpcoor=False) -- the un-vendored
// third_party/RoIAlign.pytorch CUDA extension -- as called from
//   src/MultiviewMatcher/matcher_module/fine_preprocess.py:92-106
// Semantics restated in oracle/restate.py:roi_align_crop (parity unpinned: upstream source is
// absent from the reference tree).  Gather kernel, HBM/L2 bound: 14.7 KB written and <= 15.6 KB
// read per 3x35x35 patch.  One workgroup per patch; the sample coordinates of the patch are
// computed once into LDS; outputs are written x-fastest (coalesced), the four bilinear taps come
// from L2.  Optionally fuses the per-channel (x-mean)/std of S2DNet._forward (s2dnet.py:132-133)
// and scatters patches to caller-chosen slots.
//
// fp32 operation order follows the upstream implementation exactly (normalise by size-1, scale
// back), and FMA contraction is disabled so the result is bit-identical to the oracle.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAX_CROP = 128;

__global__ __launch_bounds__(256) void roi_align_kernel(
    const float* __restrict__ feat, int C, int H, int W, const float* __restrict__ boxes,
    const int32_t* __restrict__ box_ind, const int64_t* __restrict__ out_slot, int crop_h, int crop_w,
    float extrapolation, const float* __restrict__ mean, const float* __restrict__ stdv,
    float* __restrict__ out, int out_nhwc) {
    __shared__ float s_y[MAX_CROP], s_x[MAX_CROP];
    const int m = blockIdx.x;
    const float bx1 = boxes[m * 4 + 0], by1 = boxes[m * 4 + 1];
    const float bx2 = boxes[m * 4 + 2], by2 = boxes[m * 4 + 3];
    const float hm1 = (float)(H - 1), wm1 = (float)(W - 1);
    const float x1 = bx1 / wm1, x2 = bx2 / wm1, y1 = by1 / hm1, y2 = by2 / hm1;
    const int tid = threadIdx.x;
    if (tid < crop_h) {
        float v;
        if (crop_h > 1) {
            const float hs = ((y2 - y1) * hm1) / (float)(crop_h - 1);
            v = (y1 * hm1) + (float)tid * hs;
        } else {
            v = (0.5f * (y1 + y2)) * hm1;
        }
        s_y[tid] = v;
    }
    if (tid >= 128 && tid - 128 < crop_w) {
        const int ix = tid - 128;
        float v;
        if (crop_w > 1) {
            const float ws = ((x2 - x1) * wm1) / (float)(crop_w - 1);
            v = (x1 * wm1) + (float)ix * ws;
        } else {
            v = (0.5f * (x1 + x2)) * wm1;
        }
        s_x[ix] = v;
    }
    __syncthreads();

    const int b = box_ind ? box_ind[m] : 0;
    const float* img = feat + (int64_t)b * C * H * W;
    const int64_t slot = out_slot ? out_slot[m] : (int64_t)m;
    float* o = out + slot * C * crop_h * crop_w;
    const int plane = crop_h * crop_w, total = C * plane;
    for (int idx = tid; idx < total; idx += blockDim.x) {
        int c, rem;
        if (out_nhwc) {            // [crop_h, crop_w, C]: channel fastest (feeds the NHWC conv kernels)
            rem = idx / C;
            c = idx - rem * C;
        } else {                   // [C, crop_h, crop_w]
            c = idx / plane;
            rem = idx - c * plane;
        }
        const int iy = rem / crop_w, ix = rem - iy * crop_w;
        const float in_y = s_y[iy], in_x = s_x[ix];
        float val = extrapolation;
        if (in_y >= 0.f && in_y <= hm1 && in_x >= 0.f && in_x <= wm1) {
            const float ty = floorf(in_y), by = ceilf(in_y), lx = floorf(in_x), rx = ceilf(in_x);
            const float yl = in_y - ty, xl = in_x - lx;
            const float* pc = img + (int64_t)c * H * W;
            const float tl = pc[(int)ty * W + (int)lx], tr = pc[(int)ty * W + (int)rx];
            const float bl = pc[(int)by * W + (int)lx], br = pc[(int)by * W + (int)rx];
            const float top = tl + (tr - tl) * xl;
            const float bot = bl + (br - bl) * xl;
            val = top + (bot - top) * yl;
        }
        if (mean) val = (val - mean[c]) / stdv[c];
        o[idx] = val;
    }
}

// RGB frames (C = 3, the refinement head's case): one thread per output PIXEL and all three channels -- the sample coordinates,
// the bounds test and the bilinear weights are computed once per pixel instead of once per element, there is no division by C,
// and the PIX pixels a thread owns are processed together: their 12 * PIX taps are requested before the first one is used
// (the per-element loop waited for four L2 round trips per element, 15 times in a row).  Same fp32 operations per element in
// the same order as the generic kernel: bit-identical output.
template <int PIX>
__global__ __launch_bounds__(256) void roi_align_rgb_kernel(
    const float* __restrict__ feat, int H, int W, const float* __restrict__ boxes, const int32_t* __restrict__ box_ind,
    const int64_t* __restrict__ out_slot, int crop_h, int crop_w, float extrapolation, const float* __restrict__ mean,
    const float* __restrict__ stdv, float* __restrict__ out, int out_nhwc) {
    constexpr int C = 3;
    __shared__ float s_y[MAX_CROP], s_x[MAX_CROP];
    const int m = blockIdx.x;
    const float bx1 = boxes[m * 4 + 0], by1 = boxes[m * 4 + 1];
    const float bx2 = boxes[m * 4 + 2], by2 = boxes[m * 4 + 3];
    const float hm1 = (float)(H - 1), wm1 = (float)(W - 1);
    const float x1 = bx1 / wm1, x2 = bx2 / wm1, y1 = by1 / hm1, y2 = by2 / hm1;
    const int tid = threadIdx.x;
    if (tid < crop_h) {
        float v;
        if (crop_h > 1) {
            const float hs = ((y2 - y1) * hm1) / (float)(crop_h - 1);
            v = (y1 * hm1) + (float)tid * hs;
        } else {
            v = (0.5f * (y1 + y2)) * hm1;
        }
        s_y[tid] = v;
    }
    if (tid >= 128 && tid - 128 < crop_w) {
        const int ix = tid - 128;
        float v;
        if (crop_w > 1) {
            const float ws = ((x2 - x1) * wm1) / (float)(crop_w - 1);
            v = (x1 * wm1) + (float)ix * ws;
        } else {
            v = (0.5f * (x1 + x2)) * wm1;
        }
        s_x[ix] = v;
    }
    __syncthreads();

    const int b = box_ind ? box_ind[m] : 0;
    const int64_t hw = (int64_t)H * W;
    const float* img = feat + (int64_t)b * C * hw;
    const int64_t slot = out_slot ? out_slot[m] : (int64_t)m;
    const int plane = crop_h * crop_w;
    float* o = out + slot * C * plane;
    float mu[C] = {0.f, 0.f, 0.f}, sd[C] = {1.f, 1.f, 1.f};
    if (mean) {
#pragma unroll
        for (int c = 0; c < C; ++c) { mu[c] = mean[c]; sd[c] = stdv[c]; }
    }
    for (int p0 = tid; p0 < plane; p0 += 256 * PIX) {
        float tap[PIX][C][4], yl[PIX], xl[PIX];
        bool inside[PIX];
#pragma unroll
        for (int u = 0; u < PIX; ++u) {                    // every tap of the thread's PIX pixels is requested here ...
            const int p = p0 + u * 256;
            const int pc = p < plane ? p : 0;
            const int iy = pc / crop_w, ix = pc - iy * crop_w;
            const float in_y = s_y[iy], in_x = s_x[ix];
            inside[u] = in_y >= 0.f && in_y <= hm1 && in_x >= 0.f && in_x <= wm1;
            const float sy = inside[u] ? in_y : 0.f, sx = inside[u] ? in_x : 0.f;
            const float ty = floorf(sy), by = ceilf(sy), lx = floorf(sx), rx = ceilf(sx);
            yl[u] = sy - ty;
            xl[u] = sx - lx;
            const int o_tl = (int)ty * W + (int)lx, o_tr = (int)ty * W + (int)rx;
            const int o_bl = (int)by * W + (int)lx, o_br = (int)by * W + (int)rx;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float* pcn = img + c * hw;
                tap[u][c][0] = pcn[o_tl];
                tap[u][c][1] = pcn[o_tr];
                tap[u][c][2] = pcn[o_bl];
                tap[u][c][3] = pcn[o_br];
            }
        }
#pragma unroll
        for (int u = 0; u < PIX; ++u) {                    // ... and used here
            const int p = p0 + u * 256;
            if (p >= plane) break;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float val = extrapolation;
                if (inside[u]) {
                    const float top = tap[u][c][0] + (tap[u][c][1] - tap[u][c][0]) * xl[u];
                    const float bot = tap[u][c][2] + (tap[u][c][3] - tap[u][c][2]) * xl[u];
                    val = top + (bot - top) * yl[u];
                }
                if (mean) val = (val - mu[c]) / sd[c];
                o[out_nhwc ? p * C + c : c * plane + p] = val;
            }
        }
    }
}

}  // namespace

extern "C" int dfsfm_roi_align_f32(const float* feat, int Nimg, int C, int H, int W, const float* boxes,
                                   const int32_t* box_ind, const int64_t* out_slot, int M, int crop_h,
                                   int crop_w, float extrapolation_value, const float* mean,
                                   const float* std, float* out, int out_channels_last, void* stream_) {
    if (M == 0) return DFSFM_OK;
    if (!feat || !boxes || !out) return DFSFM_E_BADARG;
    if (Nimg <= 0 || C <= 0 || H <= 1 || W <= 1 || M < 0 || crop_h <= 0 || crop_w <= 0) return DFSFM_E_BADARG;
    if ((mean == nullptr) != (std == nullptr)) return DFSFM_E_BADARG;
    if (crop_h > MAX_CROP || crop_w > MAX_CROP) return DFSFM_E_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (C == 3 && (int64_t)H * W * 3 < 0x7fffffff)
        hipLaunchKernelGGL(roi_align_rgb_kernel<5>, dim3(M), dim3(256), 0, stream, feat, H, W, boxes, box_ind, out_slot, crop_h,
                           crop_w, extrapolation_value, mean, std, out, out_channels_last);
    else
        hipLaunchKernelGGL(roi_align_kernel, dim3(M), dim3(256), 0, stream, feat, C, H, W, boxes, box_ind,
                           out_slot, crop_h, crop_w, extrapolation_value, mean, std, out, out_channels_last);
    return dfsfm::check_launch("dfsfm_roi_align_f32");
}
