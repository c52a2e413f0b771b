// Image feeding on the device (last part) -- gfx950 (MI355X).
//
// The reference resizes every decoded frame on the host with PIL's LANCZOS filter before it reaches the matcher
// (src/dataset/utils.py:80-177: read_rgb / read_grayscale -> resize_image(..., "pil_LANCZOS") -> pad_bottom_right ->
// grayscale2tensor / rgb2tensor).  PIL resamples 8-bit images in fixed point (Pillow src/libImaging/Resample.c:
// precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc): double coefficients are
// rounded to 22 fractional bits, a pixel is   clip8((2^21 + sum_x src[xmin + x] * k[x]) >> 22)   in int32, horizontal
// pass first, its result rounded to 8 bits before the vertical pass.  That is integer arithmetic, so the kernels below
// return PIL's bytes exactly.  The coefficient tables (a few KB, doubles -> int32) are built on the host
// (detectorfreesfm_amd/images.py) and passed in; the kernels are byte streams: u8 in, u8 (and / or the fp32 / 255,
// zero-padded, channel-first tensor the matcher takes, plus the padding mask) out.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Resample.c
constexpr int HX = 64;                       // output columns per workgroup of the horizontal pass
constexpr int HROWS = 32;                    // source rows per workgroup of the horizontal pass

__device__ __forceinline__ int clip8(int v) {   // clip8_lookups[v >> PRECISION_BITS]
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[y][xx][c] = clip8(2^21 + sum_x src[y][xmin(xx) + x][c] * k[xx][x]); 64 columns x 4 row lanes per workgroup, the
// columns' coefficients transposed into LDS ([x][column]: conflict-free), rows y0 .. y0 + HROWS walked by the row lanes.
template <int C>
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, int64_t src_stride, int H,
                                                         const int32_t* __restrict__ bounds,
                                                         const int32_t* __restrict__ kk, int ksize, int Wn,
                                                         uint8_t* __restrict__ dst) {
    extern __shared__ int32_t kl[];   // [ksize][HX]
    const int lx = threadIdx.x & (HX - 1), ly = threadIdx.x / HX;
    const int xx0 = blockIdx.x * HX, xx = xx0 + lx;
    for (int i = threadIdx.x; i < ksize * HX; i += 256) {
        const int col = i / ksize, x = i % ksize;
        kl[x * HX + col] = xx0 + col < Wn ? kk[(int64_t)(xx0 + col) * ksize + x] : 0;
    }
    __syncthreads();
    if (xx >= Wn) return;
    const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
    const int y0 = blockIdx.y * HROWS;
    for (int y = y0 + ly; y < min(y0 + HROWS, H); y += 4) {
        const uint8_t* row = src + y * src_stride + (int64_t)xmin * C;
        int ss[C];
#pragma unroll
        for (int c = 0; c < C; ++c) ss[c] = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < xmax; ++x) {
            const int k = kl[x * HX + lx];
#pragma unroll
            for (int c = 0; c < C; ++c) ss[c] += (int)row[x * C + c] * k;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) dst[((int64_t)y * Wn + xx) * C + c] = (uint8_t)clip8(ss[c]);
    }
}

// Vertical pass + the tensor epilogue.  One thread = one byte column (xx, c) of 4 consecutive output rows; a row's
// bounds and coefficients are wave-uniform (scalar loads).  The grid covers the padded frame: positions outside the
// resized image get 0 (pad_bottom_right) and mask 0.
template <int C>
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ tmp, int Wn,
                                                         const int32_t* __restrict__ bounds,
                                                         const int32_t* __restrict__ kk, int ksize, int Hn,
                                                         uint8_t* __restrict__ out_u8, float* __restrict__ out_f32,
                                                         float* __restrict__ mask, int pad_h, int pad_w,
                                                         const float* __restrict__ lut) {
    const int col = blockIdx.x * 256 + threadIdx.x;       // byte column of the (padded) frame: x * C + c
    const int x = col / C, c = col % C;
    const int wmax = out_f32 ? pad_w : Wn;
    if (x >= wmax) return;
    const int64_t rowbytes = (int64_t)Wn * C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int yy = blockIdx.y * 4 + r;
        if (yy >= (out_f32 ? pad_h : Hn)) return;
        const bool inside = yy < Hn && x < Wn;
        int v = 0;
        if (inside) {
            const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
            const int32_t* k = kk + (int64_t)yy * ksize;
            int ss = 1 << (PRECISION_BITS - 1);
            for (int y = 0; y < ymax; ++y) ss += (int)tmp[(ymin + y) * rowbytes + col] * k[y];
            v = clip8(ss);
            if (out_u8) out_u8[yy * rowbytes + col] = (uint8_t)v;
        }
        if (out_f32) out_f32[((int64_t)c * pad_h + yy) * pad_w + x] = inside ? lut[v] : 0.f;
        if (mask && c == 0) mask[(int64_t)yy * pad_w + x] = inside ? 1.f : 0.f;
    }
}

template <int C>
int launch(const uint8_t* src, int64_t src_stride, int H, const int32_t* bx, const int32_t* kx, int ksx, int Wn,
           const int32_t* by, const int32_t* ky, int ksy, int Hn, uint8_t* tmp, uint8_t* out_u8, float* out_f32,
           float* mask, int pad_h, int pad_w, const float* lut, hipStream_t stream) {
    const size_t smem = (size_t)ksx * HX * sizeof(int32_t);
    hipLaunchKernelGGL(resample_h_kernel<C>, dim3((unsigned)((Wn + HX - 1) / HX), (unsigned)((H + HROWS - 1) / HROWS)),
                       dim3(256), smem, stream, src, src_stride, H, bx, kx, ksx, Wn, tmp);
    const int cols = (out_f32 ? pad_w : Wn) * C, rows = out_f32 ? pad_h : Hn;
    hipLaunchKernelGGL(resample_v_kernel<C>, dim3((unsigned)((cols + 255) / 256), (unsigned)((rows + 3) / 4)), dim3(256), 0,
                       stream, tmp, Wn, by, ky, ksy, Hn, out_u8, out_f32, mask, pad_h, pad_w, lut);
    return dfsfm::check_launch("dfsfm_resample_u8");
}

}  // namespace

// src [H, W, C] bytes (row pitch src_stride), C = 1 ('L') or 3 ('RGB').  bounds_* [out, 2] = (first source index, tap
// count), kk_* [out, ksize_*] = PIL's normalised 22-bit coefficients.  tmp: H * Wn * C bytes of scratch.  Outputs (any
// subset): out_u8 [Hn, Wn, C]; out_f32 [C, pad_h, pad_w] = lut256[byte] inside the image, 0 in the padding; mask
// [pad_h, pad_w] (1 / 0, needs out_f32's frame).
extern "C" int dfsfm_resample_u8(const uint8_t* src, int64_t src_stride, int H, int W, int C, const int32_t* bounds_x,
                                 const int32_t* kk_x, int ksize_x, int Wn, const int32_t* bounds_y, const int32_t* kk_y,
                                 int ksize_y, int Hn, uint8_t* tmp, uint8_t* out_u8, float* out_f32, float* mask,
                                 int pad_h, int pad_w, const float* lut256, void* stream_) {
    if (!src || !bounds_x || !kk_x || !bounds_y || !kk_y || !tmp || (!out_u8 && !out_f32)) return DFSFM_E_BADARG;
    if (H <= 0 || W <= 0 || Hn <= 0 || Wn <= 0 || ksize_x <= 0 || ksize_y <= 0 || src_stride < (int64_t)W * C) return DFSFM_E_BADARG;
    if (out_f32 && (!lut256 || pad_h < Hn || pad_w < Wn)) return DFSFM_E_BADARG;
    if (mask && !out_f32) return DFSFM_E_BADARG;
    if (C != 1 && C != 3) return DFSFM_E_UNSUPPORTED;
    if ((size_t)ksize_x * HX * sizeof(int32_t) > 64 * 1024) return DFSFM_E_UNSUPPORTED;      // > 40x reduction
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (C == 1)
        return launch<1>(src, src_stride, H, bounds_x, kk_x, ksize_x, Wn, bounds_y, kk_y, ksize_y, Hn, tmp, out_u8, out_f32,
                         mask, pad_h, pad_w, lut256, stream);
    return launch<3>(src, src_stride, H, bounds_x, kk_x, ksize_x, Wn, bounds_y, kk_y, ksize_y, Hn, tmp, out_u8, out_f32, mask,
                     pad_h, pad_w, lut256, stream);
}
