// Fused row-wise epilogue kernels for the encoder layers (K2 / K10) and the patch-feature
// hand-off (K9 -> K10) on gfx950 (MI355X).  All HBM-bound: every element is read once and written
// once, with row strides so results land directly where the next GEMM reads them.
//
//  * layernorm (+ residual, strided in/out) replaces
//        message = self.norm1(message)                      LoFTR transformer.py:50  / multiview :82
//        message = self.mlp(torch.cat([x, message], dim=2)) :55 / :87   (the concat: LN1 writes
//                                                            straight into the [x|message] buffer)
//        return x + self.norm2(message)                     :56-58 / :88-95
//  * add_scatter_tokens replaces the feature sum, 'm c h w -> m (h w) c' rearrange and the
//    original-order gather of MultiviewMatcher.py:240-270 / s2dnet.py:164-171 (fmap += upsample).
#include "common.h"

namespace {

// C = 128 (the refinement transformer and MatchFormer stage 1, whose 300K-row maps stream from HBM): 32 lanes x 16 B per
// row, two rows per wave pass, LN128_PASSES passes whose loads are all issued before the first reduction.
constexpr int LN128_PASSES = 4;
__global__ __launch_bounds__(256) void layernorm128_kernel(const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           const float* __restrict__ residual,
                                                           const _Float16* __restrict__ resh,
                                                           const _Float16* __restrict__ resl, int64_t ldr,
                                                           float* __restrict__ out, int64_t ldo, int64_t rows,
                                                           _Float16* __restrict__ outh, _Float16* __restrict__ outl,
                                                           int64_t ldos) {
    const int lane = threadIdx.x & 63, sub = lane >> 5, ch = (lane & 31) * 4;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (2 * LN128_PASSES) + sub;
    f32x4 v[LN128_PASSES];
#pragma unroll
    for (int p = 0; p < LN128_PASSES; ++p) {
        const int64_t row = row0 + 2 * p;
        v[p] = row < rows ? *reinterpret_cast<const f32x4*>(x + row * ldx + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + ch), bt = *reinterpret_cast<const f32x4*>(beta + ch);
#pragma unroll
    for (int p = 0; p < LN128_PASSES; ++p) {
        const int64_t row = row0 + 2 * p;
        float s = (v[p][0] + v[p][1]) + (v[p][2] + v[p][3]);
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        const float mean = s * (1.f / 128.f);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = v[p][e] - mean;
        float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) q += __shfl_xor(q, off);
        const float rstd = 1.f / sqrtf(q * (1.f / 128.f) + eps);
        if (row >= rows) continue;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = d[e] * rstd * g[e] + bt[e];
        if (residual) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(residual + row * ldr + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = r[e] + o[e];
        } else if (resh) {
            const dfsfm::dfsfm_half4 rh = *reinterpret_cast<const dfsfm::dfsfm_half4*>(resh + row * ldr + ch);
            const dfsfm::dfsfm_half4 rl = *reinterpret_cast<const dfsfm::dfsfm_half4*>(resl + row * ldr + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((float)rh[e] + (float)rl[e] * (1.f / 2048.f)) + o[e];
        }
        dfsfm::store4(out, outh, outl, row * ldo + ch, row * ldos + ch, o);
    }
}

// One wave per row, VEC = C/64 contiguous floats per lane (C = 64*VEC).
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ residual,
                                                        const _Float16* __restrict__ resh,
                                                        const _Float16* __restrict__ resl, int64_t ldr,
                                                        float* __restrict__ out, int64_t ldo, int64_t rows,
                                                        _Float16* __restrict__ outh, _Float16* __restrict__ outl,
                                                        int64_t ldos) {
    constexpr int C = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[VEC];
    const float* xr = x + row * ldx + lane * VEC;
    if (VEC == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(xr);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = t[e];
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = xr[e];
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += v[e];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) q += (v[e] - mean) * (v[e] - mean);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.f / sqrtf(q / (float)C + eps);
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e)
        o[e] = (v[e] - mean) * rstd * gamma[lane * VEC + e] + beta[lane * VEC + e];
    if (residual) {
        const float* rr = residual + row * ldr + lane * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = rr[e] + o[e];
    } else if (resh) {   // residual kept as split planes (hi + lo/2048): no fp32 copy of the token state exists
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            o[e] = ((float)resh[row * ldr + lane * VEC + e] + (float)resl[row * ldr + lane * VEC + e] * (1.f / 2048.f)) + o[e];
    }
    if (out) {
        float* orow = out + row * ldo + lane * VEC;
        if (VEC == 4) {
            f32x4 t;
#pragma unroll
            for (int e = 0; e < VEC; ++e) t[e] = o[e];
            *reinterpret_cast<f32x4*>(orow) = t;
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) orow[e] = o[e];
        }
    }
    if (outh) {   // the same values as split fp16 planes, for the LDS-DMA GEMM kernel
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            _Float16 a, b;
            dfsfm::split_f32(o[e], a, b);
            outh[row * ldos + lane * VEC + e] = a;
            outl[row * ldos + lane * VEC + e] = b;
        }
    }
}

// out_hi/out_lo[r, c] = split(x[r, c] (+ add[r % add_rows, c]));  optional fp32 copy of the sum.
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ add, int64_t add_rows,
                                                         float* __restrict__ out, int64_t ldo,
                                                         _Float16* __restrict__ outh, _Float16* __restrict__ outl,
                                                         int64_t ldos, int64_t rows, int C4) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C4) return;
    const int64_t r = e / C4;
    const int c = (int)(e - r * C4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    if (add) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(add + (r % add_rows) * (int64_t)(C4 * 4) + c);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += a[q];
    }
    dfsfm::store4(out, outh, outl, r * ldo + c, r * ldos + c, v);
}

// The same from a two-level source: row r lives at x + (r / blk_rows) * blk_stride + (r % blk_rows) * ldx (e.g. the tokens of
// views 1..Vq of every track inside a [T, V, WW, C] feature tensor) -- no intermediate contiguous copy.
__global__ __launch_bounds__(256) void split_rows_blocked_kernel(const float* __restrict__ x, int64_t blk_rows,
                                                                 int64_t blk_stride, int64_t ldx,
                                                                 _Float16* __restrict__ outh, _Float16* __restrict__ outl,
                                                                 int64_t ldos, int64_t rows, int C4) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C4) return;
    const int64_t r = e / C4;
    const int c = (int)(e - r * C4) * 4;
    const int64_t b = r / blk_rows;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + b * blk_stride + (r - b * blk_rows) * ldx + c);
    dfsfm::store4(nullptr, outh, outl, 0, r * ldos + c, v);
}

// dst[slot[m], p, c] = a[m, c, p] (+ b[m, c, p]);  one workgroup per patch, 32-channel slabs
// transposed through LDS so both the reads (p fastest) and the writes (c fastest) are coalesced.
__global__ __launch_bounds__(256) void add_scatter_tokens_kernel(const float* __restrict__ a,
                                                                 const float* __restrict__ b,
                                                                 const int64_t* __restrict__ slot,
                                                                 float* __restrict__ dst, int C, int P) {
    extern __shared__ float tile[];   // [32][P + 1]
    const int m = blockIdx.x;
    const int64_t s = slot ? slot[m] : (int64_t)m;
    const float* am = a + (int64_t)m * C * P;
    const float* bm = b ? b + (int64_t)m * C * P : nullptr;
    float* dm = dst + s * P * C;
    const int ld = P + 1;
    for (int c0 = 0; c0 < C; c0 += 32) {
        const int nc = min(32, C - c0);
        for (int e = threadIdx.x; e < nc * P; e += blockDim.x) {
            const int cy = e / P, p = e - cy * P;
            float v = am[(int64_t)(c0 + cy) * P + p];
            if (bm) v += bm[(int64_t)(c0 + cy) * P + p];
            tile[cy * ld + p] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < P * 32; e += blockDim.x) {
            const int p = e >> 5, cx = e & 31;
            if (cx < nc) dm[(int64_t)p * C + c0 + cx] = tile[cx * ld + p];
        }
        __syncthreads();
    }
}

}  // namespace

namespace {

// Separable resampling of NHWC patch maps: out[m, oy, ox, c] = sum_qy sum_qx By[oy, qy] * Bx[ox, qx] * y[m, qy, qx, c].
// Replaces nn.Upsample(mode='bicubic', align_corners=True) + centre crop of S2DNet's second adaptation map
// (backbone/S2DNet/s2dnet.py:164-193): By / Bx are the rows of PyTorch's own interpolation matrix that fall inside
// the window, so the coefficients are torch's.  One workgroup per (patch, 64-channel slice): the slice of y and the
// row-resampled intermediate live in LDS (21 + 35 KB for 9x9 -> 15x15: two workgroups per CU); HBM-bound
// (41 KB read, 115 KB written per 128-channel patch).
constexpr int RS_CH = 64;

__global__ __launch_bounds__(256) void resample_sep_kernel(const float* __restrict__ y, const float* __restrict__ By,
                                                           const float* __restrict__ Bx, float* __restrict__ out, int hin,
                                                           int win, int hout, int wout, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem_rs[];
    float* s_by = reinterpret_cast<float*>(smem_rs);                  // [hout][hin]
    float* s_bx = s_by + ((hout * hin + 3) & ~3);                     // [wout][win]   (16-byte aligned sections)
    f32x4* s_y = reinterpret_cast<f32x4*>(s_bx + ((wout * win + 3) & ~3));   // [hin*win][16]
    f32x4* s_t = s_y + hin * win * (RS_CH / 4);                       // [hout][win][16]
    const int m = blockIdx.x, c0 = blockIdx.y * RS_CH, tid = threadIdx.x;
    for (int e = tid; e < hout * hin; e += 256) s_by[e] = By[e];
    for (int e = tid; e < wout * win; e += 256) s_bx[e] = Bx[e];
    const float* ym = y + (int64_t)m * hin * win * C + c0;
    for (int e = tid; e < hin * win * (RS_CH / 4); e += 256) {
        const int q = e / (RS_CH / 4), c4 = e % (RS_CH / 4);
        s_y[e] = *reinterpret_cast<const f32x4*>(ym + (int64_t)q * C + c4 * 4);
    }
    __syncthreads();
    for (int e = tid; e < hout * win * (RS_CH / 4); e += 256) {       // rows first: t[oy][qx] = sum_qy By[oy][qy] y[qy][qx]
        const int c4 = e % (RS_CH / 4), r = e / (RS_CH / 4), qx = r % win, oy = r / win;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int qy = 0; qy < hin; ++qy) {
            const float w = s_by[oy * hin + qy];
            const f32x4 v = s_y[(qy * win + qx) * (RS_CH / 4) + c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = fmaf(w, v[k], a[k]);
        }
        s_t[e] = a;
    }
    __syncthreads();
    float* om = out + (int64_t)m * hout * wout * C + c0;
    for (int e = tid; e < hout * wout * (RS_CH / 4); e += 256) {      // then columns
        const int c4 = e % (RS_CH / 4), p = e / (RS_CH / 4), ox = p % wout, oy = p / wout;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int qx = 0; qx < win; ++qx) {
            const float w = s_bx[ox * win + qx];
            const f32x4 v = s_t[(oy * win + qx) * (RS_CH / 4) + c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = fmaf(w, v[k], a[k]);
        }
        *reinterpret_cast<f32x4*>(om + (int64_t)p * C + c4 * 4) = a;
    }
}

}  // namespace

extern "C" int dfsfm_resample_separable_f32(const float* y, int M, int hin, int win, int C, const float* By, const float* Bx,
                                            int hout, int wout, float* out, void* stream_) {
    if (M == 0) return DFSFM_OK;
    if (!y || !By || !Bx || !out || M < 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0 || C <= 0) return DFSFM_E_BADARG;
    if (C % RS_CH != 0 || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return DFSFM_E_UNSUPPORTED;
    const size_t smem = (size_t)(((hout * hin + 3) & ~3) + ((wout * win + 3) & ~3)) * 4 + (size_t)(hin * win + hout * win) * RS_CH * 4;
    if (smem > 64 * 1024) return DFSFM_E_UNSUPPORTED;
    hipLaunchKernelGGL(resample_sep_kernel, dim3((unsigned)M, (unsigned)(C / RS_CH)), dim3(256), smem, static_cast<hipStream_t>(stream_),
                       y, By, Bx, out, hin, win, hout, wout, C);
    return dfsfm::check_launch("dfsfm_resample_separable_f32");
}

extern "C" int dfsfm_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                   const float* residual, const void* res_hi, const void* res_lo, int64_t ldr,
                                   float* out, int64_t ldo, void* out_hi, void* out_lo, int64_t ldo_s,
                                   int64_t rows, int C, void* stream_) {
    if (rows == 0) return DFSFM_OK;
    if ((res_hi == nullptr) != (res_lo == nullptr) || (residual && res_hi)) return DFSFM_E_BADARG;
    const _Float16* rh = static_cast<const _Float16*>(res_hi);
    const _Float16* rl = static_cast<const _Float16*>(res_lo);
    if (!x || !gamma || !beta || (!out && !out_hi) || rows < 0 || C <= 0) return DFSFM_E_BADARG;
    if ((out_hi == nullptr) != (out_lo == nullptr)) return DFSFM_E_BADARG;
    if (ldx < C || (out && ldo < C) || (out_hi && ldo_s < C) || ((residual || res_hi) && ldr < C)) return DFSFM_E_BADARG;
    _Float16* oh = static_cast<_Float16*>(out_hi);
    _Float16* ol = static_cast<_Float16*>(out_lo);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
    if (C == 256) {
        if ((ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15) ||
            (out && ((ldo & 3) || (reinterpret_cast<uintptr_t>(out) & 15))))
            return DFSFM_E_UNSUPPORTED;
        hipLaunchKernelGGL(layernorm_kernel<4>, grid, blk, 0, stream, x, ldx, gamma, beta, eps, residual, rh, rl, ldr, out, ldo, rows, oh, ol, ldo_s);
    } else if (C == 128 && !(ldx & 3) && !(reinterpret_cast<uintptr_t>(x) & 15) && !(reinterpret_cast<uintptr_t>(gamma) & 15) &&
               !(reinterpret_cast<uintptr_t>(beta) & 15) && (!out || (!(ldo & 3) && !(reinterpret_cast<uintptr_t>(out) & 15))) &&
               (!oh || (!(ldo_s & 3) && !(reinterpret_cast<uintptr_t>(oh) & 7) && !(reinterpret_cast<uintptr_t>(ol) & 7))) &&
               (!residual || (!(ldr & 3) && !(reinterpret_cast<uintptr_t>(residual) & 15))) &&
               (!rh || (!(ldr & 3) && !(reinterpret_cast<uintptr_t>(rh) & 7) && !(reinterpret_cast<uintptr_t>(rl) & 7)))) {
        const int64_t per_blk = 4 * 2 * LN128_PASSES;
        hipLaunchKernelGGL(layernorm128_kernel, dim3((unsigned)((rows + per_blk - 1) / per_blk)), blk, 0, stream, x, ldx, gamma, beta, eps,
                           residual, rh, rl, ldr, out, ldo, rows, oh, ol, ldo_s);
    } else if (C == 128) {
        hipLaunchKernelGGL(layernorm_kernel<2>, grid, blk, 0, stream, x, ldx, gamma, beta, eps, residual, rh, rl, ldr, out, ldo, rows, oh, ol, ldo_s);
    } else if (C == 64) {
        hipLaunchKernelGGL(layernorm_kernel<1>, grid, blk, 0, stream, x, ldx, gamma, beta, eps, residual, rh, rl, ldr, out, ldo, rows, oh, ol, ldo_s);
    } else if (C == 192) {    // MatchFormer stage 2
        hipLaunchKernelGGL(layernorm_kernel<3>, grid, blk, 0, stream, x, ldx, gamma, beta, eps, residual, rh, rl, ldr, out, ldo, rows, oh, ol, ldo_s);
    } else if (C == 512) {    // MatchFormer stage 4
        hipLaunchKernelGGL(layernorm_kernel<8>, grid, blk, 0, stream, x, ldx, gamma, beta, eps, residual, rh, rl, ldr, out, ldo, rows, oh, ol, ldo_s);
    } else {
        return DFSFM_E_UNSUPPORTED;
    }
    return dfsfm::check_launch("dfsfm_layernorm_f32");
}

extern "C" int dfsfm_add_scatter_tokens_f32(const float* a, const float* b, const int64_t* slot, float* dst,
                                            int M, int C, int P, void* stream_) {
    if (M == 0) return DFSFM_OK;
    if (!a || !dst || M < 0 || C <= 0 || P <= 0) return DFSFM_E_BADARG;
    if ((size_t)32 * (P + 1) * 4 > 64 * 1024) return DFSFM_E_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(add_scatter_tokens_kernel, dim3(M), dim3(256), (size_t)32 * (P + 1) * 4, stream, a, b, slot,
                       dst, C, P);
    return dfsfm::check_launch("dfsfm_add_scatter_tokens_f32");
}

extern "C" int dfsfm_split_rows_f32(const float* x, int64_t ldx, const float* add, int64_t add_rows, float* out,
                                    int64_t ldo, void* out_hi, void* out_lo, int64_t ldo_s, int64_t rows, int C,
                                    void* stream_) {
    if (rows == 0) return DFSFM_OK;
    if (!x || (!out && !out_hi) || rows < 0 || C <= 0 || (add && add_rows <= 0)) return DFSFM_E_BADARG;
    if ((out_hi == nullptr) != (out_lo == nullptr)) return DFSFM_E_BADARG;
    if ((C & 3) || (ldx & 3) || (out && (ldo & 3)) || (out_hi && (ldo_s & 3))) return DFSFM_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (out && (reinterpret_cast<uintptr_t>(out) & 15)) ||
        (add && (reinterpret_cast<uintptr_t>(add) & 15)))
        return DFSFM_E_UNSUPPORTED;
    const int64_t total = rows * (C / 4);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), x, ldx, add, add_rows, out, ldo,
                       static_cast<_Float16*>(out_hi), static_cast<_Float16*>(out_lo), ldo_s, rows, C / 4);
    return dfsfm::check_launch("dfsfm_split_rows_f32");
}

extern "C" int dfsfm_split_rows_blocked_f32(const float* x, int64_t blk_rows, int64_t blk_stride, int64_t ldx, void* out_hi,
                                            void* out_lo, int64_t ldo_s, int64_t rows, int C, void* stream_) {
    if (rows == 0) return DFSFM_OK;
    if (!x || !out_hi || !out_lo || rows < 0 || C <= 0 || blk_rows <= 0 || ldx < C || ldo_s < C) return DFSFM_E_BADARG;
    if ((C & 3) || (ldx & 3) || (blk_stride & 3) || (ldo_s & 3) || (reinterpret_cast<uintptr_t>(x) & 15) ||
        (reinterpret_cast<uintptr_t>(out_hi) & 7) || (reinterpret_cast<uintptr_t>(out_lo) & 7))
        return DFSFM_E_UNSUPPORTED;
    const int64_t total = rows * (C / 4);
    hipLaunchKernelGGL(split_rows_blocked_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), x, blk_rows, blk_stride, ldx, static_cast<_Float16*>(out_hi),
                       static_cast<_Float16*>(out_lo), ldo_s, rows, C / 4);
    return dfsfm::check_launch("dfsfm_split_rows_blocked_f32");
}
