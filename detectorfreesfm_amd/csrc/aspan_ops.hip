// Kernels the ASpanFormer coarse matcher needs beside the conv / linear / matching kernels it shares with LoFTR
//  -- gfx950 (MI355X).  Paths below are relative to
// third_party/aspantransformer/src/ASpanFormer/.
//
//  * avgpool:          F.avg_pool2d(x, k, stride=k) on NHWC maps       aspan_module/transformer.py:163-167, attention.py:60-66
//  * full_attention:   FullAttention.forward (softmax attention)       aspan_module/attention.py:141-165
//  * span_attention:   HierachicalAttention.partition_token + group_attention for one level: span statistics from the
//                      flow map, 8x8 bilinear samples of K / V per 2x2 query group, softmax over the 64 samples
//                                                                      aspan_module/attention.py:49-53, 64-66, 92-133
//  * layernorm2d:      (x - mean) / (std_unbiased + 1e-6) * affine + bias (+ residual)    aspan_module/attention.py:7-19
//  * upsample_bilinear: F.upsample(scale_factor=s, mode='bilinear') (align_corners=False)   aspan_module/transformer.py:177-180
//  * upsample_nearest: F.upsample(mode='nearest') into a column slice of the fused message  aspan_module/attention.py:85-86
//  * flow_decode:      sigmoid(flow[:2]) * (w, h) | flow[2:]            aspan_module/transformer.py:125-133
//
// The model runs one pair at a time (aspanformer.py:43), maps are 60x80 tokens and smaller: these are latency-sized
// kernels, written for exact operation order first (ATen's CPU kernels are the oracle) and coalesced NHWC access.
#include "common.h"
#include <cstdlib>

namespace {

using namespace dfsfm;

// ---------------------------------------------------------------- avg_pool2d(k, stride k)
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, int64_t ldx, int H, int W, int C4, int k,
                                                      float* __restrict__ out, int64_t ldo, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c4 = (int)(e % C4);
    int64_t t = e / C4;
    const int Wo = W / k, Ho = H / k;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int64_t n = t / Ho;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((n * H + oy * k + ky) * W + ox * k + kx) * ldx + c4 * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += v[q];
        }
    const float div = (float)(k * k);
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = s[q] / div;
    *reinterpret_cast<f32x4*>(out + ((n * Ho + oy) * Wo + ox) * ldo + c4 * 4) = s;
}

// ---------------------------------------------------------------- softmax attention, D = 32
// 256 threads = 32 queries x 8 key splits of one head.  Keys / values of the head pass through LDS in 64-row tiles (row
// pitch 33 floats: the 8 splits of a query read 8 different rows conflict-free, the queries of a split share a broadcast);
// every lane keeps an online-softmax partial (running maximum, rescaled sums) over its keys j = split, split + 8, ...; the 8
// partials of a query are merged with three xor-butterfly steps and each lane stores 4 of the 32 output channels.
template <int D>
__global__ __launch_bounds__(256) void full_attention_kernel(const float* __restrict__ q, int64_t ldq, int64_t sq,
                                                             const float* __restrict__ k, int64_t ldk, int64_t sk,
                                                             const float* __restrict__ v, int64_t ldv, int64_t sv,
                                                             float* __restrict__ out, int64_t ldo, int64_t so, int L, int S,
                                                             int kv_swap, float scale) {
    constexpr int LD = D + 1;
    __shared__ float ks[64 * LD], vs[64 * LD];
    const int tid = threadIdx.x, split = tid & 7, h = blockIdx.y, n = blockIdx.z, nk = n ^ kv_swap;
    const int l = blockIdx.x * 32 + (tid >> 3);
    float qr[D], acc[D];
    const float* qp = q + n * sq + (int64_t)min(l, L - 1) * ldq + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(qp + d);
#pragma unroll
        for (int e = 0; e < 4; ++e) qr[d + e] = t[e];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    float m = -INFINITY, lsum = 0.f;
    const int lrow = tid >> 2, lq = (tid & 3) * (D / 4);           // tile loader: 4 threads per key row, D/4 floats each
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int row = min(s0 + lrow, S - 1);
        const float* kp = k + nk * sk + (int64_t)row * ldk + h * D + lq;
        const float* vp = v + nk * sv + (int64_t)row * ldv + h * D + lq;
#pragma unroll
        for (int d = 0; d < D / 4; d += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(kp + d), b = *reinterpret_cast<const f32x4*>(vp + d);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ks[lrow * LD + lq + d + e] = a[e];
                vs[lrow * LD + lq + d + e] = b[e];
            }
        }
        __syncthreads();
        const int cnt = min(64, S - s0);
        for (int j = split; j < cnt; j += 8) {
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) dot = fmaf(qr[d], ks[j * LD + d], dot);
            const float sc = dot * scale;
            const float mn = fmaxf(m, sc);
            const float corr = expf(m - mn), p = expf(sc - mn);
            lsum = lsum * corr + p;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = fmaf(p, vs[j * LD + d], acc[d] * corr);
            m = mn;
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {                          // merge the 8 key splits of a query
        const float mo = __shfl_xor(m, off), lo = __shfl_xor(lsum, off);
        const float mn = fmaxf(m, mo);
        const float ca = m == -INFINITY ? 0.f : expf(m - mn), cb = mo == -INFINITY ? 0.f : expf(mo - mn);
        lsum = lsum * ca + lo * cb;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = acc[d] * ca + __shfl_xor(acc[d], off) * cb;
        m = mn;
    }
    if (l >= L) return;
    f32x4 t;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if ((d >> 2) == split) t[d & 3] = acc[d] / lsum;             // D = 32: split s owns channels 4s .. 4s + 3
    *reinterpret_cast<f32x4*>(out + n * so + (int64_t)l * ldo + h * D + split * 4) = t;
}

// ---------------------------------------------------------------- span (group) attention, one level
// One workgroup = one 2x2 query group g of the level map.  C = 256 channels = 8 heads x 32; thread c owns channel c
// for the gathers.  LDS: sampled rows [64][257] (keys, later reused for the values), the four query rows, the 4x8x64
// attention weights and the 64 sample descriptors.
constexpr int SP_C = 256, SP_M = 64, SP_LD = SP_C + 1;

struct SpanArgs {
    const float* q; int64_t ldq;          // this image's level map, row = y * w + x
    const float* k; int64_t ldk;          // the other image's level maps
    const float* v; int64_t ldv;
    const float* flow;                    // this image's full-resolution flow [H0*W0, 4] = (x, y, var_x, var_y)
    const float* sample_offset;           // [64, 2]
    float* out; int64_t ldo;              // [h*w, 256]; row g*4 + n (the reference views (g, n) as a raster index)
    int h, w, hk, wk, W0, win;            // level map sizes; full-resolution width; win = 2*s full-resolution cells per group side
    float inv_s, radius_scale, nsample1, scale;
    int64_t sq, sk, sv, sflow, so;        // batch strides (floats); blockIdx.y = image, its keys / values come from image ^ kv_swap
    int kv_swap;
};

// WIDE: the gathers as 16-byte loads (a lane owns four channels, a wave a sample row) with the 16 taps of four samples in
// flight per lane -- for launches of many groups (several pairs per pass), where the bytes in flight per CU set the rate; the
// narrow form (thread = channel, four 4-byte taps per trip) stays for unaligned operands.  Same products and sums per element.
template <bool WIDE>
__global__ __launch_bounds__(256) void span_attention_kernel(SpanArgs a) {
    extern __shared__ float smem[];
    float* rows = smem;                          // [64][257]
    float* qs = rows + SP_M * SP_LD;             // [4][256]
    float* att = qs + 4 * SP_C;                  // [4][8][64]
    float* sw = att + 4 * 8 * SP_M;              // [64][4] bilinear weights (nw, ne, sw, se; 0 where the corner is outside)
    int* si = reinterpret_cast<int*>(sw + SP_M * 4);   // [64][4] corner row indices (clamped)
    __shared__ float grp[4];                     // offset x, y, span x, y of the group
    {
        const int nb = blockIdx.y, nk = nb ^ a.kv_swap;
        a.q += nb * a.sq; a.k += nk * a.sk; a.v += nk * a.sv; a.flow += nb * a.sflow; a.out += nb * a.so;
    }
    const int tid = threadIdx.x, g = blockIdx.x, gw = a.w / 2;
    const int gy = g / gw, gx = g % gw;
    // avg_pool2d(offset, win) / s and avg_pool2d(span_scale, win) over the group's full-resolution cells: the per-cell terms
    // (two exps each) in parallel, the window sums in the reference's (ky, kx) order by one thread
    float* cell = att;                           // [win*win][4] scratch (att is not live yet)
    if (tid < a.win * a.win) {
        const float* f = a.flow + ((int64_t)(gy * a.win + tid / a.win) * a.W0 + gx * a.win + tid % a.win) * 4;
        const float vx = expf(0.5f * f[2]) * a.radius_scale, vy = expf(0.5f * f[3]) * a.radius_scale;
        cell[tid * 4 + 0] = f[0];
        cell[tid * 4 + 1] = f[1];
        cell[tid * 4 + 2] = fmaxf(vx * 2.f / a.nsample1, 1.f);
        cell[tid * 4 + 3] = fmaxf(vy * 2.f / a.nsample1, 1.f);
    }
    __syncthreads();
    if (tid == 0) {
        float ox = 0.f, oy = 0.f, sx = 0.f, sy = 0.f;
        for (int c = 0; c < a.win * a.win; ++c) {
            ox += cell[c * 4 + 0];
            oy += cell[c * 4 + 1];
            sx += cell[c * 4 + 2];
            sy += cell[c * 4 + 3];
        }
        const float d = (float)(a.win * a.win);
        grp[0] = ox / d * a.inv_s;
        grp[1] = oy / d * a.inv_s;
        grp[2] = sx / d;
        grp[3] = sy / d;
    }
    // the group's queries: pixel (gy*2 + iy, gx*2 + ix), n = iy*2 + ix
    for (int n = 0; n < 4; ++n)
        qs[n * SP_C + tid] = a.q[((int64_t)(gy * 2 + (n >> 1)) * a.w + gx * 2 + (n & 1)) * a.ldq + tid];
    __syncthreads();
    if (tid < SP_M) {
        // grid_sample(align_corners=False, padding zeros): normalise as the reference does, then un-normalise as ATen does
        const float px = grp[0] + a.sample_offset[tid * 2] * grp[2];
        const float py = grp[1] + a.sample_offset[tid * 2 + 1] * grp[3];
        const float nx = px / ((float)a.wk / 2.f) - 1.f, ny = py / ((float)a.hk / 2.f) - 1.f;
        const float ix = ((nx + 1.f) * (float)a.wk - 1.f) / 2.f, iy = ((ny + 1.f) * (float)a.hk - 1.f) / 2.f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = fx + 1.f - ix, wy0 = fy + 1.f - iy;
        const bool xa = x0 >= 0 && x0 < a.wk, xb = x0 + 1 >= 0 && x0 + 1 < a.wk;
        const bool ya = y0 >= 0 && y0 < a.hk, yb = y0 + 1 >= 0 && y0 + 1 < a.hk;
        const int cx0 = min(max(x0, 0), a.wk - 1), cx1 = min(max(x0 + 1, 0), a.wk - 1);
        const int cy0 = min(max(y0, 0), a.hk - 1), cy1 = min(max(y0 + 1, 0), a.hk - 1);
        sw[tid * 4 + 0] = xa && ya ? wx0 * wy0 : 0.f;
        sw[tid * 4 + 1] = xb && ya ? wx1 * wy0 : 0.f;
        sw[tid * 4 + 2] = xa && yb ? wx0 * wy1 : 0.f;
        sw[tid * 4 + 3] = xb && yb ? wx1 * wy1 : 0.f;
        si[tid * 4 + 0] = cy0 * a.wk + cx0;
        si[tid * 4 + 1] = cy0 * a.wk + cx1;
        si[tid * 4 + 2] = cy1 * a.wk + cx0;
        si[tid * 4 + 3] = cy1 * a.wk + cx1;
    }
    __syncthreads();
    auto gather = [&](const float* src, int64_t ld) {
#pragma unroll 1      // measured: deeper unrolling (more gathers in flight per lane) is 7 % slower end to end
        for (int m = 0; m < SP_M; ++m) {
            float r = src[(int64_t)si[m * 4 + 0] * ld + tid] * sw[m * 4 + 0];
            r += src[(int64_t)si[m * 4 + 1] * ld + tid] * sw[m * 4 + 1];
            r += src[(int64_t)si[m * 4 + 2] * ld + tid] * sw[m * 4 + 2];
            r += src[(int64_t)si[m * 4 + 3] * ld + tid] * sw[m * 4 + 3];
            rows[m * SP_LD + tid] = r;
        }
    };
    auto gather_wide = [&](const float* src, int64_t ld) {
        constexpr int U = 4;                     // samples per trip: 16 loads of 16 B in flight per lane (8 measured slower)
        const int wv = tid >> 6, ln = tid & 63;
#pragma unroll 1
        for (int m0 = wv * 16; m0 < wv * 16 + 16; m0 += U) {
            f32x4 t[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) t[u][c] = *reinterpret_cast<const f32x4*>(src + (int64_t)si[(m0 + u) * 4 + c] * ld + ln * 4);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = m0 + u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float r = t[u][0][e] * sw[m * 4 + 0];
                    r += t[u][1][e] * sw[m * 4 + 1];
                    r += t[u][2][e] * sw[m * 4 + 2];
                    r += t[u][3][e] * sw[m * 4 + 3];
                    rows[m * SP_LD + ln * 4 + e] = r;
                }
            }
        }
    };
    if constexpr (WIDE) gather_wide(a.k, a.ldk); else gather(a.k, a.ldk);
    __syncthreads();
    {   // QK[n][h][m]: thread = (n, m), the eight heads in sequence
        const int n = tid >> 6, m = tid & 63;
        for (int hd = 0; hd < 8; ++hd) {
            float dot = 0.f;
#pragma unroll 8
            for (int d = 0; d < 32; ++d) dot = fmaf(qs[n * SP_C + hd * 32 + d], rows[m * SP_LD + hd * 32 + d], dot);
            att[(n * 8 + hd) * SP_M + m] = dot * a.scale;
        }
    }
    __syncthreads();
    {   // softmax over the 64 samples of (n, head) = tid >> 3: 8 lanes x 8 samples, xor butterflies inside the 8-lane group
        float* p = att + (tid >> 3) * SP_M + (tid & 7) * 8;
        float e[8], mx = p[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) mx = fmaxf(mx, p[i]);
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            e[i] = expf(p[i] - mx);
            sum += e[i];
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) sum += __shfl_xor(sum, off);
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = e[i] / sum;
    }
    if constexpr (WIDE) gather_wide(a.v, a.ldv); else gather(a.v, a.ldv);      // rows <- sampled values (keys no longer needed)
    __syncthreads();
    const int hd = tid >> 5;
    for (int n = 0; n < 4; ++n) {
        float r = 0.f;
        for (int m = 0; m < SP_M; ++m) r = fmaf(att[(n * 8 + hd) * SP_M + m], rows[m * SP_LD + tid], r);
        a.out[((int64_t)g * 4 + n) * a.ldo + tid] = r;
    }
}

// ---------------------------------------------------------------- layernorm2d
// One wave per token, VEC = C/64 channels per lane.
template <int VEC>
__global__ __launch_bounds__(256) void layernorm2d_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ affine, const float* __restrict__ bias,
                                                          const _Float16* __restrict__ resh, const _Float16* __restrict__ resl,
                                                          int64_t ldr, float* __restrict__ out, int64_t ldo,
                                                          _Float16* __restrict__ outh, _Float16* __restrict__ outl,
                                                          int64_t ldos, int64_t rows) {
    constexpr int C = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = x[row * ldx + lane + 64 * e];
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
    const float den = sqrtf(q / (float)(C - 1)) + 1e-6f;        // torch.std: unbiased; eps on the std
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int c = lane + 64 * e;
        float o = affine[c] * (v[e] - mean) / den + bias[c];
        if (resh) o = ((float)resh[row * ldr + c] + (float)resl[row * ldr + c] * (1.f / 2048.f)) + o;
        if (out) out[row * ldo + c] = o;
        if (outh) {
            _Float16 hh, ll;
            split_f32(o, hh, ll);
            outh[row * ldos + c] = hh;
            outl[row * ldos + c] = ll;
        }
    }
}

// ---------------------------------------------------------------- bilinear xS (align_corners=False), nearest xS
__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ x, int64_t ldx, int hin, int win,
                                                                int C4, int s, float* __restrict__ out, int64_t ldo,
                                                                _Float16* __restrict__ outh, _Float16* __restrict__ outl,
                                                                int64_t ldos, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c4 = (int)(e % C4);
    int64_t t = e / C4;
    const int wout = win * s, hout = hin * s;
    const int ox = (int)(t % wout);
    t /= wout;
    const int oy = (int)(t % hout);
    const int64_t n = t / hout;
    const float rs = 1.f / (float)s;                                   // area_pixel_compute_scale with scale_factor given
    const float fy = fmaxf(rs * ((float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(rs * ((float)ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* b = x + n * hin * win * ldx + c4 * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((int64_t)y0 * win + x0) * ldx);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((int64_t)y0 * win + x1) * ldx);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((int64_t)y1 * win + x0) * ldx);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((int64_t)y1 * win + x1) * ldx);
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = ly0 * (lx0 * v00[q] + lx1 * v01[q]) + ly1 * (lx0 * v10[q] + lx1 * v11[q]);
    const int64_t row = (n * hout + oy) * wout + ox;
    store4(out, outh, outl, row * ldo + c4 * 4, row * ldos + c4 * 4, r);
}

__global__ __launch_bounds__(256) void upsample_nearest_kernel(const float* __restrict__ x, int64_t ldx, int hin, int win,
                                                               int C4, int s, float* __restrict__ out, int64_t ldo,
                                                               _Float16* __restrict__ outh, _Float16* __restrict__ outl,
                                                               int64_t ldos, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c4 = (int)(e % C4);
    int64_t t = e / C4;
    const int wout = win * s, hout = hin * s;
    const int ox = (int)(t % wout);
    t /= wout;
    const int oy = (int)(t % hout);
    const int64_t n = t / hout;
    const f32x4 r = *reinterpret_cast<const f32x4*>(x + ((n * hin + oy / s) * win + ox / s) * ldx + c4 * 4);
    const int64_t row = (n * hout + oy) * wout + ox;
    store4(out, outh, outl, row * ldo + c4 * 4, row * ldos + c4 * 4, r);
}

// F.interpolate(x, size=(hout, wout), mode='bilinear', align_corners=False) of single-channel planes: what
// torchvision 0.9.1's transforms.Resize does to a float tensor (aspanformer.py:131-139).  ATen's arithmetic: scale = in / out in
// fp32, src = max(scale * (dst + 0.5) - 0.5, 0), weights (1 - l, l), rows combined as wy0 * (wx0 a + wx1 b) + wy1 * (...).
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, int hin, int win, int hout, int wout,
                                                              float sy, float sx, float* __restrict__ out, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int ox = (int)(e % wout);
    int64_t t = e / wout;
    const int oy = (int)(t % hout);
    const int64_t n = t / hout;
    const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)fy, hin - 1), x0 = min((int)fx, win - 1);
    const int y1 = min(y0 + 1, hin - 1), x1 = min(x0 + 1, win - 1);
    const float ly1 = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), lx1 = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* b = x + n * hin * win;
    const float v00 = b[(int64_t)y0 * win + x0], v01 = b[(int64_t)y0 * win + x1];
    const float v10 = b[(int64_t)y1 * win + x0], v11 = b[(int64_t)y1 * win + x1];
    out[e] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
}

__global__ __launch_bounds__(256) void flow_decode_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, float wk,
                                                          float hk, float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx);
    f32x4 o;
    o[0] = 1.f / (1.f + expf(-v[0])) * wk;
    o[1] = 1.f / (1.f + expf(-v[1])) * hk;
    o[2] = v[2];
    o[3] = v[3];
    *reinterpret_cast<f32x4*>(out + r * 4) = o;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace

extern "C" int dfsfm_avgpool_nhwc_f32(const float* x, int64_t ldx, int N, int H, int W, int C, int k, float* out,
                                      int64_t ldo, void* stream_) {
    if (N == 0) return DFSFM_OK;
    if (!x || !out || N < 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || ldx < C || ldo < C) return DFSFM_E_BADARG;
    if (H % k || W % k || C % 4 || (ldx & 3) || (ldo & 3) || !al16(x) || !al16(out)) return DFSFM_E_UNSUPPORTED;
    const int64_t total = (int64_t)N * (H / k) * (W / k) * (C / 4);
    hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       x, ldx, H, W, C / 4, k, out, ldo, total);
    return dfsfm::check_launch("dfsfm_avgpool_nhwc_f32");
}

extern "C" int dfsfm_full_attention_f32(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk,
                                        const float* v, int64_t ldv, int64_t sv, float* out, int64_t ldo, int64_t so, int N,
                                        int L, int S, int H, int D, int kv_swap, float scale, void* stream_) {
    if (N == 0 || L == 0) return DFSFM_OK;
    if (!q || !k || !v || !out || N < 0 || L < 0 || S <= 0 || H <= 0 || (kv_swap != 0 && kv_swap != 1)) return DFSFM_E_BADARG;
    if (kv_swap && (N & 1)) return DFSFM_E_BADARG;
    if (D != 32) return DFSFM_E_UNSUPPORTED;
    if ((ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3) || (sq & 3) || (sk & 3) || (sv & 3) || (so & 3) || !al16(q) || !al16(k) ||
        !al16(v) || !al16(out))
        return DFSFM_E_UNSUPPORTED;
    hipLaunchKernelGGL(full_attention_kernel<32>, dim3((unsigned)((L + 31) / 32), (unsigned)H, (unsigned)N), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), q, ldq, sq, k, ldk, sk, v, ldv, sv, out, ldo, so, L, S, kv_swap, scale);
    return dfsfm::check_launch("dfsfm_full_attention_f32");
}

extern "C" int dfsfm_span_attention_f32(const float* q, int64_t ldq, int64_t sq, int h, int w, const float* k, int64_t ldk,
                                        int64_t sk, const float* v, int64_t ldv, int64_t sv, int hk, int wk, const float* flow,
                                        int H0, int W0, const float* sample_offset, int nhead, int C, int nsample0, int nsample1,
                                        float radius_scale, float temp, float* out, int64_t ldo, int N, int kv_swap,
                                        void* stream_) {
    if (N == 0) return DFSFM_OK;
    if (!q || !k || !v || !flow || !sample_offset || !out || h <= 0 || w <= 0 || hk <= 0 || wk <= 0 || H0 <= 0 || W0 <= 0 || N < 0)
        return DFSFM_E_BADARG;
    if (ldq < C || ldk < C || ldv < C || ldo < C || (kv_swap != 0 && kv_swap != 1) || (kv_swap && (N & 1))) return DFSFM_E_BADARG;
    if (C != SP_C || nhead != 8 || nsample0 != 2 || nsample1 != 8) return DFSFM_E_UNSUPPORTED;   // the released configuration
    if (h % 2 || w % 2 || H0 % h || W0 % w || H0 / h != W0 / w) return DFSFM_E_UNSUPPORTED;
    const int s = H0 / h;
    SpanArgs a;
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.flow = flow; a.sample_offset = sample_offset;
    a.out = out; a.ldo = ldo; a.h = h; a.w = w; a.hk = hk; a.wk = wk; a.W0 = W0; a.win = 2 * s;
    a.inv_s = 1.f / (float)s; a.radius_scale = radius_scale; a.nsample1 = (float)nsample1;
    a.scale = temp / sqrtf((float)(C / nhead));
    a.sq = sq; a.sk = sk; a.sv = sv; a.sflow = (int64_t)H0 * W0 * 4; a.so = (int64_t)h * w * ldo; a.kv_swap = kv_swap;
    const int smem = (SP_M * SP_LD + 4 * SP_C + 4 * 8 * SP_M + SP_M * 4 + SP_M * 4) * 4;
    static dfsfm::SmemAttr attr, attr_w;
    const bool aligned = ((reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) == 0 && ldk % 4 == 0 && ldv % 4 == 0 &&
                         sk % 4 == 0 && sv % 4 == 0;
    const dim3 grid((unsigned)((h / 2) * (w / 2)), (unsigned)N);
    if (aligned) {      // 16-byte gathers; rows that are not 16-byte aligned (column slices at odd offsets) take the 4-byte kernel
        attr_w.ensure(reinterpret_cast<const void*>(span_attention_kernel<true>), smem);
        hipLaunchKernelGGL(span_attention_kernel<true>, grid, dim3(256), smem, static_cast<hipStream_t>(stream_), a);
    } else {
        attr.ensure(reinterpret_cast<const void*>(span_attention_kernel<false>), smem);
        hipLaunchKernelGGL(span_attention_kernel<false>, grid, dim3(256), smem, static_cast<hipStream_t>(stream_), a);
    }
    return dfsfm::check_launch("dfsfm_span_attention_f32");
}

extern "C" int dfsfm_layernorm2d_f32(const float* x, int64_t ldx, const float* affine, const float* bias, const void* res_hi,
                                     const void* res_lo, int64_t ldr, float* out, int64_t ldo, void* out_hi, void* out_lo,
                                     int64_t ldo_s, int64_t rows, int C, void* stream_) {
    if (rows == 0) return DFSFM_OK;
    if (!x || !affine || !bias || (!out && !out_hi) || rows < 0 || ldx < C) return DFSFM_E_BADARG;
    if ((res_hi == nullptr) != (res_lo == nullptr) || (out_hi == nullptr) != (out_lo == nullptr)) return DFSFM_E_BADARG;
    if ((out && ldo < C) || (out_hi && ldo_s < C) || (res_hi && ldr < C)) return DFSFM_E_BADARG;
    const _Float16* rh = static_cast<const _Float16*>(res_hi);
    const _Float16* rl = static_cast<const _Float16*>(res_lo);
    _Float16* oh = static_cast<_Float16*>(out_hi);
    _Float16* ol = static_cast<_Float16*>(out_lo);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
    if (C == 256) hipLaunchKernelGGL(layernorm2d_kernel<4>, grid, blk, 0, stream, x, ldx, affine, bias, rh, rl, ldr, out, ldo, oh, ol, ldo_s, rows);
    else if (C == 384) hipLaunchKernelGGL(layernorm2d_kernel<6>, grid, blk, 0, stream, x, ldx, affine, bias, rh, rl, ldr, out, ldo, oh, ol, ldo_s, rows);
    else return DFSFM_E_UNSUPPORTED;
    return dfsfm::check_launch("dfsfm_layernorm2d_f32");
}

extern "C" int dfsfm_upsample_nhwc_f32(const float* x, int64_t ldx, int N, int hin, int win, int C, int scale, int bilinear,
                                       float* out, int64_t ldo, void* out_hi, void* out_lo, int64_t ldo_s, void* stream_) {
    if (N == 0) return DFSFM_OK;
    if (!x || (!out && !out_hi) || N < 0 || hin <= 0 || win <= 0 || C <= 0 || scale <= 0 || ldx < C) return DFSFM_E_BADARG;
    if ((out_hi == nullptr) != (out_lo == nullptr) || (out && ldo < C) || (out_hi && ldo_s < C)) return DFSFM_E_BADARG;
    if (C % 4 || (ldx & 3) || (out && ((ldo & 3) || !al16(out))) || (out_hi && ((ldo_s & 3) || !al8(out_hi) || !al8(out_lo))) || !al16(x))
        return DFSFM_E_UNSUPPORTED;
    const int64_t total = (int64_t)N * hin * scale * win * scale * (C / 4);
    _Float16* oh = static_cast<_Float16*>(out_hi);
    _Float16* ol = static_cast<_Float16*>(out_lo);
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (bilinear) hipLaunchKernelGGL(upsample_bilinear_kernel, grid, blk, 0, stream, x, ldx, hin, win, C / 4, scale, out, ldo, oh, ol, ldo_s, total);
    else hipLaunchKernelGGL(upsample_nearest_kernel, grid, blk, 0, stream, x, ldx, hin, win, C / 4, scale, out, ldo, oh, ol, ldo_s, total);
    return dfsfm::check_launch("dfsfm_upsample_nhwc_f32");
}

extern "C" int dfsfm_flow_decode_f32(const float* x, int64_t ldx, int64_t rows, float wk, float hk, float* out, void* stream_) {
    if (rows == 0) return DFSFM_OK;
    if (!x || !out || rows < 0 || ldx < 4) return DFSFM_E_BADARG;
    if ((ldx & 3) || !al16(x) || !al16(out)) return DFSFM_E_UNSUPPORTED;
    hipLaunchKernelGGL(flow_decode_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       x, ldx, rows, wk, hk, out);
    return dfsfm::check_launch("dfsfm_flow_decode_f32");
}

extern "C" int dfsfm_resize_bilinear_f32(const float* x, int N, int hin, int win, int hout, int wout, float* out, void* stream_) {
    if (N == 0) return DFSFM_OK;
    if (!x || !out || N < 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0) return DFSFM_E_BADARG;
    const float sy = (float)hin / (float)hout, sx = (float)win / (float)wout;        // area_pixel_compute_scale, size given
    const int64_t total = (int64_t)N * hout * wout;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       x, hin, win, hout, wout, sy, sx, out, total);
    return dfsfm::check_launch("dfsfm_resize_bilinear_f32");
}
