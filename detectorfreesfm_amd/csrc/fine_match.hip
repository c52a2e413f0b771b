// K11+K12 -- fine-window correlation, softmax expectation, best-candidate selection and refined
// keypoints for gfx950 (MI355X).
//
// Replaces FineMatching.forward of the reference in its test configuration
//   src/MultiviewMatcher/utils/fine_matching.py:36-98 (forward), :100-119 (select_left_point),
//   :195-219 (_s2d_heatmap), :258-285 (argsoftmax), :129-179 (_obtain_left_normalized_offset),
//   :221-252 (build_moved_query / build_mkpts)
//
// One workgroup per feature track; 576 KB of features in, < 100 B out -> HBM-bound at the fp32
// ridge (AI ~ 20 flop/B).  Wave w owns query views w, w+4, ...: it streams the view's W*W x C
// window from HBM straight into MFMA A-fragments (each lane reads 256 contiguous bytes of one
// row per 32-row tile), multiplies against the <= 64 candidate rows of the reference window held
// in LDS (v_mfma_f32_32x32x2_f32; sim^T[r][l], so the softmax axis r is lane-local), and folds
// every 32-row tile into an online softmax carrying the five moments (sum e, e*gx, e*gy, e*gx^2,
// e*gy^2).  The heat-map is never materialised.  Lane halves are merged with one shuffle; the
// masked mean over views and the first-minimum argmin over candidates run in wave 0.
#include "common.h"

#pragma clang fp contract(off)

namespace {

using namespace dfsfm;

constexpr int MAXL = 64;       // candidate rows (left*left <= 64)
constexpr int MAXWW = 256;     // window positions

struct FineArgs {
    const float* ref;      // [T][WW][C]
    const float* qry;      // [T][Vq][WW][C]
    const uint8_t* track_mask;
    const uint8_t* movable;
    int T, Vq, W, left;
    const float* query_pts;
    const float* scale_q;
    const float* ref_pts;
    const float* scale_r;
    int64_t rs_t, rs_n;
    int32_t* best_index;
    float* left_norm;
    float* coords;
    float* stdv;
    float* query_refined;
    float* ref_refined;
};

struct Moments {
    float m, s0, sx, sy, sxx, syy;
};

template <int C>
__global__ __launch_bounds__(256, 1) void fine_match_kernel(FineArgs g) {
    constexpr int KH = C / 2;            // k values owned by one lane half
    constexpr int NQ = KH / 4;           // float4 per lane per row
    constexpr int REF_LD = C + 4;        // pad: conflict-free ds_read_b128 across rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_ref = reinterpret_cast<float*>(smem);                    // [MAXL][REF_LD]
    float2* s_grid = reinterpret_cast<float2*>(s_ref + MAXL * REF_LD);  // [MAXWW]
    float* s_res = reinterpret_cast<float*>(s_grid + MAXWW);          // [Vq][MAXL][3]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int t = blockIdx.x;
    const int W = g.W, WW = W * W, left = g.left, L = left * left, Vq = g.Vq;
    const int NT = (WW + 31) / 32;       // 32-row tiles of the window

    // candidate rows: centre left x left window of the reference patch (select_left_point)
    {
        const int c0 = W / 2 - left / 2;
        const float* rbase = g.ref + (int64_t)t * WW * C;
        for (int e = tid; e < MAXL * (C / 4); e += 256) {
            const int l = e / (C / 4), c4 = e % (C / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (l < L) {
                const int r = (c0 + l / left) * W + c0 + l % left;
                v = *reinterpret_cast<const f32x4*>(rbase + (int64_t)r * C + c4 * 4);
            }
            *reinterpret_cast<f32x4*>(s_ref + l * REF_LD + c4 * 4) = v;
        }
        // kornia create_meshgrid(W, W, normalized): (x / (W-1) - 0.5) * 2
        for (int r = tid; r < MAXWW; r += 256) {
            const float gx = ((float)(r % W) / (float)(W - 1) - 0.5f) * 2.f;
            const float gy = ((float)(r / W) / (float)(W - 1) - 0.5f) * 2.f;
            s_grid[r] = make_float2(gx, gy);
        }
    }
    __syncthreads();

    const float temp = (float)(1.0 / sqrt((double)C));   // softmax_temp = 1 / C**.5 (python double -> f32)
    for (int n = wave; n < Vq; n += 4) {
        const float* qbase = g.qry + ((int64_t)t * Vq + n) * WW * C;
        Moments st[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) st[b] = Moments{-INFINITY, 0.f, 0.f, 0.f, 0.f, 0.f};

        f32x4 a_cur[NQ], a_nxt[NQ];
        auto load_tile = [&](f32x4* dst, int rt) {
            const int r = rt * 32 + col;
#pragma unroll
            for (int qd = 0; qd < NQ; ++qd) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                dst[qd] = r < WW ? *reinterpret_cast<const f32x4*>(qbase + (int64_t)r * C + half * KH + qd * 4) : z;
            }
        };
        load_tile(a_cur, 0);
        for (int rt = 0; rt < NT; ++rt) {
            if (rt + 1 < NT) load_tile(a_nxt, rt + 1);
            f32x16 acc[2] = {f32x16{0}, f32x16{0}};
#pragma unroll
            for (int qd = 0; qd < NQ; ++qd) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_ref + col * REF_LD + half * KH + qd * 4);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(s_ref + (32 + col) * REF_LD + half * KH + qd * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {   // A[i=r][k=c] = qry[r][c], B[k=c][j=l] = ref[l][c]
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[qd][e], b0[e], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[qd][e], b1[e], acc[1], 0, 0, 0);
                }
            }
            // online softmax over this tile's rows r = rt*32 + mfma32_row(reg, half)
            const int rbase_t = rt * 32 + 4 * half;
            if (rbase_t < WW) {   // at least one valid row in this lane half
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float x[16];
                    float tmax = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = rt * 32 + mfma32_row(r, half);
                        x[r] = rr < WW ? temp * acc[b][r] : -INFINITY;
                        tmax = fmaxf(tmax, x[r]);
                    }
                    const float m_new = fmaxf(st[b].m, tmax);
                    const float sc = expf(st[b].m - m_new);     // exp(-inf) = 0 on the first tile
                    float s0 = st[b].s0 * sc, sx = st[b].sx * sc, sy = st[b].sy * sc;
                    float sxx = st[b].sxx * sc, syy = st[b].syy * sc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = rt * 32 + mfma32_row(r, half);
                        if (rr < WW) {
                            const float e = expf(x[r] - m_new);
                            const float2 gxy = s_grid[rr];
                            s0 += e;
                            sx += e * gxy.x;
                            sy += e * gxy.y;
                            sxx += e * (gxy.x * gxy.x);
                            syy += e * (gxy.y * gxy.y);
                        }
                    }
                    st[b] = Moments{m_new, s0, sx, sy, sxx, syy};
                }
            }
            if (rt + 1 < NT) {
#pragma unroll
                for (int qd = 0; qd < NQ; ++qd) a_cur[qd] = a_nxt[qd];
            }
        }
        // merge the two lane halves (rows 4*half + ...) and finish: expectation and std
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float m_o = __shfl_xor(st[b].m, 32);
            const float m_all = fmaxf(st[b].m, m_o);
            const float sc = expf(st[b].m - m_all);
            float s0 = st[b].s0 * sc, sx = st[b].sx * sc, sy = st[b].sy * sc;
            float sxx = st[b].sxx * sc, syy = st[b].syy * sc;
            s0 += __shfl_xor(s0, 32);
            sx += __shfl_xor(sx, 32);
            sy += __shfl_xor(sy, 32);
            sxx += __shfl_xor(sxx, 32);
            syy += __shfl_xor(syy, 32);
            const float ex = sx / s0, ey = sy / s0;
            const float vx = sxx / s0 - ex * ex, vy = syy / s0 - ey * ey;
            const float sd = sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f));
            if (half == 0) {
                float* p = s_res + ((int64_t)n * MAXL + b * 32 + col) * 3;
                p[0] = ex;
                p[1] = ey;
                p[2] = sd;
            }
        }
    }
    __syncthreads();

    if (wave == 0) {
        // score_l = masked mean over views of std (masked_mean, fine_matching.py:254-256)
        const uint8_t* tm = g.track_mask + (int64_t)t * Vq;
        float score = INFINITY;
        if (lane < L) {
            float num = 0.f, den = 0.f;
            for (int n = 0; n < Vq; ++n) {
                const float mk = tm[n] ? 1.f : 0.f;
                num += mk * s_res[((int64_t)n * MAXL + lane) * 3 + 2];
                den += mk;
            }
            score = num / fmaxf(den, 1.f);
        }
        // first minimum over candidates (torch.min returns the lowest index among ties)
        float bs = score;
        int bi = lane;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int oi = __shfl_xor(bi, off);
            if (os < bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
        }
        const bool mov = g.movable ? g.movable[t] != 0 : true;
        const int best = mov ? bi : L / 2;
        const float lx = ((float)(best % left) / (float)(left - 1)) * 2.f - 1.f;
        const float ly = ((float)(best / left) / (float)(left - 1)) * 2.f - 1.f;
        if (lane == 0) {
            if (g.best_index) g.best_index[t] = best;
            if (g.left_norm) { g.left_norm[t * 2] = lx; g.left_norm[t * 2 + 1] = ly; }
            if (g.query_refined) {
                const float wsz = (float)(left / 2);
                g.query_refined[t * 2 + 0] = g.query_pts[t * 2 + 0] + (lx * wsz) * g.scale_q[t * 2 + 0];
                g.query_refined[t * 2 + 1] = g.query_pts[t * 2 + 1] + (ly * wsz) * g.scale_q[t * 2 + 1];
            }
        }
        for (int n = lane; n < Vq; n += 64) {
            const float* p = s_res + ((int64_t)n * MAXL + best) * 3;
            const int64_t o = (int64_t)t * Vq + n;
            if (g.coords) { g.coords[o * 2] = p[0]; g.coords[o * 2 + 1] = p[1]; }
            if (g.stdv) g.stdv[o] = p[2];
            if (g.ref_refined) {
                const float wsz = (float)(W / 2);
                const int64_t a = ((int64_t)t * g.rs_t + (int64_t)n * g.rs_n) * 2;
                g.ref_refined[o * 2 + 0] = g.ref_pts[a + 0] + (p[0] * wsz) * g.scale_r[a + 0];
                g.ref_refined[o * 2 + 1] = g.ref_pts[a + 1] + (p[1] * wsz) * g.scale_r[a + 1];
            }
        }
    }
}

template <int C>
void launch(const FineArgs& g, hipStream_t stream) {
    const size_t smem = (size_t)MAXL * (C + 4) * 4 + MAXWW * 8 + (size_t)g.Vq * MAXL * 3 * 4;
    static dfsfm::SmemAttr smem_attr;
    smem_attr.ensure(reinterpret_cast<const void*>(&fine_match_kernel<C>), 96 * 1024);
    hipLaunchKernelGGL((fine_match_kernel<C>), dim3(g.T), dim3(256), smem, stream, g);
}

}  // namespace

extern "C" int dfsfm_fine_match_f32(const float* ref, const float* qry, const uint8_t* track_mask,
                                    const uint8_t* movable, int T, int Vq, int W, int left, int C,
                                    const float* query_pts, const float* scale_q, const float* ref_pts,
                                    const float* scale_r, int64_t rs_t, int64_t rs_n, int32_t* best_index,
                                    float* left_norm, float* coords, float* std, float* query_refined,
                                    float* ref_refined, void* stream_) {
    if (T == 0) return DFSFM_OK;
    if (!ref || !qry || !track_mask) return DFSFM_E_BADARG;
    if (T < 0 || Vq <= 0 || W <= 1 || left <= 1) return DFSFM_E_BADARG;
    if (query_refined && (!query_pts || !scale_q)) return DFSFM_E_BADARG;
    if (ref_refined && (!ref_pts || !scale_r)) return DFSFM_E_BADARG;
    if (left > W || left * left > MAXL || W * W > MAXWW || (left & 1) == 0 || (W & 1) == 0) return DFSFM_E_UNSUPPORTED;
    if (Vq > 64) return DFSFM_E_UNSUPPORTED;   // s_res LDS budget
    if ((reinterpret_cast<uintptr_t>(ref) & 15) || (reinterpret_cast<uintptr_t>(qry) & 15)) return DFSFM_E_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    FineArgs g{ref, qry, track_mask, movable, T, Vq, W, left, query_pts, scale_q, ref_pts, scale_r,
               rs_t, rs_n, best_index, left_norm, coords, std, query_refined, ref_refined};
    if (C == 128) launch<128>(g, stream);
    else if (C == 64) launch<64>(g, stream);
    else return DFSFM_E_UNSUPPORTED;
    return dfsfm::check_launch("dfsfm_fine_match_f32");
}
