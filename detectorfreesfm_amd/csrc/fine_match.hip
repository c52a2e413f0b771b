// K11+K12 -- fine-window correlation, softmax expectation, best-candidate selection and refined
// keypoints for gfx950 (MI355X).
//
// Replaces FineMatching.forward of the reference in its test configuration
//   src/MultiviewMatcher/utils/fine_matching.py:36-98 (forward), :100-119 (select_left_point),
//   :195-219 (_s2d_heatmap), :258-285 (argsoftmax), :129-179 (_obtain_left_normalized_offset),
//   :221-252 (build_moved_query / build_mkpts)
//
// Two input forms: fp32 tensors (dfsfm_fine_match_f32) or the fp16x2-split planes the encoder kernels write
// (dfsfm_fine_match_split: value = hi + lo/2048; same bytes, no conversion work in the stream).
// One workgroup per feature track; 486 KB of features in (the query windows + the 49 candidate rows),
// < 100 B out -> HBM-bound.  Wave w owns query views w, w+4, ...: it DMAs the view's W*W x C window
// through a private two-stage LDS ring (whole 256-byte row segments), multiplies 32-row tiles against
// the <= 64 candidate rows of the reference window (fp16x2-split planes in LDS; three
// v_mfma_f32_32x32x16_f16 per product; sim^T[r][l], so the softmax axis r is lane-local), and folds
// every tile into an online softmax carrying the five moments (sum e, e*gx, e*gy, e*gx^2, e*gy^2).
// The heat-map is never materialised.  Lane halves are merged with one shuffle; the masked mean over
// views and the first-minimum argmin over candidates run in wave 0.
#include "common.h"

#pragma clang fp contract(off)

namespace {

using namespace dfsfm;

constexpr int MAXL = 64;       // candidate slots (two 32-column MFMA blocks; the stride of the per-view result table)
constexpr int MAXLR = 52;      // candidate rows actually STORED: left is odd and left * left <= 49; slots beyond read row 51
constexpr int MAXWW = 256;     // window positions

struct FineArgs {
    const float* ref;      // [T][WW][C]
    const float* qry;      // [T][Vq][WW][C]
    const _Float16 *ref_h, *ref_l, *qry_h, *qry_l;     // the same tensors as split planes (SPLIT kernels)
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

// v2 schedule.  The first version streamed the query windows straight into fp32-MFMA fragments: 16-byte pieces of 64
// different rows per load instruction (a quarter of the L1 / TA rate), 128 fragment registers that pinned it to one
// wave per SIMD, and 64-cycle fp32 MFMAs -- 1.6 TB/s.  Now every wave DMAs its view's window into a private LDS ring
// in whole 256-byte row segments (buffer_load ... lds, swizzle on the source address, zero fill past the window),
// reads the fragments back conflict-free, splits them into fp16 hi / lo planes in registers (v = hi + lo/2048, the
// representation of every other GEMM of the path) and runs three fp16 MFMAs per product: fp32-class similarities at
// 3/16 of the fp32-MFMA cost.  No workgroup barrier in the stream: ring, DMA queue and vmcnt are per wave.
// v3: (a) split-plane input: the windows are DMA'd as 128-byte segments of the hi and of the lo plane (a stage is still 32 rows
// x 64 channels = 8 KB) and the fragments are read back ready for the MFMA -- the register split was 6 VALU operations per
// value, as many cycles as the stage's 24 MFMAs; (b) the softmax exponentials on the compensated hardware exp2 (exp_neg):
// with one wave per SIMD nothing hides VALU time, and expf was the larger half of it; (c) a wave requests the first two
// stages of its first view before the workgroup stages the candidate rows, so that latency runs under the prologue.
// (d) r04: 4-KB stages (KC = 32) and 52 stored candidate rows = 79 KB of LDS per workgroup (Vq <= 5): TWO workgroups per CU, two
// waves per SIMD -- one wave's DMA issue, LDS reads and softmax VALU work run under the other's MFMAs: 0.305 -> 0.218 ms per 2000
// tracks (4.5 TB/s).  r03 had measured that schedule and dropped it because 1-14 of 2000 tracks changed from run to run; r04 found that
// the deviations vanish when the file is built without the SLP vectoriser (and called packed fp32 unreliable beside MFMAs), r05 showed
// that plain packed fp32 is reliable there.  THE CAUSE (r06, profiles/r06_fine_match_bisect.txt): the vectoriser had turned the moment
// update of candidate block 0 into packed instructions, one of them `v_pk_mul_f32 v[152:153], v[202:203], v[152:153] op_sel:[0,1]` --
// the LOW lane reads the HIGH register of src1.  On gfx950 that form reads the operand as 0 in ~3e-3 of its executions while an MFMA the
// same wave issued earlier is still in flight (tools/ubench/pk_opsel_inplace.hip; no other packed form does, and not without own
// MFMAs).  Block 0's update runs while the last MFMA of block 1 can still be queued behind the OTHER workgroup's MFMAs -- hence only
// with two workgroups per CU, only in the second moments (one e * g^2 term of 225 lost), and not in block 1's twin of the instruction.
// Rewriting only the four op_sel instructions of the SLP build's assembly in place makes it bit-reproducible; rewriting the other
// 175 does not.  Rule (csrc/Makefile): this file is built without the vectoriser (also 7 % faster: packed fp32 is an anti-lever beside
// MFMAs), and the build fails if a packed-fp32 instruction with op_sel on src1 / src2 appears in any translation unit with MFMAs.
template <int C, bool SPLIT>
__global__ __launch_bounds__(256, 2) void fine_match_kernel(FineArgs g) {
    constexpr int KC = 32;                   // channels per stage: 4-KB stages, 79 KB of LDS per workgroup (Vq <= 5), two per CU
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int NKH = C / KC;              // DMA stages per 32-row tile
    constexpr int SLOTS = C / 8;             // 16-byte slots of one fp16 reference row
    constexpr int STAGE = 32 * KC * 4;       // one A stage: 32 rows x KC channels (fp32, or fp16 hi plane + fp16 lo plane)
    constexpr int PIECES = STAGE / 1024;     // DMA instructions per stage
    constexpr int NSTG = 3;                  // ring depth per wave: two stages in flight while one is multiplied
    // (d) above: two workgroups per CU; this translation unit is built without the SLP vectoriser and the Makefile checks its ISA
    // for the packed op_sel form that misreads beside in-flight MFMAs.
    constexpr int RB = SPLIT ? KC * 2 : KC * 4;          // bytes of a row inside a stage (per plane)
    constexpr int NS = RB / 16;                          // its 16-byte slots: 16, 8 or 4
    constexpr int PLANE = 32 * RB;                       // split input: bytes of one plane of a stage
    // slot swizzle by row so that the fragment-shaped reads of 16 consecutive rows cover all 16 bank groups of 256 B
    auto aswz = [](int r) __attribute__((always_inline)) { return NS == 16 ? (r & 15) : NS == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    // slot swizzle of a reference row: 16 slots (C = 128, 256-byte rows) -> row & 15; 8 slots (C = 64) -> (row >> 1) & 7
    auto rswz = [](int l) __attribute__((always_inline)) { return C == 128 ? (l & 15) : ((l >> 1) & 7); };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* s_rh = reinterpret_cast<_Float16*>(smem);                       // [MAXL][C] hi plane, slot-swizzled; rows >= L are zero
    _Float16* s_rl = s_rh + MAXLR * C;                                        // lo plane
    float2* s_grid = reinterpret_cast<float2*>(s_rl + MAXLR * C);             // [MAXWW]
    float* s_res = reinterpret_cast<float*>(s_grid + MAXWW);                  // [Vq][MAXL][3]
    char* s_ring = reinterpret_cast<char*>(s_res) + ((g.Vq * MAXL * 3 * 4 + 15) & ~15);   // [4 waves][NSTG stages][STAGE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int t = blockIdx.x;
    const int W = g.W, WW = W * W, left = g.left, L = left * left, Vq = g.Vq;
    const int NT = (WW + 31) / 32;           // 32-row tiles of the window
    const int nstage = NT * NKH;

    char* ring = s_ring + wave * NSTG * STAGE;
    // DMA lane geometry: one instruction = 1 KB = 64 / NS whole row segments of RB bytes (of one plane for split input, whose
    // stage is [hi: 32 rows x RB][lo: the same]).  lane -> (row in piece, physical slot); the logical slot is on the source.
    const int drow = lane / NS, dps = lane % NS;
    __amdgpu_buffer_rsrc_t rq, rql;          // the current view's window (fp32, or hi plane) / its lo plane
    auto open_view = [&](int n) __attribute__((always_inline)) {
        const int64_t o = ((int64_t)t * Vq + n) * WW * C;
        if constexpr (SPLIT) {
            rq = __builtin_amdgcn_make_buffer_rsrc((void*)(g.qry_h + o), 0, WW * C * 2, 0x00020000);
            rql = __builtin_amdgcn_make_buffer_rsrc((void*)(g.qry_l + o), 0, WW * C * 2, 0x00020000);
        } else {
            rq = __builtin_amdgcn_make_buffer_rsrc((void*)(g.qry + o), 0, WW * C * 4, 0x00020000);
            rql = rq;
        }
    };
    auto issue = [&](int s) __attribute__((always_inline)) {          // stage s = (tile s / NKH, segment s % NKH)
        const int rt = s / NKH, kh = s - rt * NKH;
        char* dst = ring + (s % NSTG) * STAGE;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            // rows past the window (and stages past the end) fall outside the descriptor: zero fill, no traffic
            constexpr int PP = SPLIT ? PIECES / 2 : PIECES;           // pieces per plane
            const int row = (p % PP) * (64 / NS) + drow;
            const int elem = (rt * 32 + row) * C + kh * KC + (dps ^ aswz(row)) * (SPLIT ? 8 : 4);
            const unsigned off = s < nstage ? (unsigned)(elem * (SPLIT ? 2 : 4)) : 0xFFFFFF00u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p < PP ? rq : rql, (lds_void*)(dst + p * 1024), 16, off, 0, 0, 0);
        }
    };
    if (wave < Vq) {                          // the first view's first two stages fly while the candidate rows are staged
        open_view(wave);
        issue(0);
        issue(1);
    }

    // candidate rows: centre left x left window of the reference patch (select_left_point), split once
    {
        const int c0 = W / 2 - left / 2;
        const float* rbase = g.ref + (int64_t)t * WW * C;
        for (int e = tid; e < MAXLR * SLOTS; e += 256) {
            const int l = e / SLOTS, q = e % SLOTS;
            half8 h = {0, 0, 0, 0, 0, 0, 0, 0}, lo = h;
            if (l < L && SPLIT) {
                const int64_t o = ((int64_t)t * WW + (c0 + l / left) * W + c0 + l % left) * C + q * 8;
                h = *reinterpret_cast<const half8*>(g.ref_h + o);
                lo = *reinterpret_cast<const half8*>(g.ref_l + o);
            } else if (l < L) {
                const int r = (c0 + l / left) * W + c0 + l % left;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(rbase + (int64_t)r * C + q * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(rbase + (int64_t)r * C + q * 8 + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    _Float16 a, b;
                    split_f32(v0[k], a, b); h[k] = a; lo[k] = b;
                    split_f32(v1[k], a, b); h[4 + k] = a; lo[4 + k] = b;
                }
            }
            const int o = l * C + ((q ^ rswz(l)) * 8);       // XOR swizzle: fragment reads of 16 rows hit 16 bank groups
            *reinterpret_cast<half8*>(s_rh + o) = h;
            *reinterpret_cast<half8*>(s_rl + o) = lo;
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
        Moments st[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) st[b] = Moments{-INFINITY, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (n != wave) {
            open_view(n);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // previous view's fragment reads are done
            issue(0);
            issue(1);
        }
        f32x16 accm[2], accx[2];
        for (int s = 0; s < nstage; ++s) {
            const int rt = s / NKH, kh = s - rt * NKH;
            issue(s + 2);                                             // into the stage consumed in the previous iteration
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");   // stage s has landed (s+1, s+2 may still fly)
            if (kh == 0) {
#pragma unroll
                for (int b = 0; b < 2; ++b) { accm[b] = f32x16{0}; accx[b] = f32x16{0}; }
            }
            const char* sa = ring + (s % NSTG) * STAGE + col * RB;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {                    // 16 channels per MFMA k-step, 8 per lane half
                const int q = (kh * KC + ks * 16 + half * 8) / 8;     // fp16 slot of the reference rows
                const half8 bh0 = *reinterpret_cast<const half8*>(s_rh + col * C + ((q ^ rswz(col)) * 8));
                const half8 bl0 = *reinterpret_cast<const half8*>(s_rl + col * C + ((q ^ rswz(col)) * 8));
const int r1 = MAXLR < MAXL ? min(32 + col, MAXLR - 1) : 32 + col;
                const half8 bh1 = *reinterpret_cast<const half8*>(s_rh + r1 * C + ((q ^ rswz(r1)) * 8));
                const half8 bl1 = *reinterpret_cast<const half8*>(s_rl + r1 * C + ((q ^ rswz(r1)) * 8));
                half8 ah, al;
                if constexpr (SPLIT) {
                    const int so = (((ks * 2 + half) ^ aswz(col)) * 16);
                    ah = *reinterpret_cast<const half8*>(sa + so);
                    al = *reinterpret_cast<const half8*>(sa + PLANE + so);
                } else {
                    const int sl = (ks * 2 + half) * 2;               // first of the two fp32 slots
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(sa + ((sl ^ aswz(col)) * 16));
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(sa + (((sl + 1) ^ aswz(col)) * 16));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        _Float16 x, y;
                        split_f32(a0[k], x, y); ah[k] = x; al[k] = y;
                        split_f32(a1[k], x, y); ah[4 + k] = x; al[4 + k] = y;
                    }
                }
                // A[i = window row][k], B[k][j = candidate]: sim^T keeps the softmax axis (rows) lane-local
                accm[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh0, accm[0], 0, 0, 0);
                accm[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh1, accm[1], 0, 0, 0);
                accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl0, accx[0], 0, 0, 0);
                accx[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl1, accx[1], 0, 0, 0);
                accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh0, accx[0], 0, 0, 0);
                accx[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh1, accx[1], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // reads of this stage retired before it is refilled
            if (kh != NKH - 1) continue;
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
                        x[r] = rr < WW ? temp * (accm[b][r] + accx[b][r] * (1.f / 2048.f)) : -INFINITY;
                        tmax = fmaxf(tmax, x[r]);
                    }
                    const float m_new = fmaxf(st[b].m, tmax);
                    const float sc = exp_neg(st[b].m - m_new);                    // exp(-inf) = 0 on the first tile
                    float s0 = st[b].s0 * sc, sx = st[b].sx * sc, sy = st[b].sy * sc;
                    float sxx = st[b].sxx * sc, syy = st[b].syy * sc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {          // rows past the window carry x = -inf -> e = 0: no branch
                        const int rr = rt * 32 + mfma32_row(r, half);
                        const float e = exp_neg(x[r] - m_new);
                        const float2 gxy = s_grid[rr];       // rr < MAXWW always
                        s0 += e;
                        sx += e * gxy.x;
                        sy += e * gxy.y;
                        sxx += e * (gxy.x * gxy.x);
                        syy += e * (gxy.y * gxy.y);
                    }
                    st[b] = Moments{m_new, s0, sx, sy, sxx, syy};
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the zero-fill tail stage
        // merge the two lane halves (rows 4*half + ...) and finish: expectation and std
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float m_o = from_xor32(st[b].m, lane);
            const float m_all = fmaxf(st[b].m, m_o);
            const float sc = expf(st[b].m - m_all);
            float s0 = st[b].s0 * sc, sx = st[b].sx * sc, sy = st[b].sy * sc;
            float sxx = st[b].sxx * sc, syy = st[b].syy * sc;
            s0 = add_xor32(s0);
            sx = add_xor32(sx);
            sy = add_xor32(sy);
            sxx = add_xor32(sxx);
            syy = add_xor32(syy);
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

// dynamic LDS of one workgroup: candidate planes (hi, lo) + grid + per-view results + 4 waves x 3 stages x 8 KB of DMA ring
inline size_t fine_smem_bytes(int C, int Vq) {
    return (size_t)MAXLR * C * 4 + MAXWW * 8 + (((size_t)Vq * MAXL * 3 * 4 + 15) & ~(size_t)15) + (size_t)4 * 3 * 32 * 128;
}

template <int C, bool SPLIT>
void launch(const FineArgs& g, hipStream_t stream) {
    static dfsfm::SmemAttr attr;
    attr.ensure(reinterpret_cast<const void*>(&fine_match_kernel<C, SPLIT>), 160 * 1024);
    hipLaunchKernelGGL((fine_match_kernel<C, SPLIT>), dim3(g.T), dim3(256), fine_smem_bytes(C, g.Vq), stream, g);
}

int fine_match_any(FineArgs g, bool split, int C, hipStream_t stream, const char* what) {
    if (g.T == 0) return DFSFM_OK;
    if (!g.track_mask || (split ? (!g.ref_h || !g.ref_l || !g.qry_h || !g.qry_l) : (!g.ref || !g.qry))) return DFSFM_E_BADARG;
    if (g.T < 0 || g.Vq <= 0 || g.W <= 1 || g.left <= 1) return DFSFM_E_BADARG;
    if (g.query_refined && (!g.query_pts || !g.scale_q)) return DFSFM_E_BADARG;
    if (g.ref_refined && (!g.ref_pts || !g.scale_r)) return DFSFM_E_BADARG;
    if (g.left > g.W || g.left * g.left > MAXLR || g.W * g.W > MAXWW || (g.left & 1) == 0 || (g.W & 1) == 0) return DFSFM_E_UNSUPPORTED;
    if (C != 128 && C != 64) return DFSFM_E_UNSUPPORTED;
    if (fine_smem_bytes(C, g.Vq) > 160 * 1024) return DFSFM_E_UNSUPPORTED;   // LDS budget: Vq <= 40 at C = 128, <= 61 at C = 64
    const uintptr_t al = split ? (reinterpret_cast<uintptr_t>(g.ref_h) | reinterpret_cast<uintptr_t>(g.ref_l) |
                                  reinterpret_cast<uintptr_t>(g.qry_h) | reinterpret_cast<uintptr_t>(g.qry_l))
                               : (reinterpret_cast<uintptr_t>(g.ref) | reinterpret_cast<uintptr_t>(g.qry));
    if (al & 15) return DFSFM_E_UNSUPPORTED;
    if (C == 128) { if (split) launch<128, true>(g, stream); else launch<128, false>(g, stream); }
    else { if (split) launch<64, true>(g, stream); else launch<64, false>(g, stream); }
    return dfsfm::check_launch(what);
}

}  // namespace

extern "C" int dfsfm_fine_match_f32(const float* ref, const float* qry, const uint8_t* track_mask,
                                    const uint8_t* movable, int T, int Vq, int W, int left, int C,
                                    const float* query_pts, const float* scale_q, const float* ref_pts,
                                    const float* scale_r, int64_t rs_t, int64_t rs_n, int32_t* best_index,
                                    float* left_norm, float* coords, float* std, float* query_refined,
                                    float* ref_refined, void* stream_) {
    FineArgs g{ref, qry, nullptr, nullptr, nullptr, nullptr, track_mask, movable, T, Vq, W, left, query_pts, scale_q, ref_pts,
               scale_r, rs_t, rs_n, best_index, left_norm, coords, std, query_refined, ref_refined};
    return fine_match_any(g, false, C, static_cast<hipStream_t>(stream_), "dfsfm_fine_match_f32");
}

extern "C" int dfsfm_fine_match_split(const void* ref_hi, const void* ref_lo, const void* qry_hi, const void* qry_lo,
                                      const uint8_t* track_mask, const uint8_t* movable, int T, int Vq, int W, int left, int C,
                                      const float* query_pts, const float* scale_q, const float* ref_pts,
                                      const float* scale_r, int64_t rs_t, int64_t rs_n, int32_t* best_index,
                                      float* left_norm, float* coords, float* std, float* query_refined,
                                      float* ref_refined, void* stream_) {
    FineArgs g{nullptr, nullptr, static_cast<const _Float16*>(ref_hi), static_cast<const _Float16*>(ref_lo),
               static_cast<const _Float16*>(qry_hi), static_cast<const _Float16*>(qry_lo), track_mask, movable, T, Vq, W, left,
               query_pts, scale_q, ref_pts, scale_r, rs_t, rs_n, best_index, left_norm, coords, std, query_refined, ref_refined};
    return fine_match_any(g, true, C, static_cast<hipStream_t>(stream_), "dfsfm_fine_match_split");
}
