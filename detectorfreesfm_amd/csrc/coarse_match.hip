// K3+K4+K5 -- all-pairs coarse correlation, dual-softmax, mutual-nearest-neighbour selection and
// keypoint epilogue for gfx950 (MI355X).  The L x S similarity / confidence matrices are never
// written to HBM.
//
// Replaces CoarseMatching.forward + get_coarse_match (eval, dual_softmax) of the reference
//   third_party/LoFTR/src/loftr/utils/coarse_matching.py:84-145, 148-258 (mask_border :8-22)
//
// Launch sequence (all on the caller's stream, batch index in blockIdx.z / blockIdx.y):
//   1. cm_gemm<STATS>  : 128x128 tiles of sim = (f0/sqrt(C)).(f1/sqrt(C))^T / temperature on the
//                        fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32, 157 TF peak);
//                        epilogue leaves per-tile (max, sum exp) partials for rows and columns.
//   2. cm_reduce_stats : merges the partials into the softmax statistics of every row/column
//                        and clears the best-candidate words.
//   3. cm_gemm<SELECT> : recomputes each sim tile (cheaper than a 92 MB round trip per matrix
//                        at fp32 MFMA rate would be to keep resident at batch 8), forms
//                        conf = softmax_col * softmax_row exactly as the reference does
//                        (exp(s-max)/sum per axis, then the product) and pushes, for entries
//                        with conf > thr only, the row-best (conf, smallest j) and column-best
//                        conf with order-independent integer atomicMax -> deterministic.
//   4. cm_select       : per row i: j* = row-best; keep iff conf>thr, conf == column-best[j*]
//                        (mutual), and (i, j*) is outside the low-side border; counts per pair.
//   5. cm_compact      : ordered compaction (ascending (b,i), as torch.where) + K5 epilogue.
//
// GEMM tile: 4 waves, each a 64x64 quadrant = 2x2 MFMA tiles; BK=32 staged through LDS with a
// 16-byte XOR swizzle (slot ^= (row>>1)&7) that makes every ds_read_b128 fragment read
// conflict-free; next k-slab is prefetched into registers while the MFMAs run.
#include "common.h"
#include "sf_gemm.h"
#include <algorithm>
#include <climits>
#include <cstdlib>

namespace {

using namespace dfsfm;

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE_LD = BN + 1;                               // padded epilogue tile
constexpr int SMEM_TILE = BM * TILE_LD * 4;                  // 66048 B
constexpr int SMEM_STATS = (BM + BN) * (8 + 4);              // row/col (max,sum) + candidate gates
constexpr int SMEM_BYTES = SMEM_TILE + SMEM_STATS;           // 68096 B -> 2 workgroups / CU

enum { MODE_STATS = 0, MODE_SELECT = 1, MODE_CONF = 2 };
constexpr int CAND_SLOTS_MAX = 8;                             // candidate slots per (row, column tile) the workspace is sized for

struct GemmArgs {
    const float* f0;
    const float* f1;
    int L, S, C;
    int ntm, ntn;            // tiles along L, S
    float op_div;            // operand divisor (sqrt(C)) when not folded, else 1
    float acc_mul;           // accumulator multiplier (1/C) when folded, else 1
    float temperature;
    float thr;
    float2* row_part;        // [N][ntn][L]  (max, sumexp)
    float2* col_part;        // [N][ntm][S]
    const float2* row_stat;  // [N][L]
    const float2* col_stat;  // [N][S]
    unsigned long long* row_best;   // [N][L]  conf bits << 32 | ~j
    unsigned int* col_best;         // [N][S]  conf bits
    float* conf_out;         // [N][L][S] (MODE_CONF)
};

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

// Bijective XCD-aware remap (MI355X guide T1): consecutive logical tiles -> same XCD L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <int MODE, bool PRESCALE>
__global__ __launch_bounds__(256, 2) void cm_gemm(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sA = reinterpret_cast<float*>(smem);                       // [2][BM][BK]
    float* sB = sA + 2 * BM * BK;                                     // [2][BN][BK]
    float* tile = reinterpret_cast<float*>(smem);                     // [BM][TILE_LD] (aliases)
    float2* s_rstat = reinterpret_cast<float2*>(smem + SMEM_TILE);    // [BM]
    float2* s_cstat = s_rstat + BM;                                   // [BN]
    float* s_rgate = reinterpret_cast<float*>(s_cstat + BN);          // [BM] SELECT: sim needed for p_row > thr
    float* s_cgate = s_rgate + BM;                                    // [BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int n = blockIdx.y;
    const int t_id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = t_id / g.ntn, tn = t_id % g.ntn;
    const int row0 = tm * BM, col0 = tn * BN;
    const float* A = g.f0 + (int64_t)n * g.L * g.C;
    const float* B = g.f1 + (int64_t)n * g.S * g.C;

    if (MODE != MODE_STATS) {
        // conf = p_row * p_col > thr needs p_row > thr and p_col > thr, i.e.
        // sim > max + log(thr * sum) on both axes: a cheap gate (1e-3 slack covers exp/log rounding)
        // that keeps the two expf + two divisions off all but the rare candidate entries.
        if (tid < BM) {
            const int i = row0 + tid;
            const float2 st = i < g.L ? g.row_stat[(int64_t)n * g.L + i] : make_float2(0.f, 1.f);
            s_rstat[tid] = st;
            s_rgate[tid] = i < g.L ? st.x + logf(g.thr * st.y) - 1e-3f : INFINITY;
        } else {
            const int j = col0 + tid - BM;
            const float2 st = j < g.S ? g.col_stat[(int64_t)n * g.S + j] : make_float2(0.f, 1.f);
            s_cstat[tid - BM] = st;
            s_cgate[tid - BM] = j < g.S ? st.x + logf(g.thr * st.y) - 1e-3f : INFINITY;
        }
    }

    // ---- staging: each thread moves 4 float4 of A and 4 of B per k-slab --------------------
    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + 256 * j, r = idx >> 3, c = idx & 7;
            // rows past the end are clamped (branch-free); the epilogue never looks at them
            ra[j] = *reinterpret_cast<const f32x4*>(A + (int64_t)min(row0 + r, g.L - 1) * g.C + k0 + c * 4);
            rb[j] = *reinterpret_cast<const f32x4*>(B + (int64_t)min(col0 + r, g.S - 1) * g.C + k0 + c * 4);
            if (PRESCALE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { ra[j][e] = ra[j][e] / g.op_div; rb[j][e] = rb[j][e] / g.op_div; }
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + 256 * j, r = idx >> 3, c = idx & 7;
            *reinterpret_cast<f32x4*>(sA + buf * BM * BK + r * BK + swz(r, c) * 4) = ra[j];
            *reinterpret_cast<f32x4*>(sB + buf * BN * BK + r * BK + swz(r, c) * 4) = rb[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};

    const int nk = g.C / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const float* a_base = sA + buf * BM * BK;
        const float* b_base = sB + buf * BN * BK;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {        // lane half h owns k in [16h, 16h+16) of the slab
            f32x4 av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int r = wr * 64 + a * 32 + col;
                av[a] = *reinterpret_cast<const f32x4*>(a_base + r * BK + swz(r, half * 4 + qd) * 4);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int r = wc * 64 + b * 32 + col;
                bv[b] = *reinterpret_cast<const f32x4*>(b_base + r * BK + swz(r, half * 4 + qd) * 4);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][t], bv[b][t], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------
    // lane holds sim[row = wr*64 + a*32 + mfma32_row(r,half)][col = wc*64 + b*32 + (lane&31)]
    bool any_above = false;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wr * 64 + a * 32 + mfma32_row(r, half);
                const int lc = wc * 64 + b * 32 + col;
                float s = (acc[a][b][r] * g.acc_mul) / g.temperature;
                if (MODE == MODE_CONF || (MODE == MODE_SELECT && s > s_rgate[lr] && s > s_cgate[lc])) {
                    const float2 rs = s_rstat[lr], cs = s_cstat[lc];
                    // softmax over dim 1 (column j normalised over i) * softmax over dim 2
                    const float p_col = expf(s - cs.x) / cs.y;
                    const float p_row = expf(s - rs.x) / rs.y;
                    s = p_col * p_row;
                    if (MODE == MODE_SELECT) any_above |= s > g.thr;
                } else if (MODE == MODE_SELECT) {
                    s = 0.f;                    // cannot exceed thr; value is never used
                }
                acc[a][b][r] = s;
            }

    if (MODE == MODE_SELECT) {
        // The overwhelmingly common tile has no entry above thr: nothing to record.
        if (!__syncthreads_or(any_above)) return;
    } else {
        __syncthreads();   // staging buffers are dead, the tile aliases them
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wr * 64 + a * 32 + mfma32_row(r, half);
                const int lc = wc * 64 + b * 32 + col;
                tile[lr * TILE_LD + lc] = acc[a][b][r];
            }
    __syncthreads();

    const int nrow = min(BM, g.L - row0), ncol = min(BN, g.S - col0);
    if (MODE == MODE_STATS) {
        if (tid < BM) {
            if (tid < nrow) {
                float m = -INFINITY;
                for (int j = 0; j < ncol; ++j) m = fmaxf(m, tile[tid * TILE_LD + j]);
                float sum = 0.f;
                for (int j = 0; j < ncol; ++j) sum += expf(tile[tid * TILE_LD + j] - m);
                g.row_part[((int64_t)n * g.ntn + tn) * g.L + row0 + tid] = make_float2(m, sum);
            }
        } else {
            const int c = tid - BM;
            if (c < ncol) {
                float m = -INFINITY;
                for (int i = 0; i < nrow; ++i) m = fmaxf(m, tile[i * TILE_LD + c]);
                float sum = 0.f;
                for (int i = 0; i < nrow; ++i) sum += expf(tile[i * TILE_LD + c] - m);
                g.col_part[((int64_t)n * g.ntm + tm) * g.S + col0 + c] = make_float2(m, sum);
            }
        }
    } else if (MODE == MODE_SELECT) {
        if (tid < BM) {
            if (tid < nrow) {
                float best = g.thr;
                int bj = -1;
                for (int j = 0; j < ncol; ++j) {
                    const float v = tile[tid * TILE_LD + j];
                    if (v > best) { best = v; bj = j; }      // strict > keeps the smallest j on ties
                }
                if (bj >= 0) {
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(~(unsigned)(col0 + bj));
                    atomicMax(&g.row_best[(int64_t)n * g.L + row0 + tid], key);
                }
            }
        } else {
            const int c = tid - BM;
            if (c < ncol) {
                float best = g.thr;
                bool found = false;
                for (int i = 0; i < nrow; ++i) {
                    const float v = tile[i * TILE_LD + c];
                    if (v > best) { best = v; found = true; }
                }
                if (found) atomicMax(&g.col_best[(int64_t)n * g.S + col0 + c], __float_as_uint(best));
            }
        }
    } else {   // MODE_CONF: dump the confidence tile, coalesced
        for (int e = tid; e < BM * BN; e += 256) {
            const int r = e / BN, c = e % BN;
            if (r < nrow && c < ncol)
                g.conf_out[((int64_t)n * g.L + row0 + r) * g.S + col0 + c] = tile[r * TILE_LD + c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Split-plane variant of the correlation GEMM: features arrive as fp16 hi / lo planes (x = hi + lo/2048,
// written by the last encoder layer's LayerNorm) and sim = f0.f1^T runs as three fp16 MFMA products per
// k-step on the 512-thread LDS-DMA ping-pong main loop of sf_gemm.h (the 1x1 / linear schedule: f0 rows are
// the "activations", f1 rows the "weights").  256 x 128 tiles; epilogues as in cm_gemm.  Column partials are
// written per 128-row half tile so all 512 threads share the exp work.
// ---------------------------------------------------------------------------------------------
constexpr int SF_BM = 256, SF_BN = 128, SF_LD = SF_BN + 1;
using SfRing = dfsfm_sf::VS<SF_BN, 1>;
constexpr int SF_STATS_OFF = SfRing::RING;                            // stats live past the DMA ring
constexpr int SF_SMEM = SF_STATS_OFF + (SF_BM + SF_BN) * (8 + 4);
static_assert(SF_BM * SF_LD * 4 <= SF_STATS_OFF, "epilogue tile must not reach the stats");
static_assert(SF_SMEM <= 160 * 1024, "LDS");

struct GemmSfArgs {
    const _Float16 *f0h, *f0l, *f1h, *f1l;     // [N][L][C], [N][S][C]
    int L, S, C;
    int ntn;                 // tiles along S
    unsigned ntiles;         // tiles per pair (grid.x is padded to a multiple of 8)
    float acc_mul, temperature, thr;
    float2* row_part;        // [N][ntn][L]
    float2* col_part;        // [N][2 * ntm][S]: one partial per 128-row half tile
    int nhalf;               // 2 * ntm
    const float2* row_stat;
    const float2* col_stat;
    unsigned long long* row_best;
    unsigned int* col_best;
    const uint8_t* mask0;    // optional padding masks [N][L], [N][S] (1 = valid): masked rows / columns take no part
    const uint8_t* mask1;    //     (the reference fills their similarities with -1e9, coarse_matching.py:110-113)
    float2* cand;            // cm_gemm_sf2_cand: [N][ntn][L][slots] (sim, column index bits) of entries that pass the tile-local gates
    uint8_t* cand_cnt;       //            [N][ntn][L] number of valid slots
    int slots;
    int sb;                  // cm_gemm_sf2_cand: tile rows per traversal group (L2-aware tile order)
};

// exp for the softmax denominators of the split path: v_exp_f32(x * log2 e).  The argument is <= 0; the product's
// rounding error is relative to |x|, so the terms that dominate a sum (x near 0) are accurate to ~1 ulp and the
// terms it perturbs by up to 1e-6 relative (x ~ -20) carry weight e^-20.  Three instructions instead of ~15.
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

// All-reduce over the 32 lanes of a half wave (lanes that hold the 32 columns of one accumulator row): four DPP
// steps inside the 16-lane row (quad swaps, half mirror, mirror) + one cross-row exchange.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_wave_max(float x) {
    x = fmaxf(x, dpp_mov<0xB1>(x));      // quad_perm [1,0,3,2]
    x = fmaxf(x, dpp_mov<0x4E>(x));      // quad_perm [2,3,0,1]
    x = fmaxf(x, dpp_mov<0x141>(x));     // row_half_mirror
    x = fmaxf(x, dpp_mov<0x140>(x));     // row_mirror
    return fmaxf(x, __shfl_xor(x, 16));
}
__device__ __forceinline__ float half_wave_sum(float x) {
    x += dpp_mov<0xB1>(x);
    x += dpp_mov<0x4E>(x);
    x += dpp_mov<0x141>(x);
    x += dpp_mov<0x140>(x);
    return x + __shfl_xor(x, 16);
}
// (max, sum exp) pairs of two disjoint index sets -> pair of the union
__device__ __forceinline__ float2 merge_stat(float2 a, float2 b) {
    const float m = fmaxf(a.x, b.x);
    if (m == -INFINITY) return make_float2(m, 0.f);
    return make_float2(m, a.y * fast_exp(a.x - m) + b.y * fast_exp(b.x - m));
}

// xor-16 / xor-32 lane exchanges without the LDS crossbar (gfx950: v_permlane16_swap / v_permlane32_swap): both lanes of a
// pair end up with the two values {own, partner} in (a, b) order (row 0 | row 1, lower | upper half), so a commutative
// combination of a and b is the same on both.
__device__ __forceinline__ void swap16(float x, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap32(float x, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// Transposing butterfly over the 16 lanes of a DPP row: every lane brings 16 values x[r]; afterwards lane i holds
// OP over the row's 16 lanes of value r = i & 15.  15 exchanges instead of the 64 of sixteen separate all-reduces: at every
// step a lane keeps the half of its values whose index bit equals its own lane bit and sends the other half to the partner
// that keeps those (row_mirror flips lane bit 3, row_half_mirror bit 2, the quad permutations bits 1 and 0; the lower bits
// they flip as well do not matter before their own step).
template <class Op>
__device__ __forceinline__ float butterfly16(const float (&x)[16], int lane, Op op) {
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    float a[8], b[4], c[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = op(b3 ? x[k + 8] : x[k], dpp_mov<0x140>(b3 ? x[k] : x[k + 8]));
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = op(b2 ? a[k + 4] : a[k], dpp_mov<0x141>(b2 ? a[k] : a[k + 4]));
#pragma unroll
    for (int k = 0; k < 2; ++k) c[k] = op(b1 ? b[k + 2] : b[k], dpp_mov<0x4E>(b1 ? b[k] : b[k + 2]));
    return op(b0 ? c[1] : c[0], dpp_mov<0xB1>(b0 ? c[0] : c[1]));
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void cm_gemm_sf(GemmSfArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);                             // [SF_BM][SF_LD] (aliases the ring)
    float2* s_rstat = reinterpret_cast<float2*>(smem + SF_STATS_OFF);         // [SF_BM]
    float2* s_cstat = s_rstat + SF_BM;                                        // [SF_BN]
    float* s_rgate = reinterpret_cast<float*>(s_cstat + SF_BN);               // [SF_BM]
    float* s_cgate = s_rgate + SF_BM;                                         // [SF_BN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int n = blockIdx.y;
    const unsigned t_id = dfsfm_sf::xcd_band_tile(blockIdx.x, gridDim.x >> 3);
    if (t_id >= g.ntiles) return;
    const int tm = t_id / g.ntn, tn = t_id % g.ntn;
    const int row0 = tm * SF_BM, col0 = tn * SF_BN;

    if (MODE == MODE_SELECT) {
        if (tid < SF_BM) {
            const int i = row0 + tid;
            const float2 st = i < g.L ? g.row_stat[(int64_t)n * g.L + i] : make_float2(0.f, 1.f);
            s_rstat[tid] = st;
            s_rgate[tid] = (i < g.L && st.x != -INFINITY) ? st.x + logf(g.thr * st.y) - 1e-3f : INFINITY;
        } else if (tid < SF_BM + SF_BN) {
            const int j = col0 + tid - SF_BM;
            const float2 st = j < g.S ? g.col_stat[(int64_t)n * g.S + j] : make_float2(0.f, 1.f);
            s_cstat[tid - SF_BM] = st;
            s_cgate[tid - SF_BM] = (j < g.S && st.x != -INFINITY) ? st.x + logf(g.thr * st.y) - 1e-3f : INFINITY;
        }
    }

    dfsfm_sf::ConvArgs a{};
    a.xh = g.f0h + (int64_t)n * g.L * g.C;
    a.xl = g.f0l + (int64_t)n * g.L * g.C;
    a.wh = g.f1h + (int64_t)n * g.S * g.C;
    a.wl = g.f1l + (int64_t)n * g.S * g.C;
    a.M = g.L; a.H = 1; a.W = g.L; a.Cin = g.C; a.Kpad = g.C; a.ldx = g.C;
    a.sxh = a.sxn = (int64_t)g.L * g.C;
    a.xbytes = (unsigned)((int64_t)g.L * g.C * 2);           // rows past L / S are zero-filled by the DMA
    a.wbytes = (unsigned)((int64_t)g.S * g.C * 2);
    f32x16 accm[2][2], accx[2][2];
    dfsfm_sf::sf_same_mainloop<SF_BN, 1>(a, smem, accm, accx, row0, col0);
    __syncthreads();                                          // ring is dead: the tile aliases it

    // lane holds sim[row = wr*64 + i*32 + mfma32_row(r,half)][col = wc*64 + j*32 + (lane&31)]
    const int nrow = min(SF_BM, g.L - row0), ncol = min(SF_BN, g.S - col0);
    if (MODE == MODE_STATS) {
        // Statistics straight from the accumulator registers: a lane's 16 registers of one block are 16 rows of ONE
        // column (column sums are lane-local), and the 32 lanes of a half wave hold the 32 columns of one row (row
        // sums are a 32-lane butterfly).  Only the per-wave partials go through LDS.
        float2* s_rp = reinterpret_cast<float2*>(smem);                   // [2 wc][SF_BM]
        float2* s_cp = s_rp + 2 * SF_BM;                                  // [4 wr][SF_BN]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = wr * 64 + i * 32 + mfma32_row(r, half);
                    const int lc = wc * 64 + j * 32 + col;
                    const float sv = ((accm[i][j][r] + accx[i][j][r] * (1.f / 2048.f)) * g.acc_mul) / g.temperature;
                    bool ok = lr < nrow && lc < ncol;                              // rows past L / columns past S
                    if (g.mask0) ok = ok && g.mask0[(int64_t)n * g.L + min(row0 + lr, g.L - 1)] && g.mask1[(int64_t)n * g.S + min(col0 + lc, g.S - 1)];
                    accm[i][j][r] = ok ? sv : -INFINITY;
                }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                                     // columns: 32 lane-local rows, then the other half
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, accm[i][j][r]);
            float e = 0.f;
            if (m != -INFINITY) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) e += fast_exp(accm[i][j][r] - m);
            }
            const float2 st = merge_stat(make_float2(m, e), make_float2(__shfl_xor(m, 32), __shfl_xor(e, 32)));
            if (half == 0) s_cp[wr * SF_BN + wc * 64 + j * 32 + col] = st;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {                                // rows: 2 lane-local columns, then 32 lanes
                const float v0 = accm[i][0][r], v1 = accm[i][1][r];
                const float m = half_wave_max(fmaxf(v0, v1));
                float e = (m != -INFINITY) ? fast_exp(v0 - m) + fast_exp(v1 - m) : 0.f;
                e = half_wave_sum(e);
                if (col == 0) s_rp[wc * SF_BM + wr * 64 + i * 32 + mfma32_row(r, half)] = make_float2(m, e);
            }
        __syncthreads();
        if (tid < SF_BM) {
            const float2 st = merge_stat(s_rp[tid], s_rp[SF_BM + tid]);
            if (tid < nrow) g.row_part[((int64_t)n * g.ntn + tn) * g.L + row0 + tid] = st;
        } else {
            const int c = (tid - SF_BM) & (SF_BN - 1), h = (tid - SF_BM) >> 7;   // column c, 128-row half h
            const float2 st = merge_stat(s_cp[(2 * h) * SF_BN + c], s_cp[(2 * h + 1) * SF_BN + c]);
            if (c < ncol)        // an empty half (rows past L) leaves the neutral partial (-inf, 0)
                g.col_part[((int64_t)n * g.nhalf + tm * 2 + h) * g.S + col0 + c] = st;
        }
        return;
    }
    bool any_above = false;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wr * 64 + i * 32 + mfma32_row(r, half);
                const int lc = wc * 64 + j * 32 + col;
                float s = ((accm[i][j][r] + accx[i][j][r] * (1.f / 2048.f)) * g.acc_mul) / g.temperature;
                if (s > s_rgate[lr] && s > s_cgate[lc]) {
                    const float2 rs = s_rstat[lr], cs = s_cstat[lc];
                    const float p_col = expf(s - cs.x) / cs.y;
                    const float p_row = expf(s - rs.x) / rs.y;
                    s = p_col * p_row;
                    any_above |= s > g.thr;
                } else {
                    s = 0.f;
                }
                accm[i][j][r] = s;
            }
    if (MODE == MODE_SELECT) {
        if (!__syncthreads_or(any_above)) return;             // the common tile: nothing above thr
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[(wr * 64 + i * 32 + mfma32_row(r, half)) * SF_LD + wc * 64 + j * 32 + col] = accm[i][j][r];
    __syncthreads();

    {
        if (tid < SF_BM) {
            if (tid < nrow) {
                float best = g.thr;
                int bj = -1;
                for (int j = 0; j < ncol; ++j) {
                    const float v = tile[tid * SF_LD + j];
                    if (v > best) { best = v; bj = j; }      // strict > keeps the smallest j on ties
                }
                if (bj >= 0) {
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(~(unsigned)(col0 + bj));
                    atomicMax(&g.row_best[(int64_t)n * g.L + row0 + tid], key);
                }
            }
        } else if (tid < SF_BM + SF_BN) {
            const int c = tid - SF_BM;
            if (c < ncol) {
                float best = g.thr;
                bool found = false;
                for (int i = 0; i < nrow; ++i) {
                    const float v = tile[i * SF_LD + c];
                    if (v > best) { best = v; found = true; }
                }
                if (found) atomicMax(&g.col_best[(int64_t)n * g.S + col0 + c], __float_as_uint(best));
            }
        }
    }
}


// Single-GEMM candidate pass on the 128 x 128 / two-workgroups-per-CU schedule (dfsfm_sf::sf2_mainloop): the statistics /
// candidate epilogue of one workgroup runs under the correlation main loop of the co-resident one (with K = 256 the 512-thread
// kernel above spends ~10 us in its main loop and ~7 us in this epilogue with nothing else on the CU), and the tiles are half
// as tall.  Same arithmetic, same statistics, same candidates: rows and confidences are identical to the two-pass path (cm_gemm_sf<STATS> + <SELECT>).
constexpr int SF2_BM = 128;
__global__ __launch_bounds__(256, 2) void cm_gemm_sf2_cand(GemmSfArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int n = blockIdx.y;
    const unsigned t_id = dfsfm_sf::xcd_band_tile(blockIdx.x, gridDim.x >> 3);
    if (t_id >= g.ntiles) return;
    // L2-aware traversal (r05).  XCD x runs the contiguous range [x T/8, (x+1) T/8) of a tile ORDER, 64 tiles at a time (32 CUs x 2
    // workgroups).  In row-major order those 64 tiles are 1.7 tile rows x all 38 column tiles: every tile row re-streams the whole
    // f1 panel (4.9 MB per pair at 4800 rows, more than an XCD's 4-MB L2) -- 1.22 GB fetched per 8 pairs for 79 MB of features
    // (profiles/r04_pmc_traffic.json).  The order used here walks row GROUPS of g.sb tile rows column tile by column tile
    // (tn outer, tm inner): the group's f0 rows (sb x 128 KB) stay in L2 while f1 streams past once per group.
    const unsigned gsz = (unsigned)(g.sb * g.ntn), grp = t_id / gsz, rem = t_id - grp * gsz;
    const int rows_g = min(g.sb, g.nhalf - (int)grp * g.sb);              // g.nhalf = tile rows (one column partial per tile row)
    const int tn = (int)rem / rows_g, tm = (int)grp * g.sb + (int)rem % rows_g;
    const int row0 = tm * SF2_BM, col0 = tn * SF_BN;

    dfsfm_sf::ConvArgs a{};
    a.xh = g.f0h + (int64_t)n * g.L * g.C;
    a.xl = g.f0l + (int64_t)n * g.L * g.C;
    a.wh = g.f1h + (int64_t)n * g.S * g.C;
    a.wl = g.f1l + (int64_t)n * g.S * g.C;
    a.M = g.L; a.H = 1; a.W = g.L; a.Cin = g.C; a.Kpad = g.C; a.ldx = g.C;
    a.sxh = a.sxn = (int64_t)g.L * g.C;
    a.xbytes = (unsigned)((int64_t)g.L * g.C * 2);           // rows past L / S are zero-filled by the DMA
    a.wbytes = (unsigned)((int64_t)g.S * g.C * 2);
    f32x16 accm[2][2], accx[2][2];
    dfsfm_sf::sf2_mainloop<1>(a, smem, accm, accx, row0, col0);
    __syncthreads();                                          // ring is dead

    const int nrow = min(SF2_BM, g.L - row0), ncol = min(SF_BN, g.S - col0);
    float2* s_rp = reinterpret_cast<float2*>(smem);                       // [2 wc][128]
    float2* s_cp = s_rp + 2 * SF2_BM;                                     // [2 wr][128]
    float* s_rg = reinterpret_cast<float*>(smem + 8192);                  // [128] tile-local row gate
    float* s_cg = s_rg + SF2_BM;                                          // [128] tile-local column gate
    int* s_cnt = reinterpret_cast<int*>(s_cg + SF_BN);                    // [128] slots taken per row
    // validity as additive penalties (0 / -inf) per accumulator row / column instead of two byte loads and a select per element
    {
        float rpen[2][16], cpen[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wr * 64 + i * 32 + mfma32_row(r, half);
                bool ok = lr < nrow;
                if (g.mask0) ok = ok && g.mask0[(int64_t)n * g.L + min(row0 + lr, g.L - 1)];
                rpen[i][r] = ok ? 0.f : -INFINITY;
            }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int lc = wc * 64 + j * 32 + col;
            bool ok = lc < ncol;
            if (g.mask1) ok = ok && g.mask1[(int64_t)n * g.S + min(col0 + lc, g.S - 1)];
            cpen[j] = ok ? 0.f : -INFINITY;
        }
        // similarity = (acc * acc_mul) / temperature as ONE multiplication by acc_mul / temperature (r04: the IEEE division was
        // ten instructions per element, 640 of this epilogue's ~1500; candidates, statistics and cm_eval all see this value,
        // which differs from the two-pass kernels' by at most one rounding)
        const float scale = g.acc_mul / g.temperature;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    accm[i][j][r] = (((accm[i][j][r] + accx[i][j][r] * (1.f / 2048.f)) * scale) + rpen[i][r]) + cpen[j];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {                                         // columns: 32 lane-local rows, then the other half
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, accm[i][j][r]);
        float e = 0.f;
        if (m != -INFINITY) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) e += fast_exp(accm[i][j][r] - m);
        }
        float m_lo, m_hi, e_lo, e_hi;
        swap32(m, m_lo, m_hi);
        swap32(e, e_lo, e_hi);
        const float2 st = merge_stat(make_float2(m_lo, e_lo), make_float2(m_hi, e_hi));
        if (half == 0) s_cp[wr * SF_BN + wc * 64 + j * 32 + col] = st;
    }
    // rows: 2 lane-local columns, then the 32 lanes of the half wave -- as ONE transposing butterfly per 32-row block (15 DPP
    // exchanges + one xor-16 swap leave row r's total in lanes r, r + 16 of each half; r04: sixteen separate all-reduces per block
    // were 2 x 16 x (4 DPP steps + an LDS-crossbar exchange)); the maxima return to all lanes through 256 B of LDS per wave
    float* s_m = reinterpret_cast<float*>(smem + 12288) + (wr * 2 + wc) * 64;       // [4 waves][64 rows]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr_own = i * 32 + mfma32_row(lane & 15, half);         // row of this wave's 64 whose totals this lane receives
        float pm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pm[r] = fmaxf(accm[i][0][r], accm[i][1][r]);
        float ma, mb;
        swap16(butterfly16(pm, lane, [](float x, float y) { return fmaxf(x, y); }), ma, mb);
        const float m_own = fmaxf(ma, mb);
        if ((lane & 16) == 0) s_m[lr_own] = m_own;
        float pe[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                     // rows mfma32_row(4 q + k, half) = 8 q + 4 half + k
            const f32x4 mr = *reinterpret_cast<const f32x4*>(s_m + i * 32 + 8 * q + 4 * half);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                pe[4 * q + k] = (mr[k] != -INFINITY) ? fast_exp(accm[i][0][4 * q + k] - mr[k]) + fast_exp(accm[i][1][4 * q + k] - mr[k]) : 0.f;
        }
        float ea, eb;
        swap16(butterfly16(pe, lane, [](float x, float y) { return x + y; }), ea, eb);
        if ((lane & 16) == 0) s_rp[wc * SF2_BM + wr * 64 + lr_own] = make_float2(m_own, ea + eb);
    }
    __syncthreads();
    if (tid < SF2_BM) {
        const float2 st = merge_stat(s_rp[tid], s_rp[SF2_BM + tid]);
        if (tid < nrow) g.row_part[((int64_t)n * g.ntn + tn) * g.L + row0 + tid] = st;
        s_rg[tid] = (tid < nrow && st.x != -INFINITY) ? st.x + logf(g.thr * st.y) - 1e-3f : INFINITY;
        s_cnt[tid] = 0;
    } else {
        const int c = tid - SF2_BM;
        const float2 st = merge_stat(s_cp[c], s_cp[SF_BN + c]);
        if (c < ncol) g.col_part[((int64_t)n * g.nhalf + tm) * g.S + col0 + c] = st;    // one partial per 128-row tile
        s_cg[c] = (c < ncol && st.x != -INFINITY) ? st.x + logf(g.thr * st.y) - 1e-3f : INFINITY;
    }
    __syncthreads();
    float cg[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) cg[j] = s_cg[wc * 64 + j * 32 + col];
    const int64_t slot0 = ((int64_t)n * g.ntn + tn) * g.L + row0;
    // candidates are rare (at most floor(1 / thr) per row and tile, none in most tiles): the gates of this lane's 32 rows come
    // back as eight 16-byte LDS reads, ONE wave-uniform test per tile decides whether the divergent per-element path runs at all
    float rgv[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                     // rows mfma32_row(4 q + k, half) = 8 q + 4 half + k
            const f32x4 v = *reinterpret_cast<const f32x4*>(s_rg + wr * 64 + i * 32 + 8 * q + 4 * half);
            rgv[i][4 * q] = v[0]; rgv[i][4 * q + 1] = v[1]; rgv[i][4 * q + 2] = v[2]; rgv[i][4 * q + 3] = v[3];
        }
    bool any = false;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            any = any || fmaxf(accm[i][0][r] - fmaxf(rgv[i][r], cg[0]), accm[i][1][r] - fmaxf(rgv[i][r], cg[1])) > 0.f;
    if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wr * 64 + i * 32 + mfma32_row(r, half);
                const float rg = rgv[i][r];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float sv = accm[i][j][r];                       // -inf outside the matrix
                    if (sv > rg && sv > cg[j]) {
                        const int slot = atomicAdd(&s_cnt[lr], 1);
                        if (slot < g.slots)
                            g.cand[(slot0 + lr) * g.slots + slot] = make_float2(sv, __int_as_float(col0 + wc * 64 + j * 32 + col));
                    }
                }
            }
    }
    __syncthreads();
    if (tid < nrow) g.cand_cnt[slot0 + tid] = (uint8_t)min(s_cnt[tid], g.slots);
}

// Merge per-tile (max, sumexp) partials; clear row_best / col_best.
__global__ __launch_bounds__(256) void cm_reduce_stats(const float2* __restrict__ row_part,
                                                       const float2* __restrict__ col_part,
                                                       float2* __restrict__ row_stat,
                                                       float2* __restrict__ col_stat,
                                                       unsigned long long* __restrict__ row_best,
                                                       unsigned int* __restrict__ col_best, int L, int S,
                                                       int ntm, int ntn) {
    const int n = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < L) {
        float m = -INFINITY;
        for (int t = 0; t < ntn; ++t) m = fmaxf(m, row_part[((int64_t)n * ntn + t) * L + e].x);
        float sum = 0.f;
        for (int t = 0; t < ntn; ++t) {
            const float2 p = row_part[((int64_t)n * ntn + t) * L + e];
            sum += p.y * expf(p.x - m);
        }
        row_stat[(int64_t)n * L + e] = make_float2(m, sum);
        row_best[(int64_t)n * L + e] = 0ull;
    } else if (e < L + S) {
        const int j = e - L;
        float m = -INFINITY;
        for (int t = 0; t < ntm; ++t) m = fmaxf(m, col_part[((int64_t)n * ntm + t) * S + j].x);
        float sum = 0.f;
        for (int t = 0; t < ntm; ++t) {
            const float2 p = col_part[((int64_t)n * ntm + t) * S + j];
            sum += p.y * expf(p.x - m);
        }
        col_stat[(int64_t)n * S + j] = make_float2(m, sum);
        col_best[(int64_t)n * S + j] = 0u;
    }
}

// Single-GEMM path: exact confidences of the stored candidates (reference operation order: exp(s - max) / sum per
// axis, then the product -- coarse_matching.py:106-116) with the GLOBAL statistics; row-best (conf, smallest j) needs no
// atomics (one thread sees every candidate of its row), column-best via the same order-independent atomicMax as the
// two-pass path.  Decisions and values are identical to cm_gemm_sf<SELECT>: same fp32 similarity, same arithmetic.
__global__ __launch_bounds__(256) void cm_eval(const float2* __restrict__ cand, const uint8_t* __restrict__ cand_cnt,
                                               const float2* __restrict__ row_stat, const float2* __restrict__ col_stat,
                                               unsigned long long* __restrict__ row_best, unsigned int* __restrict__ col_best,
                                               int L, int S, int ntn, int slots, float thr) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const float2 rs = row_stat[(int64_t)n * L + i];
    unsigned long long key = 0ull;
    // the slot counts of eight column tiles at a time: independent byte loads in flight together (almost all of them are zero;
    // one dependent load per tile made this kernel 58 us of pure latency at 38 tiles)
    for (int t0 = 0; t0 < ntn; t0 += 8) {
      int cnt[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cnt[u] = t0 + u < ntn ? (int)cand_cnt[((int64_t)n * ntn + t0 + u) * L + i] : 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t e = ((int64_t)n * ntn + t0 + u) * L + i;
        const int c = cnt[u];
        for (int q = 0; q < c; ++q) {
            const float2 cd = cand[e * slots + q];
            const float s = cd.x;
            const int j = __float_as_int(cd.y);
            const float2 cs = col_stat[(int64_t)n * S + j];
            const float p_col = expf(s - cs.x) / cs.y;
            const float p_row = expf(s - rs.x) / rs.y;
            const float conf = p_col * p_row;
            if (conf > thr) {
                atomicMax(&col_best[(int64_t)n * S + j], __float_as_uint(conf));
                const unsigned long long k = ((unsigned long long)__float_as_uint(conf) << 32) | (unsigned)(~(unsigned)j);
                key = k > key ? k : key;
            }
        }
      }
    }
    if (key != 0ull) row_best[(int64_t)n * L + i] = key;
}

// Valid extents of the padded frames of pair n, as mask_border_with_padding computes them (coarse_matching.py:35-36):
// h = max over the columns of the column sums, w = max over the rows of the row sums.  One workgroup per (pair, frame);
// ext[n] = {h0, w0, h1, w1}.
__global__ __launch_bounds__(256) void cm_mask_extents(const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                                       int32_t* __restrict__ ext, int h0c, int w0c, int h1c, int w1c) {
    const int n = blockIdx.x, which = blockIdx.y;
    const int h = which ? h1c : h0c, w = which ? w1c : w0c;
    const uint8_t* m = (which ? mask1 : mask0) + (int64_t)n * h * w;
    __shared__ int s_h, s_w;
    if (threadIdx.x == 0) { s_h = 0; s_w = 0; }
    __syncthreads();
    int best_h = 0, best_w = 0;
    for (int x = threadIdx.x; x < w; x += blockDim.x) {
        int c = 0;
        for (int y = 0; y < h; ++y) c += m[y * w + x] ? 1 : 0;
        best_h = max(best_h, c);
    }
    for (int y = threadIdx.x; y < h; y += blockDim.x) {
        int c = 0;
        for (int x = 0; x < w; ++x) c += m[y * w + x] ? 1 : 0;
        best_w = max(best_w, c);
    }
    atomicMax(&s_h, best_h);
    atomicMax(&s_w, best_w);
    __syncthreads();
    if (threadIdx.x == 0) { ext[n * 4 + which * 2] = s_h; ext[n * 4 + which * 2 + 1] = s_w; }
}

// First index a Python slice ``m[start:]`` with start = extent - border removes on an axis of length len
// (a negative start counts from the end, clamped at 0).
__device__ __forceinline__ int py_slice_start(int extent, int border, int len) {
    int s = extent - border;
    if (s < 0) s = max(s + len, 0);
    return s;
}

// Per-row decision; one workgroup per pair.  flags[n][i] = 1 iff row i yields a match.
__global__ __launch_bounds__(1024) void cm_select(const unsigned long long* __restrict__ row_best,
                                                  const unsigned int* __restrict__ col_best,
                                                  uint8_t* __restrict__ flags, int32_t* __restrict__ counts,
                                                  int L, int S, float thr, int border, int w0c, int w1c,
                                                  const int32_t* __restrict__ ext) {
    const int n = blockIdx.x;
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    // mask_border: only the LOW side of each grid axis is removed (reference quirk, coarse_matching.py:8-22).  With
    // padding masks (mask_border_with_padding, :25-41; ext != nullptr) also [extent - border, end) of every axis.
    int hi_y0 = INT_MAX, hi_x0 = INT_MAX, hi_y1 = INT_MAX, hi_x1 = INT_MAX;
    if (ext && border > 0) {
        hi_y0 = py_slice_start(ext[n * 4 + 0], border, L / w0c);
        hi_x0 = py_slice_start(ext[n * 4 + 1], border, w0c);
        hi_y1 = py_slice_start(ext[n * 4 + 2], border, S / w1c);
        hi_x1 = py_slice_start(ext[n * 4 + 3], border, w1c);
    }
    int local = 0;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const unsigned long long key = row_best[(int64_t)n * L + i];
        bool ok = false;
        if (key != 0ull) {
            const unsigned bits = (unsigned)(key >> 32);
            const int j = (int)(~(unsigned)(key & 0xffffffffull));
            const float v = __uint_as_float(bits);
            ok = v > thr && col_best[(int64_t)n * S + j] == bits;
            ok = ok && (i / w0c >= border) && (i % w0c >= border) && (j / w1c >= border) && (j % w1c >= border);
            ok = ok && (i / w0c < hi_y0) && (i % w0c < hi_x0) && (j / w1c < hi_y1) && (j % w1c < hi_x1);
        }
        flags[(int64_t)n * L + i] = ok ? 1 : 0;
        local += ok ? 1 : 0;
    }
    atomicAdd(&s_cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) counts[n] = s_cnt;
}

// Ordered compaction + keypoint epilogue; one workgroup per pair.
__global__ __launch_bounds__(1024) void cm_compact(const unsigned long long* __restrict__ row_best,
                                                   const uint8_t* __restrict__ flags,
                                                   const int32_t* __restrict__ counts, int N, int L, int w0c,
                                                   int w1c, const float* __restrict__ scale0,
                                                   const float* __restrict__ scale1, float coarse_scale,
                                                   int64_t* __restrict__ b_ids, int64_t* __restrict__ i_ids,
                                                   int64_t* __restrict__ j_ids, float* __restrict__ mconf,
                                                   float* __restrict__ mkpts0, float* __restrict__ mkpts1,
                                                   int32_t* __restrict__ total) {
    const int n = blockIdx.x;
    __shared__ int s_wave[16];
    int base = 0;
    for (int b = 0; b < n; ++b) base += counts[b];
    if (n == N - 1 && threadIdx.x == 0) *total = base + counts[n];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (x, y) scale = coarse_scale * scale[b][{1,0}]   (coarse_matching.py:239-247)
    const float s0x = scale0 ? coarse_scale * scale0[n * 2 + 1] : coarse_scale;
    const float s0y = scale0 ? coarse_scale * scale0[n * 2 + 0] : coarse_scale;
    const float s1x = scale1 ? coarse_scale * scale1[n * 2 + 1] : coarse_scale;
    const float s1y = scale1 ? coarse_scale * scale1[n * 2 + 0] : coarse_scale;
    for (int i0 = 0; i0 < L; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        const bool f = i < L && flags[(int64_t)n * L + i];
        const unsigned long long bal = __ballot(f);
        const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
            const int c = s_wave[w];
            if (w < wave) off += c;
            tot += c;
        }
        if (f) {
            const int pos = base + off + in_wave;
            const unsigned long long key = row_best[(int64_t)n * L + i];
            const int j = (int)(~(unsigned)(key & 0xffffffffull));
            b_ids[pos] = n;
            i_ids[pos] = i;
            j_ids[pos] = j;
            mconf[pos] = __uint_as_float((unsigned)(key >> 32));
            mkpts0[pos * 2 + 0] = (float)(i % w0c) * s0x;
            mkpts0[pos * 2 + 1] = (float)(i / w0c) * s0y;
            mkpts1[pos * 2 + 0] = (float)(j % w1c) * s1x;
            mkpts1[pos * 2 + 1] = (float)(j / w1c) * s1y;
        }
        base += tot;
        __syncthreads();
    }
}

struct Workspace {
    float2 *row_part, *col_part, *row_stat, *col_stat;
    unsigned long long* row_best;
    unsigned int* col_best;
    uint8_t* flags;
    int32_t* counts;
    int32_t* extents;        // [N][4] valid (h0, w0, h1, w1) of padded frames (masked entry point)
    float2* cand;            // [N][ntn][L][CAND_SLOTS_MAX]
    uint8_t* cand_cnt;       // [N][ntn][L]
    size_t bytes;
};

Workspace carve(void* base, int N, int L, int S) {
    // col_part holds one partial per 128-row block: ceil(L/128) for cm_gemm and cm_gemm_sf2_cand, 2*ceil(L/256) (>= that) for
    // the two half tiles of cm_gemm_sf
    const int ntm = 2 * ((L + SF_BM - 1) / SF_BM), ntn = (S + BN - 1) / BN;
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(base) + off);
        off += align_up(bytes, 256);
        return p;
    };
    w.row_part = reinterpret_cast<float2*>(take((size_t)N * ntn * L * sizeof(float2)));
    w.col_part = reinterpret_cast<float2*>(take((size_t)N * ntm * S * sizeof(float2)));
    w.row_stat = reinterpret_cast<float2*>(take((size_t)N * L * sizeof(float2)));
    w.col_stat = reinterpret_cast<float2*>(take((size_t)N * S * sizeof(float2)));
    w.row_best = reinterpret_cast<unsigned long long*>(take((size_t)N * L * 8));
    w.col_best = reinterpret_cast<unsigned int*>(take((size_t)N * S * 4));
    w.flags = reinterpret_cast<uint8_t*>(take((size_t)N * L));
    w.counts = reinterpret_cast<int32_t*>(take((size_t)N * 4));
    w.extents = reinterpret_cast<int32_t*>(take((size_t)N * 16));
    w.cand_cnt = reinterpret_cast<uint8_t*>(take((size_t)N * ntn * L));
    w.cand = reinterpret_cast<float2*>(take((size_t)N * ntn * L * CAND_SLOTS_MAX * sizeof(float2)));
    w.bytes = off;
    return w;
}

bool is_pow4(int c) { return c > 0 && (c & (c - 1)) == 0 && (__builtin_ctz(c) % 2 == 0); }

template <int MODE>
void launch_gemm(const GemmArgs& g, int N, bool prescale, hipStream_t stream) {
    dim3 grid(g.ntm * g.ntn, N), blk(256);
    // > 64 KiB of dynamic LDS needs the opt-in attribute (set once per kernel instance).
    static dfsfm::SmemAttr smem_attr[2];
    if (prescale) {
        smem_attr[1].ensure(reinterpret_cast<const void*>(&cm_gemm<MODE, true>), SMEM_BYTES);
        hipLaunchKernelGGL((cm_gemm<MODE, true>), grid, blk, SMEM_BYTES, stream, g);
    } else {
        smem_attr[0].ensure(reinterpret_cast<const void*>(&cm_gemm<MODE, false>), SMEM_BYTES);
        hipLaunchKernelGGL((cm_gemm<MODE, false>), grid, blk, SMEM_BYTES, stream, g);
    }
}

int check_common(const float* f0, const float* f1, int N, int L, int S, int C, float temperature,
                 void* ws, size_t ws_bytes) {
    if (!f0 || !f1 || !ws) return DFSFM_E_BADARG;
    if (N <= 0 || L <= 0 || S <= 0 || C <= 0 || !(temperature > 0.f)) return DFSFM_E_BADARG;
    if (C % BK != 0 || N > 65535) return DFSFM_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(f0) & 15) || (reinterpret_cast<uintptr_t>(f1) & 15)) return DFSFM_E_UNSUPPORTED;
    if (ws_bytes < dfsfm_coarse_match_workspace(N, L, S)) return DFSFM_E_WORKSPACE;
    return DFSFM_OK;
}

GemmArgs make_args(const float* f0, const float* f1, int L, int S, int C, float temperature, float thr,
                   const Workspace& w, bool* prescale) {
    GemmArgs g{};
    g.f0 = f0; g.f1 = f1; g.L = L; g.S = S; g.C = C;
    g.ntm = (L + BM - 1) / BM; g.ntn = (S + BN - 1) / BN;
    // feat / sqrt(C) on both operands (coarse_matching.py:103-104).  For C a power of 4 the
    // division is exact and commutes with the dot product, so it is folded into one multiply.
    *prescale = !is_pow4(C);
    g.op_div = *prescale ? sqrtf((float)C) : 1.f;
    g.acc_mul = *prescale ? 1.f : 1.f / (float)C;
    g.temperature = temperature; g.thr = thr;
    g.row_part = w.row_part; g.col_part = w.col_part; g.row_stat = w.row_stat; g.col_stat = w.col_stat;
    g.row_best = w.row_best; g.col_best = w.col_best; g.conf_out = nullptr;
    return g;
}

}  // namespace

extern "C" size_t dfsfm_coarse_match_workspace(int N, int L, int S) {
    if (N <= 0 || L <= 0 || S <= 0) return 0;
    return carve(nullptr, N, L, S).bytes;
}

namespace {

template <int MODE>
void launch_gemm_sf(const GemmSfArgs& g, int N, hipStream_t stream) {
    static dfsfm::SmemAttr smem_attr;
    smem_attr.ensure(reinterpret_cast<const void*>(&cm_gemm_sf<MODE>), SF_SMEM);
    hipLaunchKernelGGL(cm_gemm_sf<MODE>, dim3((g.ntiles + 7) / 8 * 8, N), dim3(512), SF_SMEM, stream, g);
}

// feat0/feat1 fp32, or (f0h, f0l, f1h, f1l) fp16 split planes of the same tensors.
int coarse_match_impl(const float* feat0, const float* feat1, const _Float16* f0h, const _Float16* f0l,
                      const _Float16* f1h, const _Float16* f1l, int N, int L, int S, int C, float temperature,
                      float thr, int border, int h0c, int w0c, int h1c, int w1c, const float* scale0,
                      const float* scale1, float coarse_scale, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids,
                      float* mconf, float* mkpts0, float* mkpts1, int32_t* count, void* workspace,
                      size_t workspace_bytes, hipStream_t stream, const char* what, const uint8_t* mask0 = nullptr,
                      const uint8_t* mask1 = nullptr) {
    if (!b_ids || !i_ids || !j_ids || !mconf || !mkpts0 || !mkpts1 || !count) return DFSFM_E_BADARG;
    if (h0c * w0c != L || h1c * w1c != S || border < 0) return DFSFM_E_BADARG;
    if (!(thr >= 0.f)) return DFSFM_E_UNSUPPORTED;   // best-candidate words use 0 as "none"
    if ((mask0 == nullptr) != (mask1 == nullptr) || (mask0 && !f0h)) return DFSFM_E_UNSUPPORTED;   // masks: split entry point only
    Workspace w = carve(workspace, N, L, S);
    int nparts;
    if (f0h) {
        // the 1/sqrt(C) operand scaling folds into one exact multiply only for C a power of 4
        if (!is_pow4(C) || (int64_t)L * C * 2 >= (1ll << 32) || (int64_t)S * C * 2 >= (1ll << 32))
            return DFSFM_E_UNSUPPORTED;
        GemmSfArgs g{};
        g.f0h = f0h; g.f0l = f0l; g.f1h = f1h; g.f1l = f1l; g.L = L; g.S = S; g.C = C;
        const int ntm = (L + SF_BM - 1) / SF_BM;
        g.ntn = (S + SF_BN - 1) / SF_BN;
        g.ntiles = (unsigned)(ntm * g.ntn);
        g.nhalf = nparts = 2 * ntm;
        g.acc_mul = 1.f / (float)C; g.temperature = temperature; g.thr = thr;
        g.row_part = w.row_part; g.col_part = w.col_part; g.row_stat = w.row_stat; g.col_stat = w.col_stat;
        g.row_best = w.row_best; g.col_best = w.col_best;
        g.mask0 = mask0; g.mask1 = mask1;
        // Single-GEMM path: at most floor(1 / (thr e^-1e-3)) entries of a row (or column) can pass the row gate, so for
        // thr >= 1/8 a fixed number of candidate slots per (row, column tile) holds every possible match and the second
        // correlation pass is replaced by an O(candidates) evaluation (thr < 1/8 takes the two-pass path below).
        const int slots = thr > 0.f ? (int)floorf(1.f / (thr * 0.998f)) : CAND_SLOTS_MAX + 1;
        if (slots <= CAND_SLOTS_MAX) {
            g.cand = w.cand; g.cand_cnt = w.cand_cnt; g.slots = slots;
            {
                const int ntm2 = (L + SF2_BM - 1) / SF2_BM;
                g.ntiles = (unsigned)(ntm2 * g.ntn);
                g.nhalf = nparts = ntm2;
                // tile rows per traversal group: about one group per XCD range (8 ranges), at most 12 (1.5 MB of f0 at C = 256
                // in a 4-MB L2 that f1 is streaming through)
                g.sb = std::min(12, std::max(1, (ntm2 + 7) / 8));
                static dfsfm::SmemAttr smem_attr2;
                smem_attr2.ensure(reinterpret_cast<const void*>(&cm_gemm_sf2_cand), dfsfm_sf::V2S<1>::SMEM);
                hipLaunchKernelGGL(cm_gemm_sf2_cand, dim3((g.ntiles + 7) / 8 * 8, N), dim3(256), dfsfm_sf::V2S<1>::SMEM, stream, g);
            }
            hipLaunchKernelGGL(cm_reduce_stats, dim3((L + S + 255) / 256, N), dim3(256), 0, stream, w.row_part,
                               w.col_part, w.row_stat, w.col_stat, w.row_best, w.col_best, L, S, nparts, g.ntn);
            hipLaunchKernelGGL(cm_eval, dim3((L + 255) / 256, N), dim3(256), 0, stream, w.cand, w.cand_cnt, w.row_stat,
                               w.col_stat, w.row_best, w.col_best, L, S, g.ntn, slots, thr);
        } else {
            launch_gemm_sf<MODE_STATS>(g, N, stream);
            hipLaunchKernelGGL(cm_reduce_stats, dim3((L + S + 255) / 256, N), dim3(256), 0, stream, w.row_part,
                               w.col_part, w.row_stat, w.col_stat, w.row_best, w.col_best, L, S, nparts, g.ntn);
            launch_gemm_sf<MODE_SELECT>(g, N, stream);
        }
    } else {
        bool prescale;
        GemmArgs g = make_args(feat0, feat1, L, S, C, temperature, thr, w, &prescale);
        launch_gemm<MODE_STATS>(g, N, prescale, stream);
        hipLaunchKernelGGL(cm_reduce_stats, dim3((L + S + 255) / 256, N), dim3(256), 0, stream, w.row_part,
                           w.col_part, w.row_stat, w.col_stat, w.row_best, w.col_best, L, S, g.ntm, g.ntn);
        launch_gemm<MODE_SELECT>(g, N, prescale, stream);
    }
    if (mask0 && border > 0)
        hipLaunchKernelGGL(cm_mask_extents, dim3(N, 2), dim3(256), 0, stream, mask0, mask1, w.extents, h0c, w0c, h1c, w1c);
    hipLaunchKernelGGL(cm_select, dim3(N), dim3(1024), 0, stream, w.row_best, w.col_best, w.flags, w.counts, L,
                       S, thr, border, w0c, w1c, mask0 && border > 0 ? w.extents : nullptr);
    hipLaunchKernelGGL(cm_compact, dim3(N), dim3(1024), 0, stream, w.row_best, w.flags, w.counts, N, L, w0c,
                       w1c, scale0, scale1, coarse_scale, b_ids, i_ids, j_ids, mconf, mkpts0, mkpts1, count);
    return check_launch(what);
}

}  // namespace

extern "C" int dfsfm_coarse_match_f32(const float* feat0, const float* feat1, int N, int L, int S, int C,
                                      float temperature, float thr, int border, int h0c, int w0c, int h1c,
                                      int w1c, const float* scale0, const float* scale1, float coarse_scale,
                                      int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf,
                                      float* mkpts0, float* mkpts1, int32_t* count, void* workspace,
                                      size_t workspace_bytes, void* stream_) {
    int rc = check_common(feat0, feat1, N, L, S, C, temperature, workspace, workspace_bytes);
    if (rc != DFSFM_OK) return rc;
    return coarse_match_impl(feat0, feat1, nullptr, nullptr, nullptr, nullptr, N, L, S, C, temperature, thr,
                             border, h0c, w0c, h1c, w1c, scale0, scale1, coarse_scale, b_ids, i_ids, j_ids,
                             mconf, mkpts0, mkpts1, count, workspace, workspace_bytes,
                             static_cast<hipStream_t>(stream_), "dfsfm_coarse_match_f32");
}

extern "C" int dfsfm_coarse_match_split(const void* feat0_hi, const void* feat0_lo, const void* feat1_hi,
                                        const void* feat1_lo, int N, int L, int S, int C, float temperature,
                                        float thr, int border, int h0c, int w0c, int h1c, int w1c,
                                        const float* scale0, const float* scale1, float coarse_scale,
                                        int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf,
                                        float* mkpts0, float* mkpts1, int32_t* count, void* workspace,
                                        size_t workspace_bytes, void* stream_) {
    if (!feat0_hi || !feat0_lo || !feat1_hi || !feat1_lo || !workspace) return DFSFM_E_BADARG;
    if (N <= 0 || L <= 0 || S <= 0 || C <= 0 || !(temperature > 0.f)) return DFSFM_E_BADARG;
    if (C % BK != 0 || N > 65535) return DFSFM_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(feat0_hi) | reinterpret_cast<uintptr_t>(feat0_lo) |
         reinterpret_cast<uintptr_t>(feat1_hi) | reinterpret_cast<uintptr_t>(feat1_lo)) & 15)
        return DFSFM_E_UNSUPPORTED;
    if (workspace_bytes < dfsfm_coarse_match_workspace(N, L, S)) return DFSFM_E_WORKSPACE;
    return coarse_match_impl(nullptr, nullptr, static_cast<const _Float16*>(feat0_hi),
                             static_cast<const _Float16*>(feat0_lo), static_cast<const _Float16*>(feat1_hi),
                             static_cast<const _Float16*>(feat1_lo), N, L, S, C, temperature, thr, border, h0c,
                             w0c, h1c, w1c, scale0, scale1, coarse_scale, b_ids, i_ids, j_ids, mconf, mkpts0,
                             mkpts1, count, workspace, workspace_bytes, static_cast<hipStream_t>(stream_),
                             "dfsfm_coarse_match_split");
}

extern "C" int dfsfm_coarse_match_split_masked(const void* feat0_hi, const void* feat0_lo, const void* feat1_hi,
                                               const void* feat1_lo, const uint8_t* mask0, const uint8_t* mask1, int N,
                                               int L, int S, int C, float temperature, float thr, int border, int h0c,
                                               int w0c, int h1c, int w1c, const float* scale0, const float* scale1,
                                               float coarse_scale, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids,
                                               float* mconf, float* mkpts0, float* mkpts1, int32_t* count, void* workspace,
                                               size_t workspace_bytes, void* stream_) {
    if (!feat0_hi || !feat0_lo || !feat1_hi || !feat1_lo || !workspace || !mask0 || !mask1) return DFSFM_E_BADARG;
    if (N <= 0 || L <= 0 || S <= 0 || C <= 0 || !(temperature > 0.f)) return DFSFM_E_BADARG;
    if (C % BK != 0 || N > 65535) return DFSFM_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(feat0_hi) | reinterpret_cast<uintptr_t>(feat0_lo) |
         reinterpret_cast<uintptr_t>(feat1_hi) | reinterpret_cast<uintptr_t>(feat1_lo)) & 15)
        return DFSFM_E_UNSUPPORTED;
    if (workspace_bytes < dfsfm_coarse_match_workspace(N, L, S)) return DFSFM_E_WORKSPACE;
    return coarse_match_impl(nullptr, nullptr, static_cast<const _Float16*>(feat0_hi),
                             static_cast<const _Float16*>(feat0_lo), static_cast<const _Float16*>(feat1_hi),
                             static_cast<const _Float16*>(feat1_lo), N, L, S, C, temperature, thr, border, h0c,
                             w0c, h1c, w1c, scale0, scale1, coarse_scale, b_ids, i_ids, j_ids, mconf, mkpts0,
                             mkpts1, count, workspace, workspace_bytes, static_cast<hipStream_t>(stream_),
                             "dfsfm_coarse_match_split_masked", mask0, mask1);
}

extern "C" int dfsfm_coarse_conf_matrix_f32(const float* feat0, const float* feat1, int N, int L, int S, int C,
                                            float temperature, float* conf, void* workspace,
                                            size_t workspace_bytes, void* stream_) {
    int rc = check_common(feat0, feat1, N, L, S, C, temperature, workspace, workspace_bytes);
    if (rc != DFSFM_OK) return rc;
    if (!conf) return DFSFM_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Workspace w = carve(workspace, N, L, S);
    bool prescale;
    GemmArgs g = make_args(feat0, feat1, L, S, C, temperature, 0.f, w, &prescale);
    g.conf_out = conf;
    launch_gemm<MODE_STATS>(g, N, prescale, stream);
    hipLaunchKernelGGL(cm_reduce_stats, dim3((L + S + 255) / 256, N), dim3(256), 0, stream, w.row_part,
                       w.col_part, w.row_stat, w.col_stat, w.row_best, w.col_best, L, S, g.ntm, g.ntn);
    launch_gemm<MODE_CONF>(g, N, prescale, stream);
    return check_launch("dfsfm_coarse_conf_matrix_f32");
}
