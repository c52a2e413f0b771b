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
// One family: out[m, co] = sum_{ky,kx,ci} in[pix(m,ky,kx), ci] * w[co, ky, kx, ci], NHWC, M = Nimg*Ho*Wo output
// pixels, weights pre-split to fp16 [Npad][Kpad] hi / lo; a linear layer is the 1x1 case with a row stride.
//   conv_gemm_sf_same_kernel<BN,KW>  split-plane inputs, stride-1 "same" 3x3 / 5x5 convs and every 1x1 / linear:
//                                    the activation-reuse, ping-pong LDS-DMA main loop of sf_gemm.h
//   conv_gemm_sf_kernel<BN>          split-plane inputs, any geometry (stride-2 convs, pad-0 5x5): K flattened over
//                                    (ky,kx,ci), 3-stage LDS-DMA ring of whole slabs, lock-step waves
//   conv_gemm_kernel<VEC_A>          fp32 inputs (image stem): splits in registers while staging, 128x128 tile
// All share sf_epilogue-style fusion of bias (folded BN), residual, ReLU, optional LayerNorm, fp32 / split stores.
// (The -DDFSFM_ABL_* ablation switches of rounds 1-2 are gone from these sources; their measurements are in DESIGN.md section 3.)
#include "common.h"
#include "sf_gemm.h"
#include <cstdlib>

namespace {

using namespace dfsfm;
using namespace dfsfm_sf;

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;        // v1 tile; BK = 32 and BM2 = 256 come from sf_gemm.h
constexpr int TILE_B = BM * BK * 2;                 // bytes of one fp16 tile (8 KB)
constexpr int STAGE_B = 4 * TILE_B;                 // A_hi, A_lo, B_hi, B_lo
constexpr int SMEM_BYTES = 2 * STAGE_B;             // double buffered: 64 KB
constexpr int LUT_MAX = 256;                        // scalar-gather path: k -> (ky,kx,ci) table entries

__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) { split_f32(x, hi, lo); }

__device__ __forceinline__ void split4(const f32x4 v, half4& hi, half4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        _Float16 h, l;
        split_f32(v[e], h, l);
        hi[e] = h;
        lo[e] = l;
    }
}


struct RowGeom {           // one activation row handled by this thread
    int iy0, ix0;
    int64_t base;
    bool ok;
};

template <bool VEC_A>
__device__ __forceinline__ void load_slab(const ConvArgs& g, const int* lut, int k0, int c4, int tid, int n0, const RowGeom& r0,
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
                    int ky, kx, ci;
                    if (lut) {                      // (ky, kx, ci) of every k, tabulated once per workgroup
                        const int t = lut[ke];
                        ky = t & 255; kx = (t >> 8) & 255; ci = t >> 16;
                    } else {
                        const int tap = ke / g.Cin;
                        ci = ke - tap * g.Cin;
                        ky = tap / g.kw;
                        kx = tap - ky * g.kw;
                    }
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

// the LDS-staged epilogue shared with the split-input kernels (defined below)
template <int BN_, int BM_, int NT>
__device__ __forceinline__ void sf_epilogue(const ConvArgs& g, char* smem, f32x16 (&accm)[2][BN_ * BM_ / (32 * NT)],
                                            f32x16 (&accx)[2][BN_ * BM_ / (32 * NT)], int64_t m0, int n0, int tid,
                                            int wr, int wc, int col, int kgrp);
constexpr int V1_EPI_BYTES = BM * (BN + 4) * 4;     // its fp32 staging tile: 66 KB (> the 64 KB operand ring)

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
    // scalar-gather path (Cin = 1 or 3): k -> (ky, kx, ci) table in LDS instead of two divisions per element
    const int* lut = nullptr;
    if (!VEC_A && g.Kpad <= LUT_MAX && g.kh < 256 && g.kw < 256) {
        int* l = reinterpret_cast<int*>(smem + SMEM_BYTES);
        for (int k = tid; k < g.K; k += 256) {
            const int tap = k / g.Cin, ci = k - tap * g.Cin;
            const int ky = tap / g.kw, kx = tap - ky * g.kw;
            l[k] = ky | (kx << 8) | (ci << 16);
        }
        __syncthreads();
        lut = l;
    }
#define GLOAD(k0) load_slab<VEC_A>(g, lut, (k0), c4, tid, n0, g0, g1, g2, g3, a0, a1, a2, a3, bh0, bh1, bl0, bl1)
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
    // ---- epilogue: the tile goes through LDS (the operand ring is dead) so that every thread owns 8 consecutive
    // channels of a row: bias (folded BN), residual, ReLU, 32-byte fp32 / 16+16-byte split stores.  The per-element form
    // this replaces wrote the split planes of the 7x7 stem (629 MB per step) two bytes per lane.
    sf_epilogue<BN, BM, 256>(g, smem, accm, accx, m0, n0, tid, wr, wc, col, kgrp);
}

// =================================================================================================
// v2: pre-split activations, LDS-DMA pipeline.  256 x BN tile, 8 waves (4 x 2, 64 x BN/2 each),
// BK=32, THREE LDS stages filled by buffer_load ... lds (16 B/lane, no VGPR round trip, no VALU):
// slab kt+2 is in flight while slab kt is multiplied; counted s_waitcnt vmcnt + raw s_barrier keep
// the DMA queue alive across barriers (MI355X guide: T3/T4, "pipelining across barriers").
// Out-of-image taps / rows past M use an out-of-range buffer offset, which the hardware
// zero-fills.  The epilogue stages the tile through LDS so every store is 16-32 bytes per lane.
// =================================================================================================
constexpr int NST = 3;

template <int BN_>
struct V2 {
    static constexpr int A_PLANE = BM2 * 64;                 // bytes: 256 rows x 32 halves
    static constexpr int B_PLANE = BN_ * 64;
    static constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
    static constexpr int RING = NST * STAGE;
    static constexpr int B_INSTR = (BN_ / 16) * 2 / 8;       // B LDS-DMA instructions per wave per slab
    static constexpr int G = 4 + B_INSTR;                    // LDS-DMA instructions per wave per slab
    static constexpr int TILE_LD = BN_ + 4;                  // fp32 epilogue tile row (floats)
    static constexpr int SMEM = (RING > BM2 * TILE_LD * 4) ? RING : BM2 * TILE_LD * 4;
};


// Shared epilogue of the split-input kernels: the accumulator tile goes through LDS so each thread owns 8
// consecutive channels of one output row: bias (folded BN), fp32 / split residual, ReLU, then 32-byte fp32
// stores and/or 16+16-byte split stores.  Must be entered with no LDS-DMA in flight.
// NT = threads of the workgroup (512: 8 waves; 256: the 4-wave linear kernel); every wave owns 64 rows.
// Second half of the epilogue: ROWS rows of a staged fp32 tile (row pitch BN_ + 4 floats), global rows m0 .. m0 + ROWS - 1.
template <int BN_, int ROWS, int NT>
__device__ __forceinline__ void sf_epilogue_rows(const ConvArgs& g, const float* tile, int64_t m0, int n0, int tid) {
    constexpr int TILE_LD_ = BN_ + 4;                        // fp32 staging-tile row (floats)
    constexpr int BM_ = ROWS;
    // Every thread owns one 8-channel chunk (fixed: NT % CH == 0) of IT rows.  All memory operations of the IT
    // rows are issued before anything waits on them: tile reads, then the residual loads, then arithmetic and
    // stores -- the epilogue is latency-bound otherwise (one dependent global load per row).
    constexpr int CH = BN_ / 8;                               // 8-channel chunks per row
    constexpr int RPI = NT / CH;                              // rows per pass
    constexpr int IT = BM_ / RPI;                             // passes (8 for BN=128, 4 for BN=64; 8 for 128 x 256)
    const int c8 = (tid % CH) * 8, rr = tid / CH;
    const int n = n0 + c8;
    if (n >= g.Cout && n >= g.Cout_s) return;
    const bool full = n + 8 <= g.Cout;
    const bool ln = g.lng != nullptr;                         // host guarantees Cout == BN_ (one N tile, all chunks full)
    float bv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bv[q] = (g.bias && n + q < g.Cout) ? g.bias[n + q] : 0.f;
    float v[IT][8];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int r = it * RPI + rr;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(tile + r * TILE_LD_ + c8);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(tile + r * TILE_LD_ + c8 + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[it][q] = t0[q] + bv[q]; v[it][4 + q] = t1[q] + bv[4 + q]; }
    }
    if (ln) {
        // LayerNorm over the Cout channels of each row: the CH threads that own a row are consecutive lanes of one
        // wave, so the two row reductions are CH-wide butterflies.  Same arithmetic as layernorm_kernel
        // (epilogue.hip): mean, centred sum of squares, 1/sqrt(var + eps).
        float gm[8], bt[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { gm[q] = g.lng[n + q]; bt[q] = g.lnb[n + q]; }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[it][q];
#pragma unroll
            for (int off = CH / 2; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            const float mean = s / (float)BN_;
            float sq = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sq += (v[it][q] - mean) * (v[it][q] - mean);
#pragma unroll
            for (int off = CH / 2; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
            const float rstd = 1.f / sqrtf(sq / (float)BN_ + g.ln_eps);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[it][q] = (v[it][q] - mean) * rstd * gm[q] + bt[q];
        }
    }
    if (g.resh) {              // split residual: its channel count is padded to a multiple of 8 with zeros
        half8 rh[IT], rl[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int64_t m = m0 + it * RPI + rr;
            const int64_t mm = m < g.M ? m : g.M - 1;
            rh[it] = *reinterpret_cast<const half8*>(g.resh + mm * g.ldr + n);
            rl[it] = *reinterpret_cast<const half8*>(g.resl + mm * g.ldr + n);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[it][q] += (float)rh[it][q] + (float)rl[it][q] * (1.f / 2048.f);
    } else if (g.res) {
        if (full) {
            f32x4 r0[IT], r1[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int64_t m = m0 + it * RPI + rr;
                const int64_t mm = m < g.M ? m : g.M - 1;
                r0[it] = *reinterpret_cast<const f32x4*>(g.res + mm * g.ldr + n);
                r1[it] = *reinterpret_cast<const f32x4*>(g.res + mm * g.ldr + n + 4);
            }
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[it][q] += r0[it][q]; v[it][4 + q] += r1[it][q]; }
        } else {               // chunk straddles Cout (e.g. 196 = 24*8 + 4): per-channel guards
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int64_t m = m0 + it * RPI + rr;
                if (m < g.M) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (n + q < g.Cout) v[it][q] += g.res[m * g.ldr + n + q];
                }
            }
        }
    }
    if (g.relu == 2) {     // nn.LeakyReLU() (one MatchFormer FPN conv): its own pass, so that the ReLU / linear epilogues of
#pragma unroll         // every other layer keep their instruction stream (a per-element mode select cost 2 % of both steps)
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[it][q] = v[it][q] > 0.f ? v[it][q] : 0.01f * v[it][q];
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int64_t m = m0 + it * RPI + rr;
        if (m >= g.M) continue;
        if (g.relu == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[it][q] = fmaxf(v[it][q], 0.f);
        }
        if (!full) {          // only the chunk that straddles Cout (196 = 24 * 8 + 4): padded split channels are zeros
#pragma unroll
            for (int q = 0; q < 8; ++q) v[it][q] = n + q < g.Cout ? v[it][q] : 0.f;
        }
        if (g.out) {
            if (full) {
                *reinterpret_cast<f32x4*>(g.out + m * g.ldo + n) = f32x4{v[it][0], v[it][1], v[it][2], v[it][3]};
                *reinterpret_cast<f32x4*>(g.out + m * g.ldo + n + 4) = f32x4{v[it][4], v[it][5], v[it][6], v[it][7]};
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (n + q < g.Cout) g.out[m * g.ldo + n + q] = v[it][q];
            }
        }
        if (g.outh && n < g.Cout_s) {          // Cout_s % 8 == 0: whole chunk
            half8 h, l;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                _Float16 a, b;
                split1(v[it][q], a, b);
                h[q] = a;
                l[q] = b;
            }
            *reinterpret_cast<half8*>(g.outh + m * g.ldo_s + n) = h;
            *reinterpret_cast<half8*>(g.outl + m * g.ldo_s + n) = l;
        }
    }
}

template <int BN_, int BM_ = BM2, int NT = 512>
__device__ __forceinline__ void sf_epilogue(const ConvArgs& g, char* smem, f32x16 (&accm)[2][BN_ * BM_ / (32 * NT)],
                                            f32x16 (&accx)[2][BN_ * BM_ / (32 * NT)], int64_t m0, int n0, int tid,
                                            int wr, int wc, int col, int kgrp) {
    constexpr int TILE_LD_ = BN_ + 4;                        // fp32 staging-tile row (floats)
    constexpr int NJ = BN_ * BM_ / (32 * NT);                // 32-wide column blocks per wave (waves: BM_/64 x NT/BM_)
    constexpr int WCOLS = NJ * 32;                            // columns per wave
    __syncthreads();
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[(wr * 64 + i * 32 + mfma32_row(r, kgrp)) * TILE_LD_ + wc * WCOLS + j * 32 + col] =
                    accm[i][j][r] + accx[i][j][r] * (1.f / 2048.f);
    __syncthreads();
    sf_epilogue_rows<BN_, BM_, NT>(g, tile, m0, n0, tid);
}

// Epilogue of the 16 x 16 x 32 main loop (sf_same_mainloop16): block (i, j) of wave (wr, wc) holds rows wr 64 + 16 i + 4 (lane >> 4) + r,
// column wc WCOLS + 16 j + (lane & 15) in register r.  The four lane groups of a store instruction hit rows 4 apart (bank offsets
// 0 / 16 / 32 / 48 with the 132-float pitch) with 16 consecutive columns each: conflict-free.
template <int BN_, int BM_ = BM2, int NT = 512>
__device__ __forceinline__ void sf_epilogue16(const ConvArgs& g, char* smem, f32x4 (&accm)[4][BN_ * BM_ / (16 * NT)],
                                              f32x4 (&accx)[4][BN_ * BM_ / (16 * NT)], int64_t m0, int n0, int tid, int wr, int wc,
                                              int lane) {
    constexpr int TILE_LD_ = BN_ + 4;
    constexpr int NJ = BN_ * BM_ / (16 * NT);                // 16-wide column blocks per wave
    constexpr int WCOLS = NJ * 16;
    __syncthreads();
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tile[(wr * 64 + i * 16 + 4 * (lane >> 4) + r) * TILE_LD_ + wc * WCOLS + j * 16 + (lane & 15)] =
                    accm[i][j][r] + accx[i][j][r] * (1.f / 2048.f);
    __syncthreads();
    sf_epilogue_rows<BN_, BM_, NT>(g, tile, m0, n0, tid);
}

template <int BN_>
__global__ __launch_bounds__(512, 1) void conv_gemm_sf_kernel(ConvArgs g) {
    using T = V2<BN_>;
    constexpr int NJ = BN_ / 64;                              // 32-wide column tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kgrp = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    // N tiles of one M tile are adjacent in dispatch order: they share the activation tile in L2 / MALL
    const int ntn = (g.Cout + BN_ - 1) / BN_;
    const unsigned tile_id = xcd_band_tile(blockIdx.x, gridDim.x >> 3);
    if (tile_id >= g.ntiles) return;
    const int64_t m0 = (int64_t)(tile_id / ntn) * BM2;
    const int n0 = (tile_id % ntn) * BN_;

    const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)g.xh, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)g.xl, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc((void*)g.wh, 0, g.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)g.wl, 0, g.wbytes, 0x00020000);

    // lane -> (row within a 16-row group, 16-byte k-slot); XOR swizzle applied on the SOURCE side
    const int lrow = lane >> 2;
    const int lslot = (lane & 3) ^ ((lane >> 4) & 3);        // logical slot of this lane (all groups)
    // two activation rows per lane: groups `wave` and `wave + 8`
    int iy0[2], ix0[2];
    unsigned abase[2];
    bool aok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t m = m0 + (wave + 8 * q) * 16 + lrow;
        aok[q] = m < g.M;
        const int64_t mm = aok[q] ? m : 0;
        const int ox = (int)(mm % g.Wo);
        const int64_t t = mm / g.Wo;
        const int oy = (int)(t % g.Ho);
        const int64_t n = t / g.Ho;
        iy0[q] = oy * g.stride - g.pad;
        ix0[q] = ox * g.stride - g.pad;
        abase[q] = (unsigned)(n * g.sxn);
    }
    // running decomposition of this lane's k index (advances by BK per slab); branch-free updates so the
    // whole main loop stays one basic block and the LDS-DMA issues can sit between MFMAs
    int ci = lslot * 8, ky = 0, kx = 0, kcur = lslot * 8;
#pragma unroll 1
    while (ci >= g.Cin) { ci -= g.Cin; if (++kx == g.kw) { kx = 0; ++ky; } }
    unsigned boff = (unsigned)(((int64_t)(n0 + lrow) * g.Kpad + lslot * 8) * 2);   // + group*16 rows, + k0
    unsigned offA[2], offB[2];
    auto addr = [&]() __attribute__((always_inline)) {      // offsets of the next slab's pieces, then advance
        const bool kin = kcur < g.K;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int iy = iy0[q] + ky, ix = ix0[q] + kx;
            const bool ok = aok[q] & kin & (iy >= 0) & (iy < g.H) & (ix >= 0) & (ix < g.W);
            const unsigned off = (abase[q] + (unsigned)iy * (unsigned)g.sxh + (unsigned)ix * (unsigned)g.ldx + (unsigned)ci) * 2u;
            offA[q] = ok ? off : g.xbytes;                  // out of range -> hardware writes zeros
        }
#pragma unroll
        for (int j = 0; j < T::B_INSTR; ++j) {
            const int grp = (wave + 8 * j) % (BN_ / 16);
            offB[j] = kcur - lslot * 8 < g.Kpad ? boff + (unsigned)grp * 16u * (unsigned)g.Kpad * 2u : g.wbytes;
        }
        kcur += BK;
        boff += BK * 2;
        ci += BK;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                       // up to 4 tap wraps per slab (Cin >= 8)
            const bool wrap = ci >= g.Cin;
            ci -= wrap ? g.Cin : 0;
            kx += wrap ? 1 : 0;
            const bool wrap2 = kx == g.kw;
            kx = wrap2 ? 0 : kx;
            ky += wrap2 ? 1 : 0;
        }
    };
    // pieces of a slab: A rows group `wave + 8q`, plane hi/lo (4 pieces), then the weight pieces
#define DMA_A(q, lo, stage)                                                                                   \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((lo) ? rxl : rxh,                                                \
                                             (lds_void*)(smem + (stage) * T::STAGE + (lo) * T::A_PLANE +      \
                                                         (wave + 8 * (q)) * 1024),                            \
                                             16, offA[q], 0, 0, 0)
#define DMA_B(j, stage)                                                                                       \
    do {                                                                                                      \
        const int ib_ = wave + 8 * (j);                                                                       \
        const int plane_ = ib_ / (BN_ / 16), grp_ = ib_ % (BN_ / 16);                                         \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(plane_ ? rwl : rwh,                                          \
                                                 (lds_void*)(smem + (stage) * T::STAGE + 2 * T::A_PLANE +     \
                                                             plane_ * T::B_PLANE + grp_ * 1024),              \
                                                 16, offB[j], 0, 0, 0);                                       \
    } while (0)
    auto issue_all = [&](int stage) __attribute__((always_inline)) {
        addr();
        DMA_A(0, 0, stage);
        DMA_A(0, 1, stage);
        DMA_A(1, 0, stage);
        DMA_A(1, 1, stage);
        DMA_B(0, stage);
        if constexpr (T::B_INSTR > 1) DMA_B(1, stage);
    };

    f32x16 accm[2][NJ], accx[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            accm[i][j] = f32x16{0};
            accx[i][j] = f32x16{0};
        }

    // Fragment sets F0 (k-step 0 of a slab) and F1 (k-step 1) are loaded one half-slab ahead of the MFMAs
    // that use them, so LDS latency hides behind matrix work.  One barrier per slab, placed between
    // the two k-steps: it publishes slab kt+1 (needed for the F0 prefetch) and frees stage (kt-1)%3.
    half8 a0h[2], a0l[2], b0h[NJ], b0l[NJ], a1h[2], a1l[2], b1h[NJ], b1l[NJ];
    auto read_frags = [&](const char* st, int ks, half8 (&ah)[2], half8 (&al)[2], half8 (&bh)[NJ],
                          half8 (&bl)[NJ]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = tile_off(wr * 64 + i * 32 + col, ks * 2 + kgrp);
            ah[i] = *reinterpret_cast<const half8*>(st + off);
            al[i] = *reinterpret_cast<const half8*>(st + T::A_PLANE + off);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int off = tile_off(wc * (BN_ / 2) + j * 32 + col, ks * 2 + kgrp);
            bh[j] = *reinterpret_cast<const half8*>(st + 2 * T::A_PLANE + off);
            bl[j] = *reinterpret_cast<const half8*>(st + 2 * T::A_PLANE + T::B_PLANE + off);
        }
    };
    auto mma = [&](const half8 (&ah)[2], const half8 (&al)[2], const half8 (&bh)[NJ], const half8 (&bl)[NJ])
                   __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], accm[i][j], 0, 0, 0);
                accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
            }
    };

    // Ring schedule (3 stages): at the mid-slab point of slab kt every read of stage kt%3 has been issued
    // (F0 one half-slab earlier, F1 at the top of this slab), so slab kt+3 is DMA'd into it right there:
    // two slabs are always in flight while one is being multiplied.  Every slab issues exactly G pieces
    // (past the end of K they are out-of-range = zero fill, no memory traffic), which keeps the vmcnt
    // arithmetic constant and the loop body branch-free; the pieces are spread between MFMA groups.
    const int nk = g.Kpad / BK;
    issue_all(0);
    issue_all(1);
    issue_all(2);
    wait_vmcnt<2 * T::G>();                                           // slab 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    read_frags(smem, 0, a0h, a0l, b0h, b0l);
    auto mma3 = [&](const half8& ah, const half8& al, const half8& bh, const half8& bl, f32x16& m, f32x16& x)
                    __attribute__((always_inline)) {
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, m, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, x, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, x, 0, 0, 0);
    };
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt % NST;
        read_frags(smem + stage * T::STAGE, 1, a1h, a1l, b1h, b1l);  // prefetch k-step 1 of this slab
        mma(a0h, a0l, b0h, b0l);                                      // k-step 0
        addr();                                                       // offsets of slab kt+3 (VALU only)
        wait_vmcnt<T::G>();                                           // slab kt+1 landed (kt+2 may still fly)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wave's reads of stage kt%3 are done
        __builtin_amdgcn_s_barrier();
        read_frags(smem + ((kt + 1) % NST) * T::STAGE, 0, a0h, a0l, b0h, b0l);   // prefetch next slab's k-step 0
        // k-step 1, with slab kt+3's DMA pieces (-> the stage just retired) spread between the MFMA groups
        DMA_A(0, 0, stage);
        mma3(a1h[0], a1l[0], b1h[0], b1l[0], accm[0][0], accx[0][0]);
        DMA_A(0, 1, stage);
        if constexpr (NJ == 2) {
            mma3(a1h[0], a1l[0], b1h[1], b1l[1], accm[0][1], accx[0][1]);
            DMA_A(1, 0, stage);
            DMA_A(1, 1, stage);
            mma3(a1h[1], a1l[1], b1h[0], b1l[0], accm[1][0], accx[1][0]);
            DMA_B(0, stage);
            DMA_B(1, stage);
            mma3(a1h[1], a1l[1], b1h[1], b1l[1], accm[1][1], accx[1][1]);
        } else {
            DMA_A(1, 0, stage);
            DMA_A(1, 1, stage);
            mma3(a1h[1], a1l[1], b1h[0], b1l[0], accm[1][0], accx[1][0]);
            DMA_B(0, stage);
        }
#if 1   // pinned interleave measured +5 % over the compiler's order (DMA issues hoisted in front of the MFMAs)
        // pin the interleave: F0 prefetch reads first, then {DMA piece(s), 3 MFMAs} groups
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * NJ, 0);
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        if constexpr (NJ == 2) {
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x20, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x20, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x20, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        }
#endif
    }
    wait_vmcnt<0>();                                                  // the zero-fill tail pieces, before LDS is reused
#undef DMA_A
#undef DMA_B

    sf_epilogue<BN_>(g, smem, accm, accx, m0, n0, tid, wr, wc, col, kgrp);
}

// "same" convolution / 1x1 / linear kernel: sf_same_mainloop (sf_gemm.h) + the shared epilogue.
template <int BN_, int KW, int WM = 4>
__global__ __launch_bounds__(512, 1) void conv_gemm_sf_same_kernel(ConvArgs g) {
    using S_ = VS<BN_, KW, WM>;
    constexpr int NJ = BN_ / (32 * S_::WN);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = (g.Cout + BN_ - 1) / BN_;
    const unsigned tile_id = xcd_band_tile(blockIdx.x, gridDim.x >> 3);
    if (tile_id >= g.ntiles) return;
    const int64_t m0 = (int64_t)(tile_id / ntn) * S_::BM;
    const int n0 = (tile_id % ntn) * BN_;
    if constexpr (WM == 4 || WM == 8) {      // the 256 x BN and 512 x 64 tiles: v_mfma_f32_16x16x32_f16 (sf_gemm.h, M16)
        f32x4 am[4][2 * NJ], ax[4][2 * NJ];
        sf_same_mainloop16<BN_, KW, WM>(g, smem, am, ax, m0, n0);
        sf_epilogue16<BN_, S_::BM>(g, smem, am, ax, m0, n0, tid, wave / S_::WN, wave % S_::WN, lane);
        return;
    }
    f32x16 accm[2][NJ], accx[2][NJ];
    sf_same_mainloop<BN_, KW, WM>(g, smem, accm, accx, m0, n0);
    sf_epilogue<BN_, S_::BM>(g, smem, accm, accx, m0, n0, tid, wave / S_::WN, wave % S_::WN, lane & 31, lane >> 5);
}

// LayerNorm-fused 256-channel linear on 160-row tiles (ln160_mainloop, sf_gemm.h).  Epilogue: the 160 x 256 fp32 tile does not
// fit the LDS at once, so it is staged and finished in two passes of 96 and 64 rows with the same per-row code as every
// other kernel (sf_epilogue_rows): a row's arithmetic does not depend on the pass it is in.
__global__ __launch_bounds__(512, 1) void linear_ln160_kernel(ConvArgs g) {
    using T = V160;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kgrp = lane >> 5;
    const unsigned tile_id = xcd_band_tile(blockIdx.x, gridDim.x >> 3);
    if (tile_id >= g.ntiles) return;
    const int64_t m0 = (int64_t)tile_id * T::BM;
    f32x16 accm[5], accx[5];
    ln160_mainloop(g, smem, accm, accx, m0);
    constexpr int LD = T::BN + 4;
    float* tile = reinterpret_cast<float*>(smem);
    auto stage = [&](int i, int r0) __attribute__((always_inline)) {        // accumulator block i -> tile rows r0 ..
#pragma unroll
        for (int r = 0; r < 16; ++r)
            tile[(r0 + mfma32_row(r, kgrp)) * LD + wave * 32 + col] = accm[i][r] + accx[i][r] * (1.f / 2048.f);
    };
    __syncthreads();
    stage(0, 0);
    stage(1, 32);
    stage(2, 64);
    __syncthreads();
    sf_epilogue_rows<T::BN, 96, T::NT>(g, tile, m0, 0, tid);
    __syncthreads();
    stage(3, 0);
    stage(4, 32);
    __syncthreads();
    sf_epilogue_rows<T::BN, 64, T::NT>(g, tile, m0 + 96, 0, tid);
}

// =================================================================================================
// Linear layers / 1x1 convolutions, second schedule: 128 x 128 tile, 256 threads (4 waves as 2 x 2, 64 x 64 each),
// TWO workgroups per CU.  The 512-thread kernel above owns a whole CU, so while it runs its epilogue (LDS staging,
// LayerNorm / split arithmetic, the output burst) nothing feeds the matrix pipe or the load path, and with K = 128..512
// the epilogue is 30-40 % of a tile; the K <= 256 linears of the refinement head are HBM-bound and reach only a third of
// the HBM rate because loads (main loop) and stores (epilogue) never overlap inside a CU.  Here the co-resident
// workgroup's main loop runs under this one's epilogue, tiles are 2-4x finer (less quantisation loss on 256 CUs), and
// the HBM stream of one workgroup overlaps the MFMA phase of the other.
// Ring: A NA_ deep (3: two slabs of the HBM-resident operand in flight), B (weights, L2-resident) 2 deep; per iteration
// ONE barrier: [wait own pieces of slab t] [barrier] [DMA B(t+1), A(t+NA_-1)] [16 fragment reads] [24 MFMAs].
// B is issued before A, so the newest (NA_-2)*4 pieces in the queue are exactly the A slabs that may stay in flight.
// =================================================================================================
template <int NA_>
struct LIN {
    static constexpr int BM = 128, BN = 128, NT = 256;
    static constexpr int PLANE = 128 * 64;                   // [128 rows][32 halves]
    static constexpr int STAGE = 2 * PLANE;                  // hi, lo
    static constexpr int NA = NA_, NB = 2;
    static constexpr int OFF_B = NA * STAGE;
    static constexpr int RING = OFF_B + NB * STAGE;          // 80 KB (NA 3) / 64 KB (NA 2)
    static constexpr int TILE_BYTES = BM * (BN + 4) * 4;
    static constexpr int SMEM = RING > TILE_BYTES ? RING : TILE_BYTES;
    static constexpr int NWAIT = (NA - 2) * 4;               // own pieces allowed in flight when slab t is published
    static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");
};

template <int NA_>
__global__ __launch_bounds__(256, 2) void linear_gemm_sf_kernel(ConvArgs g) {
    using T = LIN<NA_>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kgrp = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntn = (g.Cout + T::BN - 1) / T::BN;
    const unsigned tile_id = xcd_band_tile(blockIdx.x, gridDim.x >> 3);
    if (tile_id >= g.ntiles) return;
    const int64_t m0 = (int64_t)(tile_id / ntn) * T::BM;
    const int n0 = (tile_id % ntn) * T::BN;
    const int nk = g.Kpad / BK;

    const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)g.xh, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)g.xl, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc((void*)g.wh, 0, g.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)g.wl, 0, g.wbytes, 0x00020000);

    // lane -> (row within a 16-row group, 16-byte k-slot); the XOR swizzle sits on the SOURCE address
    const int lrow = lane >> 2;
    const int lslot = (lane & 3) ^ ((lane >> 4) & 3);
    // this wave stages 16-row groups `wave` and `wave + 4` of both operands (hi and lo plane each: 4 + 4 pieces)
    int64_t abase[2];
    bool aok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t pix = m0 + (wave + 4 * q) * 16 + lrow;
        aok[q] = pix < g.M;
        const int64_t pp = aok[q] ? pix : 0;
        const int ox = (int)(pp % g.W);
        const int64_t t = pp / g.W;
        const int oy = (int)(t % g.H);
        abase[q] = (t / g.H) * g.sxn + (int64_t)oy * g.sxh + (int64_t)ox * g.ldx + lslot * 8;
    }
    const unsigned bbase = (unsigned)(((int64_t)(n0 + lrow) * g.Kpad + lslot * 8) * 2);
    unsigned offA[2], offB[2];
    auto addrA = [&](int t) __attribute__((always_inline)) {
        const bool in = t < nk && t * BK + lslot * 8 < g.Cin;
#pragma unroll
        for (int q = 0; q < 2; ++q) offA[q] = (aok[q] && in) ? (unsigned)((abase[q] + t * BK) * 2) : g.xbytes;
    };
    auto addrB = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            offB[q] = t < nk ? bbase + (unsigned)(wave + 4 * q) * 16u * (unsigned)g.Kpad * 2u + (unsigned)(t * BK * 2) : g.wbytes;
    };
#define LDMA_A(stage)                                                                                                  \
    do {                                                                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                              \
            char* d_ = smem + (stage) * T::STAGE + (wave + 4 * q_) * 1024;                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_void*)d_, 16, offA[q_], 0, 0, 0);                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (lds_void*)(d_ + T::PLANE), 16, offA[q_], 0, 0, 0);           \
        }                                                                                                               \
    } while (0)
#define LDMA_B(stage)                                                                                                  \
    do {                                                                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                              \
            char* d_ = smem + T::OFF_B + (stage) * T::STAGE + (wave + 4 * q_) * 1024;                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, (lds_void*)d_, 16, offB[q_], 0, 0, 0);                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, (lds_void*)(d_ + T::PLANE), 16, offB[q_], 0, 0, 0);           \
        }                                                                                                               \
    } while (0)

    f32x16 accm[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            accm[i][j] = f32x16{0};
            accx[i][j] = f32x16{0};
        }

    // prologue: B(0), A(0) .. A(NA-2)
    addrB(0);
    LDMA_B(0);
#pragma unroll
    for (int t = 0; t < T::NA - 1; ++t) {
        addrA(t);
        LDMA_A(t);
    }
    int ast = 0, bst = 0;                                    // stages of A(t), B(t)
    for (int t = 0; t < nk; ++t) {
        wait_vmcnt<T::NWAIT>();                              // own pieces of B(t), A(t) have landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                        // slab t published; every wave is done with slab t-1
        __builtin_amdgcn_sched_barrier(0);
        const int adm = ast == 0 ? T::NA - 1 : ast - 1;      // stage of A(t-1) = where A(t+NA-1) goes
        addrB(t + 1);
        LDMA_B(bst ^ 1);
        addrA(t + T::NA - 1);
        LDMA_A(adm);
        const char* sa = smem + ast * T::STAGE;
        const char* sb = smem + T::OFF_B + bst * T::STAGE;
        // fragments of both k-steps live in registers (64 VGPRs): k-step 1 is read under k-step 0's MFMAs; the other
        // workgroup's wave on this SIMD covers the head of the segment
        half8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
        auto read_ks = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = tile_off(wr * 64 + i * 32 + col, ks * 2 + kgrp);
                ah[ks][i] = *reinterpret_cast<const half8*>(sa + off);
                al[ks][i] = *reinterpret_cast<const half8*>(sa + T::PLANE + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = tile_off(wc * 64 + j * 32 + col, ks * 2 + kgrp);
                bh[ks][j] = *reinterpret_cast<const half8*>(sb + off);
                bl[ks][j] = *reinterpret_cast<const half8*>(sb + T::PLANE + off);
            }
        };
        auto mma_ks = [&](int ks) __attribute__((always_inline)) {      // consecutive MFMAs never share an accumulator
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bh[ks][j], accm[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bl[ks][j], accx[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], bh[ks][j], accx[i][j], 0, 0, 0);
        };
        read_ks(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        read_ks(1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mma_ks(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // k-step 1 fragments; also: this wave's reads of slab t are complete (WAR for t+1's DMA)
        __builtin_amdgcn_sched_barrier(0);
        mma_ks(1);
        __builtin_amdgcn_s_setprio(0);
        ast = ast == T::NA - 1 ? 0 : ast + 1;
        bst ^= 1;
    }
    wait_vmcnt<0>();                                         // the zero-fill tail pieces, before LDS is reused
#undef LDMA_A
#undef LDMA_B
    sf_epilogue<T::BN, T::BM, T::NT>(g, smem, accm, accx, m0, n0, tid, wr, wc, col, kgrp);
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

// max pooling on split activations: reconstruct hi + lo/2048, take the max, split again.  One thread = 8 channels of one
// output COLUMN of a strip of MP_R output rows: the 2*MP_R + 1 input rows are streamed once (row maxima over the three
// kx taps, then the running 3-row window), 6 loads per input row instead of 18 per output; workgroups are dealt to the
// XCDs in contiguous bands so the columns a strip shares with its neighbours meet in one L2.  max is exact and order-free:
// the result is bit-identical to the one-thread-per-output form.
constexpr int MP_R = 9;
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_sf_kernel(const _Float16* __restrict__ xh,
                                                                   const _Float16* __restrict__ xl,
                                                                   _Float16* __restrict__ yh, _Float16* __restrict__ yl,
                                                                   int H, int W, int C, int Ho, int Wo, int nstrips,
                                                                   int64_t total, int64_t nblk) {
    const int64_t per_xcd = (nblk + 7) >> 3;
    const int64_t blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t e = blk * 256 + threadIdx.x;
    if (blk >= nblk || e >= total) return;
    const int c8 = (int)(e % (C / 8));
    int64_t t = e / (C / 8);
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy0 = (int)(t % nstrips) * MP_R;
    const int64_t n = t / nstrips;
    const int noy = min(MP_R, Ho - oy0);
    float prev[8], cur[8];                       // row maxima of input rows iy - 2 and iy - 1
#pragma unroll
    for (int q = 0; q < 8; ++q) prev[q] = cur[q] = -INFINITY;
    const int iy_begin = oy0 * 2 - 1, iy_end = (oy0 + noy - 1) * 2 + 1;      // inclusive
    for (int iy = iy_begin; iy <= iy_end; ++iy) {
        float rm[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) rm[q] = -INFINITY;
        if (iy >= 0 && iy < H) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                const int64_t o = ((n * H + iy) * W + ix) * C + c8 * 8;
                const half8 h = *reinterpret_cast<const half8*>(xh + o);
                const half8 l = *reinterpret_cast<const half8*>(xl + o);
#pragma unroll
                for (int q = 0; q < 8; ++q) rm[q] = fmaxf(rm[q], (float)h[q] + (float)l[q] * (1.f / 2048.f));
            }
        }
        const int k = iy - iy_begin;             // rows k = 2, 4, ... close an output row: window (iy - 2, iy - 1, iy)
        if (k >= 2 && (k & 1) == 0) {
            half8 oh, ol;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                _Float16 a, b;
                split1(fmaxf(fmaxf(prev[q], cur[q]), rm[q]), a, b);
                oh[q] = a;
                ol[q] = b;
            }
            const int64_t o = ((n * Ho + oy0 + (k >> 1) - 1) * Wo + ox) * C + c8 * 8;
            *reinterpret_cast<half8*>(yh + o) = oh;
            *reinterpret_cast<half8*>(yl + o) = ol;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            prev[q] = cur[q];
            cur[q] = rm[q];
        }
    }
}

template <int BN_>
void launch_v2(const ConvArgs& g, hipStream_t stream) {
    using T = V2<BN_>;
    static dfsfm::SmemAttr smem_attr;
    smem_attr.ensure(reinterpret_cast<const void*>(&conv_gemm_sf_kernel<BN_>), T::SMEM);
    ConvArgs a = g;
    a.ntiles = (unsigned)((g.M + BM2 - 1) / BM2) * (unsigned)((g.Cout + BN_ - 1) / BN_);
    const dim3 grid((a.ntiles + 7) / 8 * 8);                 // 8 XCD bands (xcd_band_tile); surplus WGs exit
    hipLaunchKernelGGL(conv_gemm_sf_kernel<BN_>, grid, dim3(512), T::SMEM, stream, a);
}

template <int BN_, int KW, int WM = 4>
void launch_same(const ConvArgs& g, hipStream_t stream) {
    using S_ = VS<BN_, KW, WM>;
    static dfsfm::SmemAttr smem_attr;
    smem_attr.ensure(reinterpret_cast<const void*>(&conv_gemm_sf_same_kernel<BN_, KW, WM>), S_::SMEM);
    ConvArgs a = g;
    a.ntiles = (unsigned)((g.M + S_::BM - 1) / S_::BM) * (unsigned)((g.Cout + BN_ - 1) / BN_);
    const dim3 grid((a.ntiles + 7) / 8 * 8);                 // 8 XCD bands (xcd_band_tile); surplus WGs exit
    hipLaunchKernelGGL((conv_gemm_sf_same_kernel<BN_, KW, WM>), grid, dim3(512), S_::SMEM, stream, a);
}

void launch_ln160(const ConvArgs& g, hipStream_t stream) {
    using T = V160;
    static dfsfm::SmemAttr smem_attr;
    smem_attr.ensure(reinterpret_cast<const void*>(&linear_ln160_kernel), T::SMEM);
    ConvArgs a = g;
    a.ntiles = (unsigned)((g.M + T::BM - 1) / T::BM);
    const dim3 grid((a.ntiles + 7) / 8 * 8);                 // 8 XCD bands (xcd_band_tile); surplus WGs exit
    hipLaunchKernelGGL(linear_ln160_kernel, grid, dim3(T::NT), T::SMEM, stream, a);
}

// Tile height of the LayerNorm-fused 256-channel linear: 160 rows when that saves rounds on this device's CUs (a 160-row tile
// costs 1.25 x a 128-row one); DFSFM_LN160 = 0 / 1 forces one of the two (A/B switch).
bool use_ln160(int64_t M) {
    const char* e = getenv("DFSFM_LN160");                    // read per call: the tests compare the two schedules in one process
    if (e) return atoi(e) != 0;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    const int64_t r128 = ((M + 127) / 128 + cus - 1) / cus, r160 = ((M + 159) / 160 + cus - 1) / cus;
    return 5 * r160 < 4 * r128;
}

template <int NA_>
void launch_lin(const ConvArgs& g, hipStream_t stream) {
    using T = LIN<NA_>;
    static dfsfm::SmemAttr smem_attr;
    smem_attr.ensure(reinterpret_cast<const void*>(&linear_gemm_sf_kernel<NA_>), T::SMEM);
    ConvArgs a = g;
    a.ntiles = (unsigned)((g.M + T::BM - 1) / T::BM) * (unsigned)((g.Cout + T::BN - 1) / T::BN);
    const dim3 grid((a.ntiles + 7) / 8 * 8);                 // 8 XCD bands (xcd_band_tile); surplus WGs exit
    hipLaunchKernelGGL(linear_gemm_sf_kernel<NA_>, grid, dim3(T::NT), T::SMEM, stream, a);
}

}  // namespace

extern "C" int dfsfm_conv2d_nhwc_f32(const float* x, const void* x_hi, const void* x_lo, int64_t sxn, int64_t sxh,
                                     int64_t ldx, int Nimg, int H, int W, int Cin, const void* w_hi,
                                     const void* w_lo, int Cout, int Kpad, int kh, int kw, int stride, int pad,
                                     const float* bias, const float* residual, const void* res_hi,
                                     const void* res_lo, int64_t ldr, int relu, float* out, int64_t ldo,
                                     void* out_hi, void* out_lo, int64_t ldo_s, int Cout_s, int tap_padded,
                                     const float* ln_gamma, const float* ln_beta, float ln_eps, void* stream_) {
    const bool split_in = x_hi != nullptr;
    if ((ln_gamma == nullptr) != (ln_beta == nullptr)) return DFSFM_E_BADARG;
    // fused LayerNorm: rows must sit in ONE N tile of the 1x1 schedule -> linear layers with Cout = 64, 128 or 256
    if (relu < 0 || relu > 2) return DFSFM_E_BADARG;
    if (ln_gamma && !(split_in && kh == 1 && kw == 1 && stride == 1 && pad == 0 && !relu && !tap_padded &&
                      (Cout == 64 || Cout == 128 || Cout == 256) && ln_eps > 0.f))
        return DFSFM_E_UNSUPPORTED;
    if (tap_padded && !split_in) return DFSFM_E_UNSUPPORTED;
    if ((!x && !split_in) || (x && split_in) || (split_in && !x_lo) || !w_hi || !w_lo) return DFSFM_E_BADARG;
    if (!out && !out_hi) return DFSFM_E_BADARG;
    if ((out_hi == nullptr) != (out_lo == nullptr) || (res_hi == nullptr) != (res_lo == nullptr)) return DFSFM_E_BADARG;
    if (residual && res_hi) return DFSFM_E_BADARG;
    if (Nimg <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0)
        return DFSFM_E_BADARG;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return DFSFM_E_BADARG;
    const int K = tap_padded ? Kpad : kh * kw * Cin;
    if (Kpad < K || Kpad % BK != 0 || ldx < Cin || (out && ldo < Cout) || ((residual || res_hi) && ldr < Cout))
        return DFSFM_E_BADARG;
    if (out_hi && (Cout_s < Cout || Cout_s % 8 != 0 || ldo_s < Cout_s || Cout_s > (Cout + 127) / 128 * 128))
        return DFSFM_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(w_hi) & 15) || (reinterpret_cast<uintptr_t>(w_lo) & 15)) return DFSFM_E_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ConvArgs g{};
    g.x = x; g.xh = static_cast<const _Float16*>(x_hi); g.xl = static_cast<const _Float16*>(x_lo);
    g.wh = static_cast<const _Float16*>(w_hi); g.wl = static_cast<const _Float16*>(w_lo);
    g.bias = bias; g.res = residual; g.resh = static_cast<const _Float16*>(res_hi);
    g.resl = static_cast<const _Float16*>(res_lo);
    g.out = out; g.outh = static_cast<_Float16*>(out_hi); g.outl = static_cast<_Float16*>(out_lo);
    g.ldx = ldx; g.sxh = sxh; g.sxn = sxn; g.ldr = ldr; g.ldo = ldo; g.ldo_s = ldo_s;
    g.M = (int64_t)Nimg * Ho * Wo; g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.Cout = Cout;
    g.Cout_s = out_hi ? Cout_s : 0;
    g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad; g.K = K; g.Kpad = Kpad; g.relu = relu;
    g.lng = ln_gamma; g.lnb = ln_beta; g.ln_eps = ln_eps;

    if (split_in) {
        // v2: LDS-DMA pipeline.  Needs 8-channel granularity and < 4 GiB planes (32-bit buffer offsets).
        const int64_t span = ((int64_t)(Nimg - 1) * sxn + (int64_t)(H - 1) * sxh + (int64_t)(W - 1) * ldx + Cin) * 2;
        const int npad = (Cout + 127) / 128 * 128;
        const int64_t wspan = (int64_t)npad * Kpad * 2;
        if (Cin % 8 || ldx % 8 || sxh % 8 || sxn % 8 || span >= (int64_t)0xFFFFFFF0 || wspan >= (int64_t)0xFFFFFFF0)
            return DFSFM_E_UNSUPPORTED;
        if ((reinterpret_cast<uintptr_t>(x_hi) & 15) || (reinterpret_cast<uintptr_t>(x_lo) & 15)) return DFSFM_E_UNSUPPORTED;
        if (out && ((ldo & 3) || (reinterpret_cast<uintptr_t>(out) & 15))) return DFSFM_E_UNSUPPORTED;
        if (residual && ((ldr & 3) || (reinterpret_cast<uintptr_t>(residual) & 15))) return DFSFM_E_UNSUPPORTED;
        if (res_hi && ((ldr & 7) || (reinterpret_cast<uintptr_t>(res_hi) & 15) || (reinterpret_cast<uintptr_t>(res_lo) & 15)))
            return DFSFM_E_UNSUPPORTED;
        if (out_hi && ((ldo_s & 7) || (reinterpret_cast<uintptr_t>(out_hi) & 15) || (reinterpret_cast<uintptr_t>(out_lo) & 15)))
            return DFSFM_E_UNSUPPORTED;
        g.xbytes = (unsigned)span;
        g.wbytes = (unsigned)wspan;
        if (tap_padded) {      // "same" conv with tap-padded weights: activation tile reused across the kx taps
            if (kh != kw || (kw != 3 && kw != 5) || stride != 1 || pad != kw / 2 || Kpad % (kw * kw * BK) != 0 ||
                Kpad / (kw * kw) < Cin)
                return DFSFM_E_UNSUPPORTED;
            if (kw == 3) { if (Cout <= 64) launch_same<64, 3, 8>(g, stream); else launch_same<128, 3>(g, stream); }
            else         { if (Cout <= 64) launch_same<64, 5>(g, stream); else launch_same<128, 5>(g, stream); }
            return dfsfm::check_launch("dfsfm_conv2d_nhwc_f32(same)");
        }
        if (ln_gamma && Cout == 256) {     // a 256-channel row in ONE workgroup: the 128 x 256 tile (waves 2 x 4), or 160 x 256
            if (use_ln160(g.M)) {
                launch_ln160(g, stream);
                return dfsfm::check_launch("dfsfm_conv2d_nhwc_f32(1x1, 160x256 tile)");
            }
            launch_same<256, 1, 2>(g, stream);
            return dfsfm::check_launch("dfsfm_conv2d_nhwc_f32(1x1, 128x256 tile)");
        }
        if (kh == 1 && kw == 1 && stride == 1 && pad == 0) {    // 1x1 / linear: the same schedule with one tap
            if (Cout > 64) {       // 128 x 128 tiles, two workgroups per CU (a fused LayerNorm here means Cout == 128: one N tile)
                launch_lin<3>(g, stream);
                return dfsfm::check_launch("dfsfm_conv2d_nhwc_f32(linear, 128x128 tile)");
            }
            launch_same<64, 1>(g, stream);
            return dfsfm::check_launch("dfsfm_conv2d_nhwc_f32(1x1)");
        }
        if (Cout <= 64) launch_v2<64>(g, stream);
        else launch_v2<128>(g, stream);
        return dfsfm::check_launch("dfsfm_conv2d_nhwc_f32(v2)");
    }

    const bool vec = (Cin % 4 == 0) && (ldx % 4 == 0) && (sxh % 4 == 0) && (sxn % 4 == 0) &&
                     !(reinterpret_cast<uintptr_t>(x) & 15);
    const dim3 grid((unsigned)((g.M + BM - 1) / BM), (unsigned)((Cout + BN - 1) / BN)), blk(256);
    static dfsfm::SmemAttr smem_attr[2];
    // operand ring (+ the k-table of the scalar-gather path) during the main loop, the epilogue's staging tile afterwards
    constexpr int smem_vec = SMEM_BYTES > V1_EPI_BYTES ? SMEM_BYTES : V1_EPI_BYTES;
    constexpr int smem_lut = SMEM_BYTES + LUT_MAX * 4 > V1_EPI_BYTES ? SMEM_BYTES + LUT_MAX * 4 : V1_EPI_BYTES;
    if (vec) {
        smem_attr[0].ensure(reinterpret_cast<const void*>(&conv_gemm_kernel<true>), smem_vec);
        hipLaunchKernelGGL(conv_gemm_kernel<true>, grid, blk, smem_vec, stream, g);
    } else {
        smem_attr[1].ensure(reinterpret_cast<const void*>(&conv_gemm_kernel<false>), smem_lut);
        hipLaunchKernelGGL(conv_gemm_kernel<false>, grid, blk, smem_lut, stream, g);
    }
    return dfsfm::check_launch("dfsfm_conv2d_nhwc_f32");
}

extern "C" int dfsfm_maxpool3x3s2_nhwc_f32(const float* x, const void* x_hi, const void* x_lo, int Nimg, int H, int W,
                                           int C, float* out, void* out_hi, void* out_lo, void* stream_) {
    if (Nimg <= 0 || H <= 0 || W <= 0 || C <= 0) return DFSFM_E_BADARG;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (x_hi) {
        if (!x_lo || !out_hi || !out_lo) return DFSFM_E_BADARG;
        if (C % 8 != 0) return DFSFM_E_UNSUPPORTED;
        const int nstrips = (Ho + MP_R - 1) / MP_R;
        const int64_t total = (int64_t)Nimg * nstrips * Wo * (C / 8), nblk = (total + 255) / 256;
        if (((nblk + 7) >> 3) * 8 > 0x7fffffff) return DFSFM_E_UNSUPPORTED;
        hipLaunchKernelGGL(maxpool3x3s2_nhwc_sf_kernel, dim3((unsigned)(((nblk + 7) >> 3) * 8)), dim3(256), 0, stream,
                           static_cast<const _Float16*>(x_hi), static_cast<const _Float16*>(x_lo),
                           static_cast<_Float16*>(out_hi), static_cast<_Float16*>(out_lo), H, W, C, Ho, Wo, nstrips, total, nblk);
        return dfsfm::check_launch("dfsfm_maxpool3x3s2_nhwc_f32(split)");
    }
    if (!x || !out) return DFSFM_E_BADARG;
    if (C % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return DFSFM_E_UNSUPPORTED;
    const int64_t total4 = (int64_t)Nimg * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, x, out,
                       H, W, C, Ho, Wo, total4);
    return dfsfm::check_launch("dfsfm_maxpool3x3s2_nhwc_f32");
}
