// K1 -- linear attention  out = phi(Q) (phi(K)^T V) Z   for gfx950 (MI355X).
//
// Replaces LinearAttention.forward of the reference
//   third_party/LoFTR/src/loftr/loftr_module/linear_attention.py:20-47     (coarse: H=8, D=32)
//   src/MultiviewMatcher/matcher_module/linear_attention.py:28-60          (refine: H=8, D=16)
//
// HBM-bound op (AI ~ 8 flop/B): q,k,v are read once, out written once, nothing else of size
// O(L) touches memory.  Three launches:
//   1. kv_partial : every workgroup reduces a chunk of S rows into per-head  KV = K^T (v/S)
//                   (DxD) and Ksum (D) with fp32 MFMA.  H=8 fast path (`*_staged`): whole rows are
//                   loaded with 16-byte coalesced loads, phi/mask/(1/S) applied once per element,
//                   staged through LDS and read back in MFMA fragment order; the next 32-row
//                   block is prefetched into registers.  Generic path: fragment-order global loads.
//   2. kv_finalize: sums the chunk partials, one thread per element, fixed order (deterministic;
//                   skipped when one chunk covers S, the refinement case).
//   3. apply      : out^T = KV^T Q^T per head with MFMA, Z = 1/(Q.Ksum+eps) from the same
//                   registers (one cross-half shuffle).  D=32/H=8: q rows and results go through
//                   one LDS tile so loads and stores are whole 1-KB rows; KV fragments stay in
//                   registers across up to 4 row blocks.
// MFMA k-index permutation: the hardware pairs lane halves (32x32x2: 2 halves, 16x16x4: 4
// quarter-groups) as the k index; we give each group its own k-range on BOTH operands, which
// keeps every lane's accesses contiguous.  Summation order is fixed -> results are run-to-run
// deterministic.
#include "common.h"
#include "enc_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

using namespace dfsfm;

template <int D, typename Acc>
__device__ __forceinline__ Acc mfma_step(float a, float b, Acc acc) {
    if constexpr (D == 32) return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}

__device__ __forceinline__ float mask_at(const uint8_t* m, int group, int64_t base, int idx) {
    return m ? (float)m[base + idx / group] : 1.f;
}

// ---------------------------------------------------------------------------------------------
// D = 32  (v_mfma_f32_32x32x2_f32)
// ---------------------------------------------------------------------------------------------
constexpr int KV32 = 32 * 32 + 32;   // floats per (n,h): KV[d][v] then Ksum[d]

__global__ __launch_bounds__(256) void la_kv_partial_d32(
    const float* __restrict__ k, const float* __restrict__ v, const uint8_t* __restrict__ kv_mask,
    int kv_group, float* __restrict__ part, int S, int H, int ldk, int ldv, int rows_per_chunk,
    int nchunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int s_begin = chunk * rows_per_chunk;
    const int s_end = min(S, s_begin + rows_per_chunk);
    const float Sf = (float)S;
    const int64_t mbase = kv_mask ? (int64_t)n * ((S + kv_group - 1) / kv_group) : 0;
    const float* kn = k + (int64_t)n * S * ldk;
    const float* vn = v + (int64_t)n * S * ldv;

    for (int h = wave; h < H; h += 4) {
        f32x16 acc = {0};
        float ksum = 0.f;
        for (int s0 = s_begin; s0 < s_end; s0 += 32) {
            float kk[16], vv[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int s = s0 + half * 16 + t;
                float kx = 0.f, vx = 0.f;
                if (s < s_end) {
                    const float m = mask_at(kv_mask, kv_group, mbase, s);
                    kx = elu_plus_one(kn[(int64_t)s * ldk + h * 32 + col]) * m;
                    vx = (vn[(int64_t)s * ldv + h * 32 + col] * m) / Sf;
                }
                kk[t] = kx;
                vv[t] = vx;
                ksum += kx;
            }
#pragma unroll
            for (int t = 0; t < 16; ++t)   // A[i=d][k=s] = K[s][d], B[k=s][j=v] = V[s][v]
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[t], vv[t], acc, 0, 0, 0);
        }
        ksum += __shfl_xor(ksum, 32);
        float* p = part + (((int64_t)n * nchunks + chunk) * H + h) * KV32;
#pragma unroll
        for (int r = 0; r < 16; ++r) p[mfma32_row(r, half) * 32 + col] = acc[r];
        if (half == 0) p[1024 + col] = ksum;
    }
}

__global__ __launch_bounds__(256) void la_apply_d32(
    const float* __restrict__ q, const uint8_t* __restrict__ q_mask, int q_group,
    const float* __restrict__ kvf, float* __restrict__ out, int L, int S, int H, int ldq, int ldo,
    float eps, _Float16* __restrict__ outh, _Float16* __restrict__ outl, int ldos) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int n = blockIdx.y;
    const int l = blockIdx.x * 32 + col;
    const bool valid = l < L;
    const float Sf = (float)S;
    const int64_t mbase = q_mask ? (int64_t)n * ((L + q_group - 1) / q_group) : 0;
    const float qm = valid ? mask_at(q_mask, q_group, mbase, l) : 0.f;
    const float* qrow = q + ((int64_t)n * L + (valid ? l : 0)) * ldq;
    const int64_t orow = ((int64_t)n * L + (valid ? l : 0)) * ldo;
    const int64_t srow = ((int64_t)n * L + (valid ? l : 0)) * ldos;

    for (int h = wave; h < H; h += 4) {
        const float* kv = kvf + ((int64_t)n * H + h) * KV32;
        float a[16], ks[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            a[t] = kv[(half * 16 + t) * 32 + col];     // A[i=v][k=d] = KV[d][v]
            ks[t] = kv[1024 + half * 16 + t];
        }
        float Q[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (valid) x = *reinterpret_cast<const f32x4*>(qrow + h * 32 + half * 16 + j * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) Q[j * 4 + e] = valid ? elu_plus_one(x[e]) * qm : 0.f;
        }
        float z = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) z += Q[t] * ks[t];
        z += __shfl_xor(z, 32);
        const float Z = 1.f / (z + eps);
        f32x16 acc = {0};
#pragma unroll
        for (int t = 0; t < 16; ++t)   // B[k=d][j=l] = Q[l][d]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], Q[t], acc, 0, 0, 0);
        if (valid) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // regs 4g..4g+3 -> v = 8g + 4*half + (0..3)
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (acc[g * 4 + e] * Z) * Sf;
                store4(out, outh, outl, orow + h * 32 + g * 8 + half * 4, srow + h * 32 + g * 8 + half * 4, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// D = 16  (v_mfma_f32_16x16x4_f32)
// ---------------------------------------------------------------------------------------------
constexpr int KV16 = 16 * 16 + 16;

__global__ __launch_bounds__(256) void la_kv_partial_d16(
    const float* __restrict__ k, const float* __restrict__ v, const uint8_t* __restrict__ kv_mask,
    int kv_group, float* __restrict__ part, int S, int H, int ldk, int ldv, int rows_per_chunk,
    int nchunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, grp = lane >> 4;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int s_begin = chunk * rows_per_chunk;
    const int s_end = min(S, s_begin + rows_per_chunk);
    const float Sf = (float)S;
    const int64_t mbase = kv_mask ? (int64_t)n * ((S + kv_group - 1) / kv_group) : 0;
    const float* kn = k + (int64_t)n * S * ldk;
    const float* vn = v + (int64_t)n * S * ldv;

    for (int h = wave; h < H; h += 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float ksum = 0.f;
        for (int s0 = s_begin; s0 < s_end; s0 += 32) {
            float kk[8], vv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int s = s0 + grp * 8 + t;
                float kx = 0.f, vx = 0.f;
                if (s < s_end) {
                    const float m = mask_at(kv_mask, kv_group, mbase, s);
                    kx = elu_plus_one(kn[(int64_t)s * ldk + h * 16 + col]) * m;
                    vx = (vn[(int64_t)s * ldv + h * 16 + col] * m) / Sf;
                }
                kk[t] = kx;
                vv[t] = vx;
                ksum += kx;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[t], vv[t], acc, 0, 0, 0);
        }
        ksum += __shfl_xor(ksum, 16);
        ksum += __shfl_xor(ksum, 32);
        float* p = part + (((int64_t)n * nchunks + chunk) * H + h) * KV16;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[(grp * 4 + r) * 16 + col] = acc[r];   // row d = 4*grp + r
        if (grp == 0) p[256 + col] = ksum;
    }
}

// Each workgroup covers 64 rows x all heads: wave w owns heads w, w+4, ...; 4 sub-blocks of 16 rows.
__global__ __launch_bounds__(256) void la_apply_d16(
    const float* __restrict__ q, const uint8_t* __restrict__ q_mask, int q_group,
    const float* __restrict__ kvf, float* __restrict__ out, int L, int S, int H, int ldq, int ldo,
    float eps, _Float16* __restrict__ outh, _Float16* __restrict__ outl, int ldos) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, grp = lane >> 4;
    const int n = blockIdx.y;
    const float Sf = (float)S;
    const int64_t mbase = q_mask ? (int64_t)n * ((L + q_group - 1) / q_group) : 0;

    for (int h = wave; h < H; h += 4) {
        const float* kv = kvf + ((int64_t)n * H + h) * KV16;
        float a[4], ks[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = kv[(grp * 4 + t) * 16 + col];       // A[i=v][k=d] = KV[d][v]
            ks[t] = kv[256 + grp * 4 + t];
        }
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const int l = blockIdx.x * 64 + sb * 16 + col;
            const bool valid = l < L;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            float qm = 0.f;
            if (valid) {
                x = *reinterpret_cast<const f32x4*>(q + ((int64_t)n * L + l) * ldq + h * 16 + grp * 4);
                qm = mask_at(q_mask, q_group, mbase, l);
            }
            float Q[4];
            float z = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Q[e] = valid ? elu_plus_one(x[e]) * qm : 0.f;
                z += Q[e] * ks[e];
            }
            z += __shfl_xor(z, 16);
            z += __shfl_xor(z, 32);
            const float Z = 1.f / (z + eps);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], Q[t], acc, 0, 0, 0);
            if (valid) {   // C: col = lane&15 = row l, row v = 4*grp + reg
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (acc[e] * Z) * Sf;
                store4(out, outh, outl, ((int64_t)n * L + l) * ldo + h * 16 + grp * 4,
                       ((int64_t)n * L + l) * ldos + h * 16 + grp * 4, o);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// H = 8 fast path: whole rows (all heads) are staged through LDS with 16-byte coalesced loads,
// phi / mask / (1/S) applied once per element on the way in, MFMA fragments read back with
// conflict-free ds_read_b32.  Next 32-row block is prefetched into registers during the MFMAs.
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void la_kv_partial_staged(
    const float* __restrict__ k, const float* __restrict__ v, const uint8_t* __restrict__ kv_mask,
    int kv_group, float* __restrict__ part, int S, int ldk, int ldv, int rows_per_chunk, int nchunks) {
    constexpr int H = 8, C = H * D;
    constexpr int LD = (D == 16) ? C + 16 : C;          // D=16: rows t*4+g differ by 1 -> +16 banks
    constexpr int NV = (32 * C / 4) / 256;              // float4 per thread per matrix per block
    constexpr int KVSZ = D * D + D;
    __shared__ __attribute__((aligned(16))) float sK[32 * LD];
    __shared__ __attribute__((aligned(16))) float sV[32 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int s_begin = chunk * rows_per_chunk;
    const int s_end = min(S, s_begin + rows_per_chunk);
    const float Sf = (float)S;
    const int64_t mbase = kv_mask ? (int64_t)n * ((S + kv_group - 1) / kv_group) : 0;
    const float* kn = k + (int64_t)n * S * ldk;
    const float* vn = v + (int64_t)n * S * ldv;

    f32x4 rk[NV], rv[NV];
    auto gload = [&](int s0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + 256 * j, r = idx / (C / 4), c4 = idx % (C / 4);
            const int srow = s0 + r;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (srow < s_end) {
                a = *reinterpret_cast<const f32x4*>(kn + (int64_t)srow * ldk + c4 * 4);
                b = *reinterpret_cast<const f32x4*>(vn + (int64_t)srow * ldv + c4 * 4);
                const float m = mask_at(kv_mask, kv_group, mbase, srow);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = phi_fast(a[e]) * m;
                    b[e] = (b[e] * m) / Sf;
                }
            }
            rk[j] = a;
            rv[j] = b;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + 256 * j, r = idx / (C / 4), c4 = idx % (C / 4);
            *reinterpret_cast<f32x4*>(sK + r * LD + c4 * 4) = rk[j];
            *reinterpret_cast<f32x4*>(sV + r * LD + c4 * 4) = rv[j];
        }
    };

    using acc_t = typename std::conditional<D == 32, f32x16, f32x4>::type;
    acc_t acc[2];
    float ksum[2] = {0.f, 0.f};
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) acc[hh] = acc_t{0};

    gload(s_begin);
    for (int s0 = s_begin; s0 < s_end; s0 += 32) {
        __syncthreads();               // previous block's fragment reads are done
        lstore();
        __syncthreads();
        if (s0 + 32 < s_end) gload(s0 + 32);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = wave + 4 * hh;
            if (D == 32) {
                const int col = lane & 31, half = lane >> 5;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float a = sK[(half * 16 + t) * LD + h * 32 + col];
                    const float b = sV[(half * 16 + t) * LD + h * 32 + col];
                    ksum[hh] += a;
                    acc[hh] = mfma_step<D>(a, b, acc[hh]);
                }
            } else {
                const int col = lane & 15, grp = lane >> 4;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float a = sK[(t * 4 + grp) * LD + h * 16 + col];
                    const float b = sV[(t * 4 + grp) * LD + h * 16 + col];
                    ksum[hh] += a;
                    acc[hh] = mfma_step<D>(a, b, acc[hh]);
                }
            }
        }
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int h = wave + 4 * hh;
        float* p = part + (((int64_t)n * nchunks + chunk) * H + h) * KVSZ;
        float ks = ksum[hh];
        if (D == 32) {
            const int col = lane & 31, half = lane >> 5;
            ks += __shfl_xor(ks, 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) p[mfma32_row(r, half) * 32 + col] = acc[hh][r];
            if (half == 0) p[1024 + col] = ks;
        } else {
            const int col = lane & 15, grp = lane >> 4;
            ks += __shfl_xor(ks, 16);
            ks += __shfl_xor(ks, 32);
#pragma unroll
            for (int r = 0; r < 4; ++r) p[(grp * 4 + r) * 16 + col] = acc[hh][r];
            if (grp == 0) p[256 + col] = ks;
        }
    }
}

// out for H=8, D=32: a workgroup walks up to 4 blocks of 32 rows; wave w keeps KV^T / Ksum of
// heads w and w+4 in registers; q rows are staged through LDS (coalesced 1-KB row loads),
// results go back through the same LDS tile so that stores are whole rows too.
__global__ __launch_bounds__(256, 2) void la_apply_staged_d32(
    const float* __restrict__ q, const uint8_t* __restrict__ q_mask, int q_group,
    const float* __restrict__ kvf, float* __restrict__ out, int L, int S, int ldq, int ldo, float eps,
    int blocks_per_wg, _Float16* __restrict__ outh, _Float16* __restrict__ outl, int ldos) {
    constexpr int H = 8, C = 256, LD = C + 4;          // +4 floats: conflict-free b128 row reads
    __shared__ __attribute__((aligned(16))) float sQ[32 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int n = blockIdx.y;
    const float Sf = (float)S;
    const int64_t mbase = q_mask ? (int64_t)n * ((L + q_group - 1) / q_group) : 0;
    const float* qn = q + (int64_t)n * L * ldq;
    const int64_t on = (int64_t)n * L * ldo, osn = (int64_t)n * L * ldos;

    float a[2][16], ks[2][16];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const float* kv = kvf + ((int64_t)n * H + wave + 4 * hh) * KV32;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            a[hh][t] = kv[(half * 16 + t) * 32 + col];
            ks[hh][t] = kv[1024 + half * 16 + t];
        }
    }
    f32x4 rq[8];
    auto gload = [&](int l0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, r = idx >> 6, c4 = idx & 63;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (l0 + r < L) {
                x = *reinterpret_cast<const f32x4*>(qn + (int64_t)(l0 + r) * ldq + c4 * 4);
                const float m = mask_at(q_mask, q_group, mbase, l0 + r);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = phi_fast(x[e]) * m;
            }
            rq[j] = x;
        }
    };
    const int l_first = blockIdx.x * blocks_per_wg * 32;
    const int l_last = min(L, l_first + blocks_per_wg * 32);
    gload(l_first);
    for (int l0 = l_first; l0 < l_last; l0 += 32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, r = idx >> 6, c4 = idx & 63;
            *reinterpret_cast<f32x4*>(sQ + r * LD + c4 * 4) = rq[j];
        }
        __syncthreads();
        if (l0 + 32 < l_last) gload(l0 + 32);
        f32x16 acc[2];
        float Z[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = wave + 4 * hh;
            float Q[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(sQ + col * LD + h * 32 + half * 16 + j * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) Q[j * 4 + e] = x[e];
            }
            float z = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) z += Q[t] * ks[hh][t];
            z += __shfl_xor(z, 32);
            Z[hh] = 1.f / (z + eps);
            acc[hh] = f32x16{0};
#pragma unroll
            for (int t = 0; t < 16; ++t)
                acc[hh] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[hh][t], Q[t], acc[hh], 0, 0, 0);
        }
        __syncthreads();               // every wave has its Q fragments; the tile becomes the output stage
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = wave + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (acc[hh][g * 4 + e] * Z[hh]) * Sf;
                *reinterpret_cast<f32x4*>(sQ + col * LD + h * 32 + g * 8 + half * 4) = o;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, r = idx >> 6, c4 = idx & 63;
            if (l0 + r < L)
                store4(out, outh, outl, on + (int64_t)(l0 + r) * ldo + c4 * 4, osn + (int64_t)(l0 + r) * ldos + c4 * 4,
                       *reinterpret_cast<const f32x4*>(sQ + r * LD + c4 * 4));
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Generic head dimension (D % 4 == 0, D <= 64; no masks): plain fp32 FMA kernels for the head sizes the MFMA kernels
// do not cover -- MatchFormer-LA uses D = 24 (dim 192) and D = 64 (dim 512) beside 16 and 32
// (third_party/MatchFormer/model/backbone/match_LA_large.py:64-89).  Same partial layout (KV[d][v] then Ksum[d]) and
// the same arithmetic as the reference: v / S inside the KV sum, (phi(Q) KV) Z S on the way out.
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void la_kv_partial_generic(const float* __restrict__ k, const float* __restrict__ v,
                                                             float* __restrict__ part, int S, int H, int ldk, int ldv,
                                                             int rows_per_chunk, int nchunks) {
    constexpr int EPT = (D * D + 255) / 256;                 // KV elements per thread (consecutive v of one d)
    static_assert(D % EPT == 0 && D % 4 == 0, "thread tiling of the D x D block");
    __shared__ float s_k[32][D], s_v[32][D];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, n = blockIdx.y, h = blockIdx.z;
    const int s_begin = chunk * rows_per_chunk, s_end = min(S, s_begin + rows_per_chunk);
    const float Sf = (float)S;
    const float* kn = k + (int64_t)n * S * ldk + h * D;
    const float* vn = v + (int64_t)n * S * ldv + h * D;
    const int e0 = tid * EPT, d = e0 / D, v0 = e0 % D;
    const bool own = e0 < D * D;
    float acc[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) acc[i] = 0.f;
    float ksum = 0.f;
    for (int s0 = s_begin; s0 < s_end; s0 += 32) {
        for (int e = tid; e < 32 * D; e += 256) {
            const int r = e / D, c = e % D, srow = s0 + r;
            float kx = 0.f, vx = 0.f;
            if (srow < s_end) {
                kx = elu_plus_one(kn[(int64_t)srow * ldk + c]);
                vx = vn[(int64_t)srow * ldv + c] / Sf;
            }
            s_k[r][c] = kx;
            s_v[r][c] = vx;
        }
        __syncthreads();
        if (own) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const float a = s_k[r][d];
#pragma unroll
                for (int i = 0; i < EPT; ++i) acc[i] = fmaf(a, s_v[r][v0 + i], acc[i]);
                if (v0 == 0) ksum += a;
            }
        }
        __syncthreads();
    }
    if (own) {
        float* p = part + (((int64_t)n * nchunks + chunk) * H + h) * (D * D + D);
#pragma unroll
        for (int i = 0; i < EPT; ++i) p[e0 + i] = acc[i];
        if (v0 == 0) p[D * D + d] = ksum;
    }
}

// 64 query rows x one head per workgroup; thread = (row, quarter of the D outputs)
template <int D>
__global__ __launch_bounds__(256) void la_apply_generic(const float* __restrict__ q, const float* __restrict__ kvf,
                                                        float* __restrict__ out, int L, int S, int H, int ldq, int ldo,
                                                        float eps, _Float16* __restrict__ outh, _Float16* __restrict__ outl,
                                                        int ldos) {
    constexpr int VQ = D / 4;                                // outputs per thread
    static_assert(VQ % 2 == 0, "4-float stores");
    __shared__ float s_kv[D * D + D];
    __shared__ float s_q[64][D + 1];
    const int tid = threadIdx.x, n = blockIdx.y, h = blockIdx.z;
    const int l0 = blockIdx.x * 64;
    const float* kv = kvf + ((int64_t)n * H + h) * (D * D + D);
    for (int e = tid; e < D * D + D; e += 256) s_kv[e] = kv[e];
    for (int e = tid; e < 64 * D; e += 256) {
        const int r = e / D, c = e % D, l = l0 + r;
        s_q[r][c] = l < L ? elu_plus_one(q[((int64_t)n * L + l) * ldq + h * D + c]) : 0.f;
    }
    __syncthreads();
    const int r = tid >> 2, part = tid & 3, l = l0 + r;
    if (l >= L) return;
    float z = 0.f;
#pragma unroll 8
    for (int dd = 0; dd < D; ++dd) z = fmaf(s_q[r][dd], s_kv[D * D + dd], z);
    const float Z = 1.f / (z + eps), Sf = (float)S;
    float acc[VQ];
#pragma unroll
    for (int i = 0; i < VQ; ++i) acc[i] = 0.f;
#pragma unroll 4
    for (int dd = 0; dd < D; ++dd) {
        const float a = s_q[r][dd];
#pragma unroll
        for (int i = 0; i < VQ; ++i) acc[i] = fmaf(a, s_kv[dd * D + part * VQ + i], acc[i]);
    }
    const int64_t o32 = ((int64_t)n * L + l) * ldo + h * D + part * VQ;
    const int64_t o16 = ((int64_t)n * L + l) * ldos + h * D + part * VQ;
#pragma unroll
    for (int i = 0; i < VQ; i += 2) {                         // VQ = 6 (D = 24) is not a multiple of 4: 2-float pieces
        const float a = (acc[i] * Z) * Sf, b = (acc[i + 1] * Z) * Sf;
        if (out) { out[o32 + i] = a; out[o32 + i + 1] = b; }
        if (outh) {
            _Float16 ha, la_, hb, lb;
            split_f32(a, ha, la_);
            split_f32(b, hb, lb);
            outh[o16 + i] = ha; outl[o16 + i] = la_;
            outh[o16 + i + 1] = hb; outl[o16 + i + 1] = lb;
        }
    }
}

// Sum the chunk partials: one thread per element, four interleaved partial sums (fixed order ->
// deterministic) so that the chunk loads are independent instead of one serial latency chain.
__global__ __launch_bounds__(256) void la_kv_finalize(const float* __restrict__ part,
                                                      float* __restrict__ kvf, int H, int nchunks,
                                                      int kvsz) {
    const int nh = blockIdx.x;   // n*H + h
    const int n = nh / H, h = nh % H;
    const int e = blockIdx.y * blockDim.x + threadIdx.x;
    if (e >= kvsz) return;
    const float* p = part + ((int64_t)n * nchunks * H + h) * kvsz + e;
    const int64_t stride = (int64_t)H * kvsz;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 4 <= nchunks; c += 4) {
        s0 += p[(c + 0) * stride];
        s1 += p[(c + 1) * stride];
        s2 += p[(c + 2) * stride];
        s3 += p[(c + 3) * stride];
    }
    for (; c < nchunks; ++c) s0 += p[c * stride];
    kvf[(int64_t)nh * kvsz + e] = (s0 + s1) + (s2 + s3);
}

// Rows per kv_partial workgroup: aim for ~512 workgroups (one resident wave of 2 per CU), at least
// 64 rows each.
int chunk_rows(int N, int S) {
    constexpr int wgs = 512;
    int64_t want = ((int64_t)S * N + wgs - 1) / wgs;
    int rows = (int)((want + 31) / 32 * 32);
    if (rows < 64) rows = 64;
    const int smax = (S + 31) / 32 * 32;
    if (rows > smax) rows = smax;
    return rows;
}

}  // namespace

extern "C" size_t dfsfm_linear_attention_workspace(int N, int S, int H, int D) {
    if (N <= 0 || S <= 0 || H <= 0 || (D != 16 && D != 32 && D != 24 && D != 64)) return 0;
    const int rows = chunk_rows(N, S);
    const int nchunks = (S + rows - 1) / rows;
    const size_t kvsz = (size_t)D * D + D;
    size_t fin = (size_t)N * H * kvsz * sizeof(float);
    size_t part = nchunks > 1 ? (size_t)N * nchunks * H * kvsz * sizeof(float) : 0;
    return dfsfm::align_up(fin, 256) + dfsfm::align_up(part, 256);
}

extern "C" int dfsfm_linear_attention_f32(const float* q, const float* k, const float* v,
                                          const uint8_t* q_mask, int q_group,
                                          const uint8_t* kv_mask, int kv_group, float* out, int N,
                                          int L, int S, int H, int D, int ldq, int ldk, int ldv,
                                          int ldo, float eps, void* out_hi, void* out_lo, int ldo_s,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
    if (!q || !k || !v || (!out && !out_hi) || !workspace) return DFSFM_E_BADARG;
    if ((out_hi == nullptr) != (out_lo == nullptr)) return DFSFM_E_BADARG;
    if (out_hi && (ldo_s < H * D || (ldo_s & 3) || (reinterpret_cast<uintptr_t>(out_hi) & 7) ||
                   (reinterpret_cast<uintptr_t>(out_lo) & 7)))
        return DFSFM_E_BADARG;
    _Float16* oh = static_cast<_Float16*>(out_hi);
    _Float16* ol = static_cast<_Float16*>(out_lo);
    if (N <= 0 || L <= 0 || S <= 0 || H <= 0) return DFSFM_E_BADARG;
    if (D != 16 && D != 32 && D != 24 && D != 64) return DFSFM_E_UNSUPPORTED;
    if ((D == 24 || D == 64) && (q_mask || kv_mask)) return DFSFM_E_UNSUPPORTED;   // the generic kernels take no masks
    if (ldq < H * D || ldk < H * D || ldv < H * D || (out && ldo < H * D)) return DFSFM_E_BADARG;
    if ((ldq & 3) || (out && (ldo & 3))) return DFSFM_E_UNSUPPORTED;   // float4 row access
    if ((reinterpret_cast<uintptr_t>(q) & 15) || (out && (reinterpret_cast<uintptr_t>(out) & 15)))
        return DFSFM_E_UNSUPPORTED;
    if (q_mask && (q_group <= 0)) return DFSFM_E_BADARG;
    if (kv_mask && (kv_group <= 0)) return DFSFM_E_BADARG;
    if (N > 65535) return DFSFM_E_UNSUPPORTED;
    if (workspace_bytes < dfsfm_linear_attention_workspace(N, S, H, D)) return DFSFM_E_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);

    const int rows = chunk_rows(N, S);
    const int nchunks = (S + rows - 1) / rows;
    const int kvsz = D * D + D;
    float* kvf = static_cast<float*>(workspace);
    float* part = nchunks > 1
                      ? reinterpret_cast<float*>(static_cast<char*>(workspace) +
                                                 dfsfm::align_up((size_t)N * H * kvsz * sizeof(float), 256))
                      : kvf;
    dim3 gA(nchunks, N), blk(256);
    if (D == 24 || D == 64) {
        const dim3 gG(nchunks, N, H), gP((L + 63) / 64, N, H);
        if (D == 24) hipLaunchKernelGGL(la_kv_partial_generic<24>, gG, blk, 0, stream, k, v, part, S, H, ldk, ldv, rows, nchunks);
        else hipLaunchKernelGGL(la_kv_partial_generic<64>, gG, blk, 0, stream, k, v, part, S, H, ldk, ldv, rows, nchunks);
        if (nchunks > 1)
            hipLaunchKernelGGL(la_kv_finalize, dim3(N * H, (kvsz + 255) / 256), blk, 0, stream, part, kvf, H, nchunks, kvsz);
        if (D == 24) hipLaunchKernelGGL(la_apply_generic<24>, gP, blk, 0, stream, q, kvf, out, L, S, H, ldq, ldo, eps, oh, ol, ldo_s);
        else hipLaunchKernelGGL(la_apply_generic<64>, gP, blk, 0, stream, q, kvf, out, L, S, H, ldq, ldo, eps, oh, ol, ldo_s);
        return dfsfm::check_launch("dfsfm_linear_attention_f32(generic D)");
    }
    const bool k_aligned = !(ldk & 3) && !(ldv & 3) && !(reinterpret_cast<uintptr_t>(k) & 15) &&
                           !(reinterpret_cast<uintptr_t>(v) & 15);
    const bool staged = (H == 8) && k_aligned;          // whole-row LDS staging (coalesced 16-B loads)
    if (D == 32) {
        if (staged)
            hipLaunchKernelGGL(la_kv_partial_staged<32>, gA, blk, 0, stream, k, v, kv_mask, kv_group, part, S, ldk,
                               ldv, rows, nchunks);
        else
            hipLaunchKernelGGL(la_kv_partial_d32, gA, blk, 0, stream, k, v, kv_mask, kv_group, part, S, H, ldk,
                               ldv, rows, nchunks);
    } else {
        if (staged)
            hipLaunchKernelGGL(la_kv_partial_staged<16>, gA, blk, 0, stream, k, v, kv_mask, kv_group, part, S, ldk,
                               ldv, rows, nchunks);
        else
            hipLaunchKernelGGL(la_kv_partial_d16, gA, blk, 0, stream, k, v, kv_mask, kv_group, part, S, H, ldk,
                               ldv, rows, nchunks);
    }
    if (nchunks > 1)
        hipLaunchKernelGGL(la_kv_finalize, dim3(N * H, (kvsz + 255) / 256), blk, 0, stream, part, kvf, H, nchunks,
                           kvsz);
    if (D == 32) {
        if (H == 8) {
            const int nblk = (L + 31) / 32;
            // blocks per workgroup: the whole grid in ONE round of 2 workgroups per CU when that takes <= 8 blocks each (a
            // second, quarter-full round cost as much as the first: 16 x 150 blocks ran as 608 workgroups on 512 slots)
            int bpw = (int)(((int64_t)nblk * N + 511) / 512);
            bpw = bpw < 1 ? 1 : (bpw > 8 ? 4 : bpw);
            hipLaunchKernelGGL(la_apply_staged_d32, dim3((nblk + bpw - 1) / bpw, N), blk, 0, stream, q, q_mask,
                               q_group, kvf, out, L, S, ldq, ldo, eps, bpw, oh, ol, ldo_s);
        } else {
            hipLaunchKernelGGL(la_apply_d32, dim3((L + 31) / 32, N), blk, 0, stream, q, q_mask, q_group, kvf, out,
                               L, S, H, ldq, ldo, eps, oh, ol, ldo_s);
        }
    } else {
        hipLaunchKernelGGL(la_apply_d16, dim3((L + 63) / 64, N), blk, 0, stream, q, q_mask, q_group, kvf, out, L, S,
                           H, ldq, ldo, eps, oh, ol, ldo_s);
    }
    return dfsfm::check_launch("dfsfm_linear_attention_f32");
}

extern "C" size_t dfsfm_encoder256_state_workspace(int N, int S) {
    if (N <= 0 || S <= 0) return 0;
    const int rows = chunk_rows(N, S);
    const int nchunks = (S + rows - 1) / rows;
    return dfsfm::align_up((size_t)N * nchunks * 8 * KV32 * sizeof(float), 256);
}

extern "C" int dfsfm_encoder256_state_f32(const float* k, const float* v, int ldk, int ldv, const uint8_t* kv_mask,
                                          int kv_group, int N, int S, void* kv_image, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
    if (!k || !v || !kv_image || !workspace) return DFSFM_E_BADARG;
    if (N <= 0 || S <= 0 || ldk < 256 || ldv < 256 || (kv_mask && kv_group <= 0)) return DFSFM_E_BADARG;
    if (N > 65535 || (ldk & 3) || (ldv & 3) || (reinterpret_cast<uintptr_t>(k) & 15) || (reinterpret_cast<uintptr_t>(v) & 15) ||
        (reinterpret_cast<uintptr_t>(kv_image) & 15))
        return DFSFM_E_UNSUPPORTED;
    if (workspace_bytes < dfsfm_encoder256_state_workspace(N, S)) return DFSFM_E_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int rows = chunk_rows(N, S);
    const int nchunks = (S + rows - 1) / rows;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(la_kv_partial_staged<32>, dim3(nchunks, N), dim3(256), 0, stream, k, v, kv_mask, kv_group, part, S, ldk,
                       ldv, rows, nchunks);
    dfsfm_enc::enc256_launch_image(part, static_cast<char*>(kv_image), N, nchunks, stream);
    return dfsfm::check_launch("dfsfm_encoder256_state_f32");
}
