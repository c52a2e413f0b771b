// K2 + K1 (apply half) fused for d_model 256, 8 heads of 32 -- the COARSE transformer's LoFTREncoderLayer on gfx950 (MI355X).
//
// Replaces, per layer application of
//   LoFTREncoderLayer.forward   third_party/LoFTR/src/loftr/loftr_module/transformer.py:35-58
//   LinearAttention.forward     third_party/LoFTR/src/loftr/loftr_module/linear_attention.py:31-47 (the query side)
// the launches  q GEMM -> attention apply -> merge + LayerNorm1 -> mlp.0 + ReLU -> mlp.2 + LayerNorm2 + residual  (five
// kernels that round-trip q, the message, [x | norm1(message)] and the 512-wide hidden layer through HBM) by ONE kernel
// that reads a token row once (1 KB as fp16x2 split planes) and writes it once.  The source side (k | v projection,
// phi(K)^T V) stays on the split-plane GEMM + K1's partial-sum kernel; `enc256_image_kernel` (linear_attention.hip) turns
// the per-head KV / Ksum sums into the "apply image" this kernel consumes.
//
// Same construction as encoder_fused.hip (d_model 128), re-tiled so that a 256-channel layer fits one wave's registers:
//  * TOKEN-STATIONARY, TRANSPOSED: out^T[channel][token] = W[channel][k] x^T[k][token] on v_mfma_f32_16x16x32_f16: a wave owns
//    16 tokens (lane = token n = lane & 15, lane group g = lane >> 4); an accumulator block is 16 channels x 16 tokens in
//    4 registers (row 4 g + r), and two consecutive blocks ARE the B operand of the next GEMM's 32-wide k-step after an
//    fp32 -> fp16x2 split in registers: slot (g, j) of k-step s  <->  channel 32 s + 16 (j >> 2) + 4 g + (j & 3).  The host
//    permutes the weight columns accordingly (ops.Encoder256Weights); x fragments are read from the staging tile in the
//    same order.  No activation touches LDS between the input tile and the output tile, no cross-lane traffic except the
//    lane-group reductions (xor 16, xor 32) of Z and the LayerNorm statistics.
//  * fp16x2 split arithmetic (value = hi + lo / 2048): three MFMAs per product, fp32 accumulation, lo * lo dropped.
//  * The four waves of a workgroup (64 tokens) share only the weight stream: 128 slabs of 16 KB per tile (q 16, merge 16,
//    8 x [mlp.0 chunk 8 + mlp.2 chunk 4]) through the 4-deep LDS ring of enc_common.h; one wave per SIMD; the fragment reads of
//    slab s + 1 run under the MFMAs of slab s (register double buffering, slab_step).
//  * With D = 32 a head is exactly one k-step: KV^T of a head is two 16 x 32 A fragments, the message of a head 6 MFMAs.
#include "common.h"
#include "enc_common.h"
#include <cstdlib>

namespace {

using namespace dfsfm;
using namespace dfsfm_enc;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int EC = 256;                  // d_model
constexpr int NB16 = EC / 16;            // 16-channel accumulator blocks
constexpr int NKS = EC / 32;             // 32-wide k-steps over d_model (= heads)
constexpr int STG = 16384;               // per-wave staging tile: 16 tokens x 256 channels, hi + lo planes (8 KB each)
constexpr int PLANE = 8192;
constexpr int NSLAB = 128;               // q 16, merge 16, 8 x (mlp.0 chunk 8 + mlp.2 chunk 4)
constexpr int KVIMG = 32 * 1024 + 1024;  // bytes per sequence: 16 KV^T fragment pairs (hi, lo) + Ksum[256]
constexpr int KVSZ = 32 * 32 + 32;       // floats per (n, chunk, head) of the chunk partials: KV[d][v] then Ksum[d] (K1's layout)
constexpr int SMEM_APPLY = RING + 4 * STG + 4 * EC * 4 + 4 * 2048;   // + LayerNorm gamma / beta + Ksum of a wave's two sequences

__device__ __forceinline__ f32x4 mfma16(const half8 a, const half8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16s(const half8 a, const half8 b, const f32x4 c) {     // the slab stream's MFMAs
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// Two consecutive accumulator blocks (lane (n, g): a[r] = channel 16 (2 s) + 4 g + r, b[r] = channel 16 (2 s + 1) + 4 g + r of
// token n) -> the operand fragment pair of k-step s.  Same saturating split as encoder_fused.hip's to_frags.
__device__ __forceinline__ void to_frag16(const float (&a)[4], const float (&b)[4], half8& h, half8& l) {
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    const half2_t big = {(_Float16)65504.f, (_Float16)65504.f};
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        half2_t ha = {(_Float16)a[r], (_Float16)a[r + 1]};
        ha = __builtin_elementwise_max(__builtin_elementwise_min(ha, big), -big);
        const half2_t la = {(_Float16)((a[r] - (float)ha[0]) * 2048.f), (_Float16)((a[r + 1] - (float)ha[1]) * 2048.f)};
        half2_t hb = {(_Float16)b[r], (_Float16)b[r + 1]};
        hb = __builtin_elementwise_max(__builtin_elementwise_min(hb, big), -big);
        const half2_t lb = {(_Float16)((b[r] - (float)hb[0]) * 2048.f), (_Float16)((b[r + 1] - (float)hb[1]) * 2048.f)};
        h[r] = ha[0]; h[r + 1] = ha[1]; l[r] = la[0]; l[r + 1] = la[1];
        h[4 + r] = hb[0]; h[5 + r] = hb[1]; l[4 + r] = lb[0]; l[5 + r] = lb[1];
    }
}

// byte offset of 16-byte chunk c (0..31) of token row t inside a staging plane (16 rows x 512 B): XOR swizzle of the low four
// chunk bits by the row, so that the 8-byte fragment reads of 16 different rows (same logical chunk) spread over all banks
__device__ __forceinline__ int stg_off(int t, int c) { return t * 512 + ((c ^ (t & 15)) << 4); }

// One slab = KPS k-steps x NB blocks of (hi, lo) A fragments:  acc[B0 + b] += W(b, ks) * B[ks]  with the 3-MFMA split product
// am += W_hi B_hi, ax += W_lo B_hi + W_hi B_lo (an accumulator is reused after NB >= 4 other MFMAs).
//
// SOFTWARE-PIPELINED over the slab stream: the 16 fragments of the slab being multiplied sit in registers (`cur`); between its 24
// MFMAs the NEXT slab's fragments are read from LDS into `nxt`, one read per MFMA, and the ring slot the current slab came from is
// refilled (slab next + NSTG).  A slab is 64 KB of LDS reads per CU (four waves x 16 KB) = 256 LDS cycles for 384 cycles of MFMA
// issue: read-then-multiply per slab serialised the two (first version: 65 us per 64-token tile for 20 us of MFMA issue).  One s_barrier per slab as before; it now means "every wave's pieces of slab next + 1 have landed AND every wave
// has the fragments of slab next in registers" (lgkmcnt(0) before it), so the refill cannot overtake a reader.  Callers alternate
// the two register sets (the slab sequence of a tile is static and even).
#define ENC256_REFILL(ring, gn, piece) (ring).issue_piece((gn), (piece)++)
template <int NB, int KPS, int B0>
__device__ __forceinline__ void slab_step(SlabRing& ring, int lane, f32x4 (&am)[NB16], f32x4 (&ax)[NB16], const half8* bh,
                                          const half8* bl, const half8 (&cur)[16], half8 (&nxt)[16]) {
    static_assert(NB * KPS == 8, "a slab holds 16 fragments");
    wait_vmcnt<(NSTG - 2) * 4>();                   // this wave's pieces of slab next + 1 (next + 2, next + 3 may still fly)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and its fragment reads of slab next are complete
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const char* nslab = ring.ring + ((ring.next + 1) % NSTG) * SLAB + lane * 16;
    const unsigned gn = ring.next + NSTG;           // into the slot slab `next` was read from
    // 24 MFMAs; behind each of the first 16 ONE fragment read of the next slab, behind MFMAs 3, 9, 15, 21 one DMA request of the
    // refill.  Issued as a burst after the barrier, the four waves' 64 KB of ds_read_b128 saturate the LDS for ~256 cycles during
    // which none of them issues an MFMA (stage profile: 760 cycles per slab for 384 of MFMA); one read per MFMA is the rate the
    // LDS sustains (4 waves x 4 cycles per 16-cycle MFMA).  sched_barriers pin the order.
    int m = 0, piece = 0;
    auto after = [&]() __attribute__((always_inline)) {
        if (m < 16) nxt[m] = *reinterpret_cast<const half8*>(nslab + m * 1024);
        if (m % 6 == 3) ENC256_REFILL(ring, gn, piece);
        ++m;
        __builtin_amdgcn_sched_barrier(0);
    };
    (void)nslab;
#pragma unroll
    for (int ks = 0; ks < KPS; ++ks) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            am[B0 + b] = mfma16s(cur[(ks * NB + b) * 2], bh[ks], am[B0 + b]);
            after();
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            ax[B0 + b] = mfma16s(cur[(ks * NB + b) * 2 + 1], bh[ks], ax[B0 + b]);
            after();
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            ax[B0 + b] = mfma16s(cur[(ks * NB + b) * 2], bl[ks], ax[B0 + b]);
            after();
        }
    }
    ring.advance();
}

struct Apply256Args {
    const _Float16 *xh, *xl;     // x rows as split planes, row stride ldx (elements)
    int64_t ldx;
    unsigned xbytes;
    const char* wstream;         // NSLAB slabs
    const char* kvimg;           // [N][KVIMG]
    const uint8_t* qmask;        // [N][qm_per_seq] or null
    int q_group, qm_per_seq;
    const float *g1, *b1, *g2, *b2;
    float eps1, eps2, attn_eps;
    _Float16 *oh, *ol;           // out rows as split planes (or null)
    int64_t ldo;
    float* o32;                  // out rows fp32 (or null)
    int64_t ldo32;
    int64_t M;                   // rows = N * L
    int L, N, S;
    int ntiles;                  // ceil(M / 64)
    float* dbg;                  // [M][256] fp32 dump of one intermediate (tests), or null
    int dbg_stage;               // 1 q, 2 message, 3 norm1(merge), 4 mlp output (before norm2)
};

// accumulator-layout values of block B (lane (n, g): v[r] = channel 16 B + 4 g + r) -> dbg rows.  Force-inlined like everything
// in the fused encoder kernels (a real call from a 512-register kernel corrupted caller state, encoder_fused.hip).
__device__ __forceinline__ void dump16(const Apply256Args& g, const float (&v)[4], int B, int64_t row, bool valid, int grp) {
    if (!valid) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) g.dbg[row * EC + 16 * B + 4 * grp + r] = v[r];
}

__global__ __launch_bounds__(256) void enc256_apply_kernel(Apply256Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 15, grp = lane >> 4;
    char* stg = smem + RING + wave * STG;
    float* s_ln = reinterpret_cast<float*>(smem + RING + 4 * STG);               // [g1 | b1 | g2 | b2][EC]
    if (tid < EC) {
        s_ln[tid] = g.g1[tid];
        s_ln[EC + tid] = g.b1[tid];
        s_ln[2 * EC + tid] = g.g2[tid];
        s_ln[3 * EC + tid] = g.b2[tid];
    }
    __syncthreads();

    SlabRing ring;
    ring.ring = smem;
    ring.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.wstream, 0, NSLAB * SLAB, 0x00020000);
    ring.lane_off = (unsigned)(wave * 4096 + lane * 16);
    ring.wave = wave;
    ring.nslab = NSLAB;
    // pipeline prologue: all four ring slots requested; slab 0 goes to the register set wa
#pragma unroll
    for (int gq = 0; gq < NSTG; ++gq) ring.issue(gq);
    ring.next = 0;
    half8 wa[16], wb[16];                    // fragment registers of the slab being multiplied / being fetched (ping-pong)
    wait_vmcnt<(NSTG - 1) * 4>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int f = 0; f < 16; ++f) wa[f] = *reinterpret_cast<const half8*>(smem + f * 1024 + lane * 16);

    const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)g.xh, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)g.xl, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rkv = __builtin_amdgcn_make_buffer_rsrc((void*)g.kvimg, 0, (unsigned)((int64_t)g.N * KVIMG), 0x00020000);
    const float Sf = (float)g.S;
    char* s_ks = smem + RING + 4 * STG + 4 * EC * 4 + wave * 2048;             // Ksum[256] of the wave's (at most) two sequences

    // a lane id the compiler cannot hoist out of the tile loop (as loop invariants the staging addresses were spilled, and
    // every scratch reload drained the DMA queue: encoder_fused.hip)
    auto fresh_lane = []() __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // x rows of tile t -> this wave's staging (rows of 512 B per plane, 16-byte chunks XOR-swizzled on the SOURCE side: LDS-DMA
    // destinations are lane-linear) + Ksum of the two sequences the 16 rows may belong to
    auto load_x = [&](int t) __attribute__((always_inline)) {
        const int64_t r0 = ((int64_t)t * 4 + wave) * 16;
        const int fl = fresh_lane();
        const int trow = fl >> 5, p = fl & 31;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int seq = min((int)(r0 / g.L) + u, g.N - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rkv, (lds_void*)(s_ks + u * 1024), 16,
                                                     (unsigned)((int64_t)seq * KVIMG + 32768 + fl * 16), 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tt = 2 * i + trow;
            const int64_t rr = r0 + tt;
            const unsigned off = rr < g.M ? (unsigned)((rr * g.ldx + ((p ^ (tt & 15)) << 3)) * 2) : g.xbytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_void*)(stg + i * 1024), 16, off, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (lds_void*)(stg + PLANE + i * 1024), 16, off, 0, 0, 0);
        }
    };
    // dbg_stage 100: wave 0 of every workgroup records s_memtime at the stage boundaries of each of its tiles
    // (dbg[(tile * 16 + k) * 2 ..] = low 24 bits, next 24 bits): the stage profile of DESIGN.md (tools/bench_enc256.py profile)
    const bool stamp = g.dbg && g.dbg_stage == 100 && tid == 0;
#define ENC_STAMP(k)                                                                    \
    if (stamp) {                                                                        \
        const uint64_t t_ = __builtin_amdgcn_s_memtime();                               \
        g.dbg[((int64_t)tile * 16 + (k)) * 2] = (float)(t_ & 0xFFFFFF);                 \
        g.dbg[((int64_t)tile * 16 + (k)) * 2 + 1] = (float)((t_ >> 24) & 0xFFFFFF);     \
    }
    if ((int)blockIdx.x < g.ntiles) load_x(blockIdx.x);
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        ENC_STAMP(0)
        const int64_t row0 = ((int64_t)tile * 4 + wave) * 16;
        const int64_t row = row0 + tok;
        const bool valid = row < g.M;
        // sequence of this lane's token; the wave's 16 rows touch at most two sequences (L >= 16)
        const int n_first = (int)(row0 / g.L);
        const int64_t bound = (int64_t)(n_first + 1) * g.L;
        const int n_tok = min(row >= bound ? n_first + 1 : n_first, g.N - 1);
        const int l_tok = (int)(row - (int64_t)n_tok * g.L);
        float qm = valid ? 1.f : 0.f;
        if (g.qmask && valid) qm = (float)g.qmask[(int64_t)n_tok * g.qm_per_seq + l_tok / g.q_group];
        const bool two = __builtin_amdgcn_readfirstlane((int)(row0 + 15 >= bound && n_first + 1 < g.N)) != 0;

        wait_vmcnt<0>();        // the x tile has landed (this also waits for the slabs in flight: once per tile)
        ENC_STAMP(1)
        // B fragment pair of x for k-step s, from the staging tile (it stays intact until the output overwrites it in place):
        // slots j < 4: channels 32 s + 4 g + j, slots j >= 4: channels 32 s + 16 + 4 g + (j - 4)
        auto xfrag = [&](int s, half8& fh, half8& fl) __attribute__((always_inline)) {
            const int c0 = 4 * s + (grp >> 1), o8 = 8 * (grp & 1);
            const half4 a0 = *reinterpret_cast<const half4*>(stg + stg_off(tok, c0) + o8);
            const half4 a1 = *reinterpret_cast<const half4*>(stg + stg_off(tok, c0 + 2) + o8);
            const half4 b0 = *reinterpret_cast<const half4*>(stg + PLANE + stg_off(tok, c0) + o8);
            const half4 b1 = *reinterpret_cast<const half4*>(stg + PLANE + stg_off(tok, c0 + 2) + o8);
            fh = half8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            fl = half8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        };

        // LayerNorm statistics of this lane's token: its 64 channels (16 blocks x 4) + the other three lane groups' (xor 16, 32)
#define ENC_ROWSTATS(VAL, EPS, MEAN, RSTD)                                                                            \
        float MEAN, RSTD;                                                                                             \
        {                                                                                                             \
            float sum_ = 0.f;                                                                                         \
            _Pragma("unroll") for (int b = 0; b < NB16; ++b) _Pragma("unroll") for (int r = 0; r < 4; ++r) sum_ += VAL(b, r); \
            sum_ = add_xor32(add_xor16(sum_));                                                                        \
            MEAN = sum_ / (float)EC;                                                                                  \
            float sq_ = 0.f;                                                                                          \
            _Pragma("unroll") for (int b = 0; b < NB16; ++b) _Pragma("unroll") for (int r = 0; r < 4; ++r)            \
                sq_ += (VAL(b, r) - MEAN) * (VAL(b, r) - MEAN);                                                       \
            sq_ = add_xor32(add_xor16(sq_));                                                                          \
            RSTD = 1.f / sqrtf(sq_ / (float)EC + (EPS));                                                              \
        }
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        half8 ah[NKS], al[NKS];          // operand of the next GEMM: phi(q), then the message, then norm1(merge)
        half8 kb[32];                    // KV^T fragments (hi, lo) of the wave's first sequence, between the q GEMM and S2
        float Z[NKS];
        // ---- S1: q = W_q x, phi(q), Z ------------------------------------------------------------------------------------
        {
            f32x4 am[NB16], ax[NB16];
#pragma unroll
            for (int b = 0; b < NB16; ++b) am[b] = ax[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                half8 th, tl;
                xfrag(ks, th, tl);
                slab_step<8, 1, 0>(ring, lane, am, ax, &th, &tl, wa, wb);
                slab_step<8, 1, 8>(ring, lane, am, ax, &th, &tl, wb, wa);
            }
            ENC_STAMP(2)
            // KV^T fragments of the wave's first sequence: ordinary 16-byte loads, the first half requested now, consumed in S2 (the L2
            // round trip runs under phi / Z; left to itself the compiler waited for every pair: sixteen serial round trips)
            {
                const char* img = g.kvimg + (int64_t)min(n_first, g.N - 1) * KVIMG + lane * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) kb[i] = *reinterpret_cast<const half8*>(img + i * 1024);     // heads 0-3 now,
                __builtin_amdgcn_sched_barrier(0);                                                       // 4-7 at the top of S2
            }
            const float* ks = reinterpret_cast<const float*>(s_ks + (row >= bound ? 1024 : 0));
            float zp[NKS];
#pragma unroll
            for (int h = 0; h < NKS; ++h) {                      // head h = blocks 2 h, 2 h + 1
                float v[2][4];
                float z = 0.f;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[b][r] = am[2 * h + b][r] + ax[2 * h + b][r] * (1.f / 2048.f);
                    if (g.dbg && g.dbg_stage == 1) dump16(g, v[b], 2 * h + b, row, valid, grp);
                    const f32x4 k4 = *reinterpret_cast<const f32x4*>(ks + 16 * (2 * h + b) + 4 * grp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[b][r] = phi_fast(v[b][r]) * qm;
                        z += v[b][r] * k4[r];
                    }
                }
                zp[h] = z;
                to_frag16(v[0], v[1], ah[h], al[h]);
            }
            // the four lane groups' partial sums: eight independent exchanges per round instead of eight serial pairs
#pragma unroll
            for (int h = 0; h < NKS; ++h) zp[h] = add_xor16(zp[h]);
#pragma unroll
            for (int h = 0; h < NKS; ++h) zp[h] = add_xor32(zp[h]);
#pragma unroll
            for (int h = 0; h < NKS; ++h) Z[h] = 1.f / (zp[h] + g.attn_eps);
        }
        ENC_STAMP(3)
        // ---- S2: message^T of head h = KV_h^T phi(q_h)^T: two 16-row blocks, K = 32 = one k-step ------------------------------
        {
            f32x4 mm[NB16], mx[NB16];
#pragma unroll
            for (int b = 0; b < NB16; ++b) mm[b] = mx[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            // tokens of the other sequence contribute zero columns
            {
                const bool mine = n_tok == min(n_first, g.N - 1);
                const char* img = g.kvimg + (int64_t)min(n_first, g.N - 1) * KVIMG + lane * 16;
#pragma unroll
                for (int i = 16; i < 32; ++i) kb[i] = *reinterpret_cast<const half8*>(img + i * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NB16; ++f) {
                    const half8 qh = mine ? ah[f >> 1] : zero8;
                    mm[f] = mfma16(kb[2 * f], qh, mm[f]);
                    mx[f] = mfma16(kb[2 * f + 1], qh, mx[f]);
                }
#pragma unroll
                for (int f = 0; f < NB16; ++f) {
                    const half8 ql = mine ? al[f >> 1] : zero8;
                    mx[f] = mfma16(kb[2 * f], ql, mx[f]);
                }
            }
            if (two) {                                               // the wave's second sequence (uniform branch, rare)
                const bool mine = n_tok == n_first + 1;
                const char* img = g.kvimg + (int64_t)(n_first + 1) * KVIMG + lane * 16;
#pragma unroll
                for (int i = 0; i < 32; ++i) kb[i] = *reinterpret_cast<const half8*>(img + i * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NB16; ++f) {
                    const half8 qh = mine ? ah[f >> 1] : zero8, ql = mine ? al[f >> 1] : zero8;
                    mm[f] = mfma16(kb[2 * f], qh, mm[f]);
                    mx[f] = mfma16(kb[2 * f + 1], qh, mx[f]);
                    mx[f] = mfma16(kb[2 * f], ql, mx[f]);
                }
            }
#pragma unroll
            for (int h = 0; h < NKS; ++h) {
                float v[2][4];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[b][r] = ((mm[2 * h + b][r] + mx[2 * h + b][r] * (1.f / 2048.f)) * Z[h]) * Sf;
                    if (g.dbg && g.dbg_stage == 2) dump16(g, v[b], 2 * h + b, row, valid, grp);
                }
                to_frag16(v[0], v[1], ah[h], al[h]);
            }
        }
        ENC_STAMP(4)
        // ---- S3: merge, LayerNorm1 ------------------------------------------------------------------------------------------
        {
            f32x4 am[NB16], ax[NB16];
#pragma unroll
            for (int b = 0; b < NB16; ++b) am[b] = ax[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                slab_step<8, 1, 0>(ring, lane, am, ax, &ah[ks], &al[ks], wa, wb);
                slab_step<8, 1, 8>(ring, lane, am, ax, &ah[ks], &al[ks], wb, wa);
            }
            ENC_STAMP(5)
#define ENC_V3(b, r) (am[b][r] + ax[b][r] * (1.f / 2048.f))
            ENC_ROWSTATS(ENC_V3, g.eps1, mean, rstd)
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                float v[2][4];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(s_ln + 16 * (2 * s + b) + 4 * grp);
                    const f32x4 bt = *reinterpret_cast<const f32x4*>(s_ln + EC + 16 * (2 * s + b) + 4 * grp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[b][r] = (ENC_V3(2 * s + b, r) - mean) * rstd * gm[r] + bt[r];
                    if (g.dbg && g.dbg_stage == 3) dump16(g, v[b], 2 * s + b, row, valid, grp);
                }
                to_frag16(v[0], v[1], ah[s], al[s]);
            }
#undef ENC_V3
        }
        ENC_STAMP(6)
        // ---- S4: mlp.2(relu(mlp.0([x | m]))) in eight 64-channel chunks of the hidden layer; S5: x + LayerNorm2(.) ----------
        {
            f32x4 om[NB16], ox[NB16];
#pragma unroll
            for (int b = 0; b < NB16; ++b) om[b] = ox[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int hc = 0; hc < 8; ++hc) {
                f32x4 hm[NB16], hx[NB16];                         // only blocks 0..3 are used (the arrays are register-allocated per element)
#pragma unroll
                for (int b = 0; b < 4; ++b) hm[b] = hx[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; u += 2) {                  // k-steps 2 u, 2 u + 1: x channels
                    half8 th[2], tl[2];
                    xfrag(2 * u, th[0], tl[0]);
                    xfrag(2 * u + 1, th[1], tl[1]);
                    slab_step<4, 2, 0>(ring, lane, hm, hx, th, tl, wa, wb);
                    xfrag(2 * u + 2, th[0], tl[0]);
                    xfrag(2 * u + 3, th[1], tl[1]);
                    slab_step<4, 2, 0>(ring, lane, hm, hx, th, tl, wb, wa);
                }
#pragma unroll
                for (int u = 0; u < 4; u += 2) {                  // k-steps 8 + 2 u, 9 + 2 u: norm1(merge) channels
                    slab_step<4, 2, 0>(ring, lane, hm, hx, ah + 2 * u, al + 2 * u, wa, wb);
                    slab_step<4, 2, 0>(ring, lane, hm, hx, ah + 2 * u + 2, al + 2 * u + 2, wb, wa);
                }
                half8 hh[2], hl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float hv[2][4];
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            hv[b][r] = fmaxf(hm[2 * t + b][r] + hx[2 * t + b][r] * (1.f / 2048.f), 0.f);
                    to_frag16(hv[0], hv[1], hh[t], hl[t]);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    slab_step<8, 1, 0>(ring, lane, om, ox, &hh[t], &hl[t], wa, wb);
                    slab_step<8, 1, 8>(ring, lane, om, ox, &hh[t], &hl[t], wb, wa);
                }
            }
            ENC_STAMP(7)
#define ENC_V5(b, r) (om[b][r] + ox[b][r] * (1.f / 2048.f))
            ENC_ROWSTATS(ENC_V5, g.eps2, mean, rstd)
#pragma unroll
            for (int b = 0; b < NB16; ++b) {
                float v[4];
                if (g.dbg && g.dbg_stage == 4) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ENC_V5(b, r);
                    dump16(g, v, b, row, valid, grp);
                }
                const f32x4 gm = *reinterpret_cast<const f32x4*>(s_ln + 2 * EC + 16 * b + 4 * grp);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(s_ln + 3 * EC + 16 * b + 4 * grp);
                // residual x: this lane's 4 channels of block b sit where its output goes (read, then overwritten below)
                const int o = stg_off(tok, 2 * b + (grp >> 1)) + 8 * (grp & 1);
                const half4 xrh = *reinterpret_cast<const half4*>(stg + o);
                const half4 xrl = *reinterpret_cast<const half4*>(stg + PLANE + o);
                half4 h4, l4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xr = (float)xrh[r] + (float)xrl[r] * (1.f / 2048.f);
                    const float y = xr + ((ENC_V5(b, r) - mean) * rstd * gm[r] + bt[r]);
                    _Float16 a, c;
                    split_f32(y, a, c);
                    h4[r] = a;
                    l4[r] = c;
                }
                *reinterpret_cast<half4*>(stg + o) = h4;
                *reinterpret_cast<half4*>(stg + PLANE + o) = l4;
            }
#undef ENC_V5
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ENC_STAMP(8)
            {
                // whole-row stores (lane-linear: the swizzle is undone on the LDS read).  The tile is read out of the staging first,
                // then the NEXT tile's x rows are requested into it, then the stores are issued.
                const int fl = fresh_lane();
                const int trow = fl >> 5, p = fl & 31;
                uint4 dh[8], dl[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int t = 2 * i + trow;
                    dh[i] = *reinterpret_cast<const uint4*>(stg + t * 512 + ((p ^ (t & 15)) << 4));
                    dl[i] = *reinterpret_cast<const uint4*>(stg + PLANE + t * 512 + ((p ^ (t & 15)) << 4));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (tile + (int)gridDim.x < g.ntiles) load_x(tile + (int)gridDim.x);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int t = 2 * i + trow;
                    const int64_t rr = row0 + t;
                    if (rr < g.M) {
                        const int cc = p << 3;
                        if (g.oh) {
                            *reinterpret_cast<uint4*>(g.oh + rr * g.ldo + cc) = dh[i];
                            *reinterpret_cast<uint4*>(g.ol + rr * g.ldo + cc) = dl[i];
                        }
                        if (g.o32) {
                            const half8 h8 = *reinterpret_cast<const half8*>(&dh[i]), l8 = *reinterpret_cast<const half8*>(&dl[i]);
                            float* o32 = g.o32 + rr * g.ldo32 + cc;
                            f32x4 f0, f1;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                f0[e] = (float)h8[e] + (float)l8[e] * (1.f / 2048.f);
                                f1[e] = (float)h8[4 + e] + (float)l8[4 + e] * (1.f / 2048.f);
                            }
                            *reinterpret_cast<f32x4*>(o32) = f0;
                            *reinterpret_cast<f32x4*>(o32 + 4) = f1;
                        }
                    }
                }
            }
            ENC_STAMP(9)
        }
#undef ENC_ROWSTATS
    }
#undef ENC_STAMP
    wait_vmcnt<0>();        // prefetched slabs still in flight must land before the LDS allocation is released
}


// One slab of the k | v projection (classic orientation: A = x fragments, B = weight fragments): 2 k-steps x 4 blocks, walked as
// four (k-step, block pair) groups of 4 consecutive fragments [wh(b0), wl(b0), wh(b1), wl(b1)] and 6 MFMAs.  The KV accumulators of
// the eight heads take 256 registers, so the apply kernel's whole-slab double buffer (128 registers) does not fit here: the
// fragments travel through a WINDOW of two groups (32 registers) -- while group q is multiplied, group q + 1 is read; the last
// group of a slab reads the first group of the NEXT slab.  That needs the next slab visible one step early: s_barrier(s) publishes
// slab s + 1 (so two slabs are resident and being read, one is landing, one slot is refilled: the prefetch distance is two steps
// instead of three).  On entry `wa` holds group 0 of slab `next`.
__device__ __forceinline__ void kv_slab_step(SlabRing& ring, int lane, f32x4 (&dm)[NB16], f32x4 (&dx)[NB16], const half8* xh,
                                             const half8* xl, half8 (&wa)[4], half8 (&wb)[4]) {
    wait_vmcnt<4>();                                 // this wave's pieces of slab next + 1 (next + 2 may still fly)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                    // slab next + 1 is visible; every wave is done with slab next - 1
    __builtin_amdgcn_sched_barrier(0);
    const char* cslab = ring.ring + (ring.next % NSTG) * SLAB + lane * 16;
    const char* nslab = ring.ring + ((ring.next + 1) % NSTG) * SLAB + lane * 16;
    const unsigned gn = ring.next + NSTG - 1;       // into the slot of slab next - 1
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ks = q >> 1, b0 = 2 * (q & 1), b1 = b0 + 1;
        half8(&c)[4] = (q & 1) ? wb : wa;
        half8(&nx)[4] = (q & 1) ? wa : wb;
        const char* src = q < 3 ? cslab + (q + 1) * 4096 : nslab;
        dm[b0] = mfma16(xh[ks], c[0], dm[b0]);
        nx[0] = *reinterpret_cast<const half8*>(src);
        __builtin_amdgcn_sched_barrier(0);
        dx[b0] = mfma16(xh[ks], c[1], dx[b0]);
        nx[1] = *reinterpret_cast<const half8*>(src + 1024);
        __builtin_amdgcn_sched_barrier(0);
        dm[b1] = mfma16(xh[ks], c[2], dm[b1]);
        ring.issue_piece(gn, q);
        __builtin_amdgcn_sched_barrier(0);
        dx[b1] = mfma16(xh[ks], c[3], dx[b1]);
        nx[2] = *reinterpret_cast<const half8*>(src + 2048);
        __builtin_amdgcn_sched_barrier(0);
        dx[b0] = mfma16(xl[ks], c[0], dx[b0]);
        nx[3] = *reinterpret_cast<const half8*>(src + 3072);
        __builtin_amdgcn_sched_barrier(0);
        dx[b1] = mfma16(xl[ks], c[2], dx[b1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    ring.advance();
}

// Chunk partials -> the "apply image" of the d_model-256 fused encoder layer (encoder256.hip): per sequence, for head h and
// row block rb the A fragment pair (hi, lo planes of 1 KB, lane-linear) of KV_h^T rows 16 rb .. 16 rb + 15 -- lane (i, g) slot j
// = KV_h[d][16 rb + i] with d = 16 (j >> 2) + 4 g + (j & 3), the k order in which that kernel's accumulators hold phi(q_h) --
// then Ksum[256] as fp32.  Same four-way interleaved chunk sum as la_kv_finalize (fixed order: deterministic).

__global__ __launch_bounds__(256) void enc256_image_kernel(const float* __restrict__ part, char* __restrict__ img, int nchunks) {
    const int nh = blockIdx.x, n = nh >> 3, h = nh & 7;
    const float* p0 = part + ((int64_t)n * nchunks * 8 + h) * KVSZ;
    const int64_t stride = (int64_t)8 * KVSZ;
    char* out = img + (int64_t)n * KVIMG;
    for (int e = threadIdx.x; e < KVSZ; e += 256) {
        int src = e, rb = 0, lane = 0, j = 0;
        if (e < 1024) {
            rb = e >> 9; lane = (e >> 3) & 63; j = e & 7;
            const int i = lane & 15, g = lane >> 4;
            src = (16 * (j >> 2) + 4 * g + (j & 3)) * 32 + 16 * rb + i;
        }
        const float* p = p0 + src;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int c = 0;
        for (; c + 4 <= nchunks; c += 4) {
            s0 += p[(c + 0) * stride];
            s1 += p[(c + 1) * stride];
            s2 += p[(c + 2) * stride];
            s3 += p[(c + 3) * stride];
        }
        for (; c < nchunks; ++c) s0 += p[c * stride];
        const float val = (s0 + s1) + (s2 + s3);
        if (e < 1024) {
            _Float16 hi, lo;
            split_f32(val, hi, lo);
            char* q = out + ((h * 2 + rb) * 2) * 1024 + lane * 16 + j * 2;
            *reinterpret_cast<_Float16*>(q) = hi;
            *reinterpret_cast<_Float16*>(q + 1024) = lo;
        } else {
            reinterpret_cast<float*>(out + 32768)[32 * h + (e - 1024)] = val;
        }
    }
}


// =====================================================================================================================
// enc256_kv_kernel: the SOURCE side of a layer application -- k | v projection fused with K1's partial sums.
//   k | v = W_kv x (never stored), phi(k), v / S and the padding mask applied in registers, then per head
//   KV_h += phi(k_h)^T v_h and Ksum_h += sum_t phi(k_h) for the workgroup's chunk of the sequence; the chunk partials go to memory in
//   K1's layout ([n][chunk][head][KV[d][v] | Ksum[d]]) and enc256_image_kernel reduces them (fixed order) into the apply image.
// Replaces ops.linear(src, W_kv) (a 2 KB / row fp32 round trip through HBM) + la_kv_partial_staged.
// Classic orientation D[token][channel] = x[token][k] W[channel][k] on v_mfma_f32_16x16x32_f16: lane = channel (lane & 15),
// registers = tokens 4 g + r of a 16-token block -- so that the contraction over tokens of phi(k)^T v is again an MFMA whose A and
// B fragments ARE the k and v accumulators: slots j < 4 of lane group g = tokens 4 g + j, slots j >= 4 zero (16 of the MFMA's 32
// k positions; the 12 KV MFMAs of a head are 1/9 of its MFMAs).  A wave owns a 16-token block per pass; its x fragments (natural k
// order, 64 registers) feed all eight heads; KV of the eight heads stays in 256 accumulator registers for the whole chunk.
// Weight stream: 32 slabs, head h = slabs 4 h .. 4 h + 3, slab u = k-steps 2 u, 2 u + 1 x rows [k_h (2 blocks) | v_h (2 blocks)],
// through the same ring; the fragments pass through a two-group register window (kv_slab_step).
// =====================================================================================================================
constexpr int NSLAB_KV = 32;
constexpr int KV_MASK_MAX = 8192;         // mask entries of a chunk kept in LDS
constexpr int SMEM_KV = RING + 4 * STG + 16384;      // the mask entries (8 KB) during the passes; 128 KB of KV partials + 16 KB of Ksum
// This file keeps the SLP vectoriser and its own packed-fp32 expressions; the op_sel-on-src1 form that misreads beside a wave's own
// in-flight MFMAs (r06, csrc/Makefile) is rejected by the build's ISA gate for this translation unit.  One workgroup per CU stays a
// design premise of both kernels (one ~500-register wave per SIMD): the LDS footprints say so explicitly.
static_assert(SMEM_APPLY > 80 * 1024 && SMEM_KV > 80 * 1024, "one workgroup per CU is a design premise of these kernels");
                                                      // partials for the cross-wave sums at the end
static_assert(KV_MASK_MAX <= 16384 && 4 * NKS * 16 * 64 * 4 == RING + 4 * STG, "enc256_kv_kernel: LDS layout of the final reduction");

struct Kv256Args {
    const _Float16 *xh, *xl;     // source rows as split planes, row stride ldx (elements)
    int64_t ldx;
    unsigned xbytes;
    const char* wstream;         // NSLAB_KV slabs
    const uint8_t* kvmask;       // [N][km_per_seq] or null
    int kv_group, km_per_seq;
    float* part;                 // [N][nchunks][8][KVSZ] out
    int S, N, rows_per_chunk, nchunks;
};

__global__ __launch_bounds__(256) void enc256_kv_kernel(Kv256Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, grp = lane >> 4;
    char* stg = smem + RING + wave * STG;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int s_begin = chunk * g.rows_per_chunk;
    const int s_end = min(g.S, s_begin + g.rows_per_chunk);
    const int nblocks = (s_end - s_begin + 15) / 16;
    const int niter = (nblocks + 3) / 4;            // every wave runs the same number of passes (shared weight stream)

    SlabRing ring;
    ring.ring = smem;
    ring.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.wstream, 0, NSLAB_KV * SLAB, 0x00020000);
    ring.lane_off = (unsigned)(wave * 4096 + lane * 16);
    ring.wave = wave;
    ring.nslab = NSLAB_KV;
#pragma unroll
    for (int gq = 0; gq < NSTG - 1; ++gq) ring.issue(gq);        // slabs 0, 1, 2; step s refills the slot of slab s - 1 with s + 3
    ring.next = 0;

    const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)g.xh, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)g.xl, 0, g.xbytes, 0x00020000);
    // v / S as one multiplication by 1 / S (r05: the IEEE division was ten VALU instructions per element of a per-head epilogue that runs
    // beside nothing -- one wave per SIMD; the quotient differs from the division's by at most one rounding, 6e-8 relative, inside the
    // 2^-22 of the split operands it is multiplied into next)
    const float inv_S = 1.f / (float)g.S;
    // the chunk's mask entries once into LDS (as byte loads inside the block loop every entry was its own drain of the DMA queue)
    uint8_t* s_mask = reinterpret_cast<uint8_t*>(smem + RING + 4 * STG);
    const int m_lo = s_begin / g.kv_group;
    const int m_cnt = g.kvmask ? (s_end - 1) / g.kv_group - m_lo + 1 : 0;
    const bool lds_mask = g.kvmask && m_cnt <= KV_MASK_MAX;
    if (lds_mask)
        for (int i = tid; i < m_cnt; i += 256) s_mask[i] = g.kvmask[(int64_t)n * g.km_per_seq + m_lo + i];
    // x rows of the wave's block `it` -> staging (rows of 512 B per plane, 16-byte chunks XOR-swizzled on the source side)
    auto load_x = [&](int it) __attribute__((always_inline)) {
        const int s0 = s_begin + (it * 4 + wave) * 16;
        const int trow = lane >> 5, p = lane & 31;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tt = 2 * i + trow;
            const int srow = s0 + tt;
            const unsigned off = srow < s_end ? (unsigned)((((int64_t)n * g.S + srow) * g.ldx + ((p ^ (tt & 15)) << 3)) * 2) : g.xbytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_void*)(stg + i * 1024), 16, off, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (lds_void*)(stg + PLANE + i * 1024), 16, off, 0, 0, 0);
        }
    };
    load_x(0);
    half8 wa[4], wb[4];                      // the fragment window: group being multiplied / group being fetched (ping-pong)
    wait_vmcnt<0>();                         // slabs 0-2 and the first x block
    __syncthreads();                         // ... of every wave; and the mask entries
#pragma unroll
    for (int f = 0; f < 4; ++f) wa[f] = *reinterpret_cast<const half8*>(smem + f * 1024 + lane * 16);

    f32x4 kvm[NKS][4], kvx[NKS][4];          // KV of head h, block (kb, vb) = [2 kb + vb]: rows d = 16 kb + 4 g + r, column v = 16 vb + lane & 15
    float ksum[NKS][2];                      // sum over this lane's tokens of phi(k) of channel 32 h + 16 kb + (lane & 15)
#pragma unroll
    for (int h = 0; h < NKS; ++h) {
#pragma unroll
        for (int b = 0; b < 4; ++b) kvm[h][b] = kvx[h][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        ksum[h][0] = ksum[h][1] = 0.f;
    }
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    for (int it = 0; it < niter; ++it) {
        const int s0 = s_begin + (it * 4 + wave) * 16;    // first token of this wave's block (may be past the chunk: all masked)
        // masks of the 4 tokens this lane's accumulator registers hold: token s0 + 4 g + r
        float tm[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sidx = s0 + 4 * grp + r;
            float mk = sidx < s_end ? 1.f : 0.f;
            if (lds_mask) { if (sidx < s_end) mk = (float)s_mask[sidx / g.kv_group - m_lo]; }
            else if (g.kvmask && sidx < s_end) mk = (float)g.kvmask[(int64_t)n * g.km_per_seq + sidx / g.kv_group];
            tm[r] = mk;
        }
        wait_vmcnt<0>();                                  // the x block has landed (and the slabs in flight: once per pass)
        // A fragments of x for all eight k-steps: lane (token, g) slots = channels 32 s + 8 g + j (natural order: one 16-byte read)
        half8 xh[NKS], xl[NKS];
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            xh[s] = *reinterpret_cast<const half8*>(stg + stg_off(col, 4 * s + grp));
            xl[s] = *reinterpret_cast<const half8*>(stg + PLANE + stg_off(col, 4 * s + grp));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (it + 1 < niter) load_x(it + 1);               // the staging tile is free again: the next block flies under this pass
#pragma unroll
        for (int h = 0; h < NKS; ++h) {
            f32x4 dm[NB16], dx[NB16];                     // blocks 0, 1 = k channels of head h, 2, 3 = v channels (only 0..3 used)
#pragma unroll
            for (int b = 0; b < 4; ++b) dm[b] = dx[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            kv_slab_step(ring, lane, dm, dx, xh + 0, xl + 0, wa, wb);
            kv_slab_step(ring, lane, dm, dx, xh + 2, xl + 2, wa, wb);
            kv_slab_step(ring, lane, dm, dx, xh + 4, xl + 4, wa, wb);
            kv_slab_step(ring, lane, dm, dx, xh + 6, xl + 6, wa, wb);
            half8 kfh[2], kfl[2], vfh[2], vfl[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float kf[4], vf[4];
                const float zz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    kf[r] = phi_fast(dm[b][r] + dx[b][r] * (1.f / 2048.f)) * tm[r];
                    vf[r] = ((dm[2 + b][r] + dx[2 + b][r] * (1.f / 2048.f)) * tm[r]) * inv_S;
                    ksum[h][b] += kf[r];
                }
                to_frag16(kf, zz, kfh[b], kfl[b]);        // slots 4..7 (the other 16 k positions of the MFMA) stay zero
                to_frag16(vf, zz, vfh[b], vfl[b]);
            }
            // KV_h[d][v] += sum over the block's tokens: A = phi(k)^T (lane = d), B = v (lane = v channel), same token slots
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int vb = 0; vb < 2; ++vb) {
                    kvm[h][2 * kb + vb] = mfma16(kfh[kb], vfh[vb], kvm[h][2 * kb + vb]);
                    kvx[h][2 * kb + vb] = mfma16(kfl[kb], vfh[vb], kvx[h][2 * kb + vb]);
                }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int vb = 0; vb < 2; ++vb)
                    kvx[h][2 * kb + vb] = mfma16(kfh[kb], vfl[vb], kvx[h][2 * kb + vb]);
        }
    }
    (void)zero8;
    wait_vmcnt<0>();
    __syncthreads();                                // ring and staging are free: reuse them for the cross-wave sums
    // every wave parks its partial KV (fp32) at [wave][h][b][r][lane] and its Ksum at [wave][h][kb][lane]: 4 x 32 KB + 4 x 4 KB
    float* park = reinterpret_cast<float*>(smem);
    float* ksp = park + 4 * NKS * 16 * 64;
#pragma unroll
    for (int h = 0; h < NKS; ++h) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                park[((wave * NKS + h) * 16 + b * 4 + r) * 64 + lane] = kvm[h][b][r] + kvx[h][b][r] * (1.f / 2048.f);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) ksp[((wave * NKS + h) * 2 + kb) * 64 + lane] = ksum[h][kb];
    }
    __syncthreads();
    // wave w finishes heads 2 w, 2 w + 1: fixed summation order over the waves (deterministic)
    float* out = g.part + ((int64_t)n * g.nchunks + chunk) * NKS * KVSZ;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int h = 2 * wave + hh;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) a += park[((w * NKS + h) * 16 + b * 4 + r) * 64 + lane];
                out[h * KVSZ + (16 * (b >> 1) + 4 * grp + r) * 32 + 16 * (b & 1) + col] = a;
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += ksp[((w * NKS + h) * 2 + kb) * 64 + lane];
            a = add_xor32(add_xor16(a));            // the four lane groups hold different tokens of the same channel
            if (grp == 0) out[h * KVSZ + 1024 + 16 * kb + col] = a;
        }
    }
}

dfsfm::SmemAttr attr_apply256;

dfsfm::SmemAttr attr_kv256;

}  // namespace

void dfsfm_enc::enc256_launch_image(const float* part, char* img, int N, int nchunks, hipStream_t stream) {
    hipLaunchKernelGGL(enc256_image_kernel, dim3(N * 8), dim3(256), 0, stream, part, img, nchunks);
}

// rows of a sequence per workgroup of enc256_kv_kernel: a multiple of 64 (4 waves x 16 tokens), about one workgroup per CU
static int kv256_chunk_rows(int N, int S) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t steps = ((int64_t)N * ((S + 63) / 64) + cus - 1) / cus;       // 64-token passes per workgroup
    int rows = (int)(steps < 1 ? 1 : steps) * 64;
    const int smax = (S + 63) / 64 * 64;
    return rows > smax ? smax : rows;
}

extern "C" size_t dfsfm_encoder256_kv_workspace(int N, int S) {
    if (N <= 0 || S <= 0) return 0;
    const int rows = kv256_chunk_rows(N, S);
    const int nchunks = (S + rows - 1) / rows;
    return dfsfm::align_up((size_t)N * nchunks * 8 * KVSZ * sizeof(float), 256);
}

extern "C" int dfsfm_encoder256_kv_f32(const void* src_hi, const void* src_lo, int64_t ld_src, int N, int S,
                                       const void* wstream_kv, const uint8_t* kv_mask, int kv_group, void* kv_image,
                                       void* workspace, size_t workspace_bytes, void* stream_) {
    if (N == 0) return DFSFM_OK;
    if (!src_hi || !src_lo || !wstream_kv || !kv_image || !workspace) return DFSFM_E_BADARG;
    if (N < 0 || S <= 0 || kv_group <= 0 || ld_src < EC) return DFSFM_E_BADARG;
    const int64_t span = ((int64_t)N * S - 1) * ld_src * 2 + EC * 2;
    if (N > 65535 || (ld_src & 7) || span >= (int64_t)0xFFFFFFF0) return DFSFM_E_UNSUPPORTED;
    for (const void* p : {src_hi, src_lo, wstream_kv, (const void*)kv_image, (const void*)workspace})
        if (reinterpret_cast<uintptr_t>(p) & 15) return DFSFM_E_UNSUPPORTED;
    if (workspace_bytes < dfsfm_encoder256_kv_workspace(N, S)) return DFSFM_E_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Kv256Args g{};
    g.xh = static_cast<const _Float16*>(src_hi);
    g.xl = static_cast<const _Float16*>(src_lo);
    g.ldx = ld_src;
    g.xbytes = (unsigned)span;
    g.wstream = static_cast<const char*>(wstream_kv);
    g.kvmask = kv_mask;
    g.kv_group = kv_group;
    g.km_per_seq = (S + kv_group - 1) / kv_group;
    g.part = static_cast<float*>(workspace);
    g.S = S;
    g.N = N;
    g.rows_per_chunk = kv256_chunk_rows(N, S);
    g.nchunks = (S + g.rows_per_chunk - 1) / g.rows_per_chunk;
    attr_kv256.ensure(reinterpret_cast<const void*>(&enc256_kv_kernel), SMEM_KV);
    hipLaunchKernelGGL(enc256_kv_kernel, dim3((unsigned)g.nchunks, (unsigned)N), dim3(256), SMEM_KV, stream, g);
    dfsfm_enc::enc256_launch_image(g.part, static_cast<char*>(kv_image), N, g.nchunks, stream);
    return dfsfm::check_launch("dfsfm_encoder256_kv_f32");
}

extern "C" int dfsfm_encoder256_apply_f32(const void* x_hi, const void* x_lo, int64_t ldx, int N, int L, int S,
                                          const void* wstream, const void* kv_image, const uint8_t* q_mask, int q_group,
                                          const float* gamma1, const float* beta1, float eps1, const float* gamma2,
                                          const float* beta2, float eps2, float attn_eps, void* out_hi, void* out_lo,
                                          int64_t ldo, float* out32, int64_t ldo32, float* debug, int debug_stage,
                                          void* stream_) {
    if (N == 0) return DFSFM_OK;
    if (!x_hi || !x_lo || !wstream || !kv_image || !gamma1 || !beta1 || !gamma2 || !beta2) return DFSFM_E_BADARG;
    if ((out_hi == nullptr) != (out_lo == nullptr) || (!out_hi && !out32)) return DFSFM_E_BADARG;
    if (N < 0 || L <= 0 || S <= 0 || q_group <= 0 || ldx < EC || (out_hi && ldo < EC) || (out32 && ldo32 < EC))
        return DFSFM_E_BADARG;
    if (L < 16) return DFSFM_E_UNSUPPORTED;          // a wave's 16 tokens may touch at most two sequences
    const int64_t M = (int64_t)N * L;
    const int64_t span = (M - 1) * ldx * 2 + EC * 2;
    if ((ldx & 7) || (ldo & 7) || (ldo32 & 3) || span >= (int64_t)0xFFFFFFF0) return DFSFM_E_UNSUPPORTED;
    if ((int64_t)N * KVIMG >= (int64_t)0xFFFFFFF0) return DFSFM_E_UNSUPPORTED;      // 32-bit buffer offsets into the sequence images
    for (const void* p : {x_hi, x_lo, wstream, kv_image, (const void*)out_hi, (const void*)out_lo, (const void*)out32,
                          (const void*)gamma1, (const void*)beta1, (const void*)gamma2, (const void*)beta2})
        if (reinterpret_cast<uintptr_t>(p) & 15) return DFSFM_E_UNSUPPORTED;
    Apply256Args g{};
    g.xh = static_cast<const _Float16*>(x_hi);
    g.xl = static_cast<const _Float16*>(x_lo);
    g.ldx = ldx;
    g.xbytes = (unsigned)span;
    g.wstream = static_cast<const char*>(wstream);
    g.kvimg = static_cast<const char*>(kv_image);
    g.qmask = q_mask;
    g.q_group = q_group;
    g.qm_per_seq = (L + q_group - 1) / q_group;
    g.g1 = gamma1; g.b1 = beta1; g.g2 = gamma2; g.b2 = beta2;
    g.eps1 = eps1; g.eps2 = eps2; g.attn_eps = attn_eps;
    g.oh = static_cast<_Float16*>(out_hi);
    g.ol = static_cast<_Float16*>(out_lo);
    g.ldo = ldo;
    g.o32 = out32;
    g.ldo32 = ldo32;
    g.M = M;
    g.L = L; g.N = N; g.S = S;
    g.ntiles = (int)((M + 63) / 64);
    g.dbg = debug;
    g.dbg_stage = debug_stage;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = g.ntiles < cus ? g.ntiles : cus;   // persistent: one workgroup per CU walks the tiles
    attr_apply256.ensure(reinterpret_cast<const void*>(&enc256_apply_kernel), SMEM_APPLY);
    hipLaunchKernelGGL(enc256_apply_kernel, dim3((unsigned)grid), dim3(256), SMEM_APPLY, static_cast<hipStream_t>(stream_), g);
    return dfsfm::check_launch("dfsfm_encoder256_apply_f32");
}
