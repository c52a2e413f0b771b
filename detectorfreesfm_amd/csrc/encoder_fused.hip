// K2 + K1 + K10 fused -- one LoFTREncoderLayer application in TWO kernels, for gfx950 (MI355X), d_model 128, 8 heads.
//
// Replaces, for the refinement head's transformer (and any d_model-128 LoFTR encoder layer),
//   LoFTREncoderLayer.forward   src/MultiviewMatcher/matcher_module/transformer.py:66-95
//                               (third_party/LoFTR/src/loftr/loftr_module/transformer.py:35-58)
//   LinearAttention.forward     src/MultiviewMatcher/matcher_module/linear_attention.py:28-60
// which the unfused path runs as five GEMM launches + three attention launches that round-trip every activation of the
// 2.25 M-row token matrix through HBM (q|k|v, message, merged message, the 256-wide MLP hidden: ~9 KB per row and layer).
// Here a token row is read once per kernel (512 B as fp16x2 split planes) and written once (512 B):
//
//   enc_kv_kernel     source tokens -> k|v = W_kv x (never stored) -> per sequence  KV = sum phi(k)^T (v/S), Ksum = sum phi(k)
//                     written as the "apply image": KV^T already in MFMA A-fragment order (fp16 hi/lo) + Ksum.
//   enc_apply_kernel  x tokens -> q = W_q x -> phi(q) KV Z S -> merge -> LayerNorm1 -> mlp.0([x | m]) -> ReLU -> mlp.2
//                     -> LayerNorm2 -> x + .   Every intermediate lives in registers.
//
// Arithmetic is the repository's fp16x2 split (value = hi + lo/2048, three v_mfma_f32_32x32x16_f16 per product, fp32
// accumulation): fp32-class, lo*lo (2^-22) dropped.
//
// enc_apply_kernel is TOKEN-STATIONARY and TRANSPOSED: it computes out^T[channel][token] = W[channel][k] x^T[k][token], so the
// MFMA result of a 32x32 block has lane = token, registers = 16 channels -- and the B operand of the next GEMM wants
// lane = token, 8 k-values per lane: the accumulator of one GEMM becomes the operand of the next after an fp32 -> fp16x2
// conversion in registers, with no LDS round trip and no cross-lane traffic.  The price is a fixed permutation of the k index
// inside every 16-wide k-step (reg r of a block holds channel (r&3) + 8(r>>2) + 4*half), which the host applies to the weight
// columns once (ops.EncoderFusedWeights).  A wave owns 32 tokens; the 4 waves of a workgroup share only the weight stream,
// which the host lays out as a sequence of 16-KB "slabs" of ready-made A fragments (1 KB each, lane-linear): the kernel
// streams them global -> LDS with buffer_load ... lds through a 4-deep ring (one s_barrier per slab) and reads them with
// conflict-free ds_read_b128.  One wave per SIMD (512-register budget): accumulators + operands of a whole layer stay resident.
//
// enc_kv_kernel uses the classic orientation (lane = channel, registers = tokens) so that the contraction over tokens of
// phi(k)^T v is again an MFMA fed from accumulator registers; persistent workgroups walk the sequences with the 128 KB of k|v
// weight fragments RESIDENT in LDS (r06; no ring), the 4 waves take every fourth 32-token block of a sequence, partial sums are
// combined in a fixed order (run-to-run deterministic, independent of the batch).
// Build: csrc/Makefile compiles this file with -mllvm -amdgpu-mfma-vgpr-form (MFMA results in the vector register file: every
// accumulator here is read by VALU epilogues, and a VALU instruction cannot read an AGPR).
#include "common.h"
#include "enc_common.h"
#include <cstdlib>

namespace {

using namespace dfsfm;
using namespace dfsfm_enc;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int EC = 128;                  // d_model
constexpr int NBLK = EC / 32;            // 32-channel blocks
constexpr int NKS = EC / 16;             // k-steps over d_model
constexpr int STG = 16384;               // per-wave staging tile: 32 tokens x 128 channels, hi + lo planes
constexpr int SMEM_BYTES = RING + 4 * STG;
constexpr int KV_W_BYTES = 8 * SLAB;       // enc_kv_kernel: the whole k|v weight stream, resident (128 KB)
constexpr int KV_RED = 16384;             // cross-wave partial sums of one head pair
constexpr int KV_KSP = 4096;              // Ksum partials [wave][pair][lane]
constexpr int SMEM_KV = KV_W_BYTES + KV_RED + KV_KSP;
constexpr int SMEM_APPLY = SMEM_BYTES + 4 * EC * 4 + 4 * 1024;   // + the LayerNorms' gamma / beta (2 KB) + Ksum of a tile's two sequences per wave
// This file keeps the SLP vectoriser and its own packed-fp32 expressions.  The hazard behind the r04 / r05 rule "no packed fp32 beside
// MFMAs" is the op_sel-on-src1 FORM of such an instruction inside a wave with MFMAs in flight (r06, csrc/Makefile): the build's ISA gate
// rejects that form in this translation unit, whatever the occupancy.  One workgroup per CU stays a design premise of both kernels
// (512 registers per wave; enc_kv's resident weights): the LDS footprints say so explicitly.
static_assert(SMEM_APPLY > 80 * 1024 && SMEM_KV > 80 * 1024, "one workgroup per CU is a design premise of these kernels");
constexpr int NSLAB_APPLY = 32;          // q 4, merge 4, 4 x (mlp.0 chunk 4 + mlp.2 chunk 2)
constexpr int KVIMG = 16 * 1024 + 512;   // bytes per sequence: 16 KV^T fragments + Ksum[128]

__device__ __forceinline__ f32x16 mfma(const half8 a, const half8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// channel (within a 32-channel block) that accumulator register r of lane half h holds
__device__ __forceinline__ int dch(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// fp32 accumulator block -> the two B (or A) fragments it becomes for the next MFMA: k-step t of the block = registers 8t..8t+7.
// Same split as split_f32 (common.h) with the saturation expressed on the fp16 side: fp16(v) overflows to +-inf for
// |v| >= 65520, which one v_pk_min / v_pk_max pair per two values brings back to +-65504; lo is then finite as well.
__device__ __forceinline__ void to_frags(const float (&v)[16], half8& h0, half8& l0, half8& h1, half8& l1) {
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    const half2_t big = {(_Float16)65504.f, (_Float16)65504.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        half2_t h = {(_Float16)v[r], (_Float16)v[r + 1]};
        h = __builtin_elementwise_max(__builtin_elementwise_min(h, big), -big);
        const half2_t l = {(_Float16)((v[r] - (float)h[0]) * 2048.f), (_Float16)((v[r + 1] - (float)h[1]) * 2048.f)};
        if (r < 8) {
            h0[r] = h[0]; h0[r + 1] = h[1]; l0[r] = l[0]; l0[r + 1] = l[1];
        } else {
            h1[r - 8] = h[0]; h1[r - 7] = h[1]; l1[r - 8] = l[0]; l1[r - 7] = l[1];
        }
    }
}

// byte offset of 16-byte chunk c of token row t inside a staging plane (32 rows x 256 B): XOR swizzle so that the
// fragment-shaped 8/16-byte reads of 32 different rows spread over all banks
__device__ __forceinline__ int stg_off(int t, int c) { return t * 256 + ((c ^ (t & 15)) << 4); }

// one slab = KPS k-steps x NB blocks of (hi, lo) fragments; acc[b] += W(b, ks) * B[ks] with the 3-MFMA split product:
// am += W_hi B_hi, ax += W_lo B_hi, ay += W_hi B_lo (callers with NB = 4 pass ax for ay: an accumulator is then reused
// after four other MFMAs; with NB = 2 a third accumulator keeps every MFMA independent of its two predecessors).
template <int NB, int KPS>
__device__ __forceinline__ void slab_mma(const char* slab, int lane, f32x16 (&am)[NB], f32x16 (&ax)[NB], f32x16 (&ay)[NB],
                                         const half8* bh, const half8* bl) {
    static_assert(NB * KPS == 8, "a slab holds 16 fragments");
#pragma unroll
    for (int ks = 0; ks < KPS; ++ks) {
        half8 wh[NB], wl[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            wh[b] = *reinterpret_cast<const half8*>(slab + ((ks * NB + b) * 2 + 0) * 1024 + lane * 16);
            wl[b] = *reinterpret_cast<const half8*>(slab + ((ks * NB + b) * 2 + 1) * 1024 + lane * 16);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) am[b] = mfma(wh[b], bh[ks], am[b]);
#pragma unroll
        for (int b = 0; b < NB; ++b) ax[b] = mfma(wl[b], bh[ks], ax[b]);
#pragma unroll
        for (int b = 0; b < NB; ++b) ay[b] = mfma(wh[b], bl[ks], ay[b]);
    }
}

// The same with the ring refill inside: right after the barrier the four DMA requests of the refill (address arithmetic, M0,
// ~100 issue cycles each with one wave per SIMD) stood in front of the slab's first MFMA; here each one follows a group of MFMAs
// that is already executing.  sched_barriers pin the places (the compiler hoists the requests to the top otherwise).
// (r06, measured and not kept: requesting k-step ks + 1's fragments inside k-step ks -- lo fragments behind the group that reads
// this k-step's lo fragments, hi fragments behind the last group -- to expose one LDS latency per slab instead of one per k-step:
// enc_apply 1.845 -> 1.867 ms, MLP stage 360 -> 367 us per tile, same box, gpurun_out/r6n: the compiler's counted waits already let a
// k-step's MFMAs start as its fragments land, and the extra live fragments cost more than the latency they hide.)
template <int NB, int KPS>
__device__ __forceinline__ void slab_mma(SlabRing& ring, int lane, f32x16 (&am)[NB], f32x16 (&ax)[NB], f32x16 (&ay)[NB],
                                         const half8* bh, const half8* bl) {
    static_assert(NB * KPS == 8, "a slab holds 16 fragments");
    const char* slab = ring.acquire_wait();
    const unsigned gn = ring.next + NSTG - 1;
    int piece = 0;
#pragma unroll
    for (int ks = 0; ks < KPS; ++ks) {
        half8 wh[NB], wl[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            wh[b] = *reinterpret_cast<const half8*>(slab + ((ks * NB + b) * 2 + 0) * 1024 + lane * 16);
            wl[b] = *reinterpret_cast<const half8*>(slab + ((ks * NB + b) * 2 + 1) * 1024 + lane * 16);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) am[b] = mfma(wh[b], bh[ks], am[b]);
        if (KPS == 2 || (ks & 1) == 0) {           // KPS = 2: after both first groups of a k-step; KPS = 4: once per k-step
            __builtin_amdgcn_sched_barrier(0);
            ring.issue_piece(gn, piece++);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) ax[b] = mfma(wl[b], bh[ks], ax[b]);
        if (KPS == 2 || (ks & 1) == 1) {
            __builtin_amdgcn_sched_barrier(0);
            ring.issue_piece(gn, piece++);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) ay[b] = mfma(wh[b], bl[ks], ay[b]);
    }
    ring.advance();
}

struct ApplyArgs {
    const _Float16 *xh, *xl;     // x rows as split planes, row stride ldx (elements)
    int64_t ldx;
    unsigned xbytes;
    const char* wstream;         // NSLAB_APPLY slabs
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
    int L, N, S;                 // tokens per sequence on the x side, sequences, tokens per sequence on the source side
    int ntiles;                  // ceil(M / 128)
    float* dbg;                  // [M][128] fp32 dump of one intermediate (tests), or null
    int dbg_stage;               // 1 q, 2 message, 3 norm1(merge), 4 mlp output (before norm2)
    int stagger;                 // start delay per (blockIdx % 8), in s_sleep(127) units (~8 K cycles each... see the entry point)
};

// D-layout values of one block (lane = token, v[r] = channel 32 b + dch(r, half)) -> dbg rows.  Must stay inlined: a real call
// from this 512-register kernel (256 VGPRs + 256 AGPRs live) corrupted caller state on gfx950 / ROCm 7.2 -- outputs of later
// blocks saturated, another build faulted -- so nothing in this file is ever __noinline__.
__device__ __forceinline__ void dump(const ApplyArgs& g, const float (&v)[16], int b, int64_t row, bool valid, int half) {
    if (!valid) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) g.dbg[row * EC + 32 * b + dch(r, half)] = v[r];
}

__global__ __launch_bounds__(256) void enc_apply_kernel(ApplyArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    char* stg = smem + RING + wave * STG;
    // LayerNorm parameters in LDS: as global loads inside the tile loop every use drained the VM counter -- slab DMA included
    // (the compiler waits vmcnt(0) for an ordinary load while LDS-DMA is in flight) -- and exposed an L2 round trip twice per tile
    float* s_ln = reinterpret_cast<float*>(smem + SMEM_BYTES);               // [g1 | b1 | g2 | b2][EC]
    if (tid < EC) {
        s_ln[tid] = g.g1[tid];
        s_ln[EC + tid] = g.b1[tid];
        s_ln[2 * EC + tid] = g.g2[tid];
        s_ln[3 * EC + tid] = g.b2[tid];
    }
    __syncthreads();

    SlabRing ring;
    ring.ring = smem;
    ring.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.wstream, 0, NSLAB_APPLY * SLAB, 0x00020000);
    ring.lane_off = (unsigned)(wave * 4096 + lane * 16);
    ring.wave = wave;
    ring.nslab = NSLAB_APPLY;
    ring.prologue();

    const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)g.xh, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)g.xl, 0, g.xbytes, 0x00020000);
    const float Sf = (float)g.S;

    // All workgroups run the same instruction stream on equal tiles: left alone they stay in phase and hit HBM together
    // (every CU loading its x tiles, then every CU storing).  A one-off start delay of blockIdx % 8 eighths of a tile time
    // spreads the memory phases of the chip over the whole tile period.
    if (g.stagger > 0)
        for (int i = 0; i < (int)(blockIdx.x & 7) * g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    // dbg_stage 100: wave 0 of every workgroup records s_memtime at the stage boundaries of each of its tiles
    // (dbg[(tile * 16 + k)] as two floats per stamp: low 24 bits, next 24 bits) -- the profile behind DESIGN.md's stage table
    const bool stamp = g.dbg && g.dbg_stage == 100 && tid == 0;
#define ENC_STAMP(k)                                                                    \
    if (stamp) {                                                                        \
        const uint64_t t_ = __builtin_amdgcn_s_memtime();                               \
        g.dbg[((int64_t)tile * 16 + (k)) * 2] = (float)(t_ & 0xFFFFFF);                 \
        g.dbg[((int64_t)tile * 16 + (k)) * 2 + 1] = (float)((t_ >> 24) & 0xFFFFFF);     \
    }
    // x rows of tile `t` -> this wave's staging (row-major rows of 256 B per plane, 16-byte chunks XOR-swizzled on the source side)
    // The lane's row / chunk split for the staging traffic, recomputed from a lane id the compiler cannot hoist: as loop
    // invariants these 14 values were spilled, and every reload (scratch_load + s_waitcnt vmcnt(0)) sat between two DMA
    // requests of load_x -- each request waited for the previous one's data, 8 HBM latencies per tile (20 % of the tile time).
    auto fresh_lane = []() __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // Ksum[128] of the (at most two) sequences of a wave's 32 rows travels with the x rows: one DMA piece, lanes 0-31 the first
    // sequence, lanes 32-63 the next one -- read from the sequence image inside the tile loop it was an exposed L2 round trip
    const __amdgpu_buffer_rsrc_t rkv = __builtin_amdgcn_make_buffer_rsrc((void*)g.kvimg, 0, (unsigned)((int64_t)g.N * KVIMG), 0x00020000);
    char* s_ks = smem + SMEM_BYTES + 4 * EC * 4 + wave * 1024;
    auto load_x = [&](int t) __attribute__((always_inline)) {
        const int64_t r0 = ((int64_t)t * 4 + wave) * 32;
        const int fl = fresh_lane();
        const int trow = fl >> 4, p = fl & 15;
        {
            const int seq = min((int)(r0 / g.L) + (fl >> 5), g.N - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rkv, (lds_void*)s_ks, 16, (unsigned)((int64_t)seq * KVIMG + 16384 + (fl & 31) * 16), 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tt = 4 * i + trow;
            const int64_t rr = r0 + tt;
            const unsigned off = rr < g.M ? (unsigned)((rr * g.ldx + ((p ^ (tt & 15)) << 3)) * 2) : g.xbytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_void*)(stg + i * 1024), 16, off, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (lds_void*)(stg + 8192 + i * 1024), 16, off, 0, 0, 0);
        }
    };
    if ((int)blockIdx.x < g.ntiles) load_x(blockIdx.x);
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        ENC_STAMP(0)
        const int64_t row0 = ((int64_t)tile * 4 + wave) * 32;
        const int64_t row = row0 + tok;
        const bool valid = row < g.M;
        // ---- S0: this wave's 32 token rows were requested before the previous tile's output stores (or above, first tile)
        // sequence of this lane's token; the tile touches at most two sequences (L >= 32)
        const int n_first = (int)(row0 / g.L);
        const int64_t bound = (int64_t)(n_first + 1) * g.L;
        const int n_tok = min(row >= bound ? n_first + 1 : n_first, g.N - 1);
        const int l_tok = (int)(row - (int64_t)n_tok * g.L);
        float qm = valid ? 1.f : 0.f;
        if (g.qmask && valid) qm = (float)g.qmask[(int64_t)n_tok * g.qm_per_seq + l_tok / g.q_group];
        const bool two = __builtin_amdgcn_readfirstlane((int)(row0 + 31 >= bound && n_first + 1 < g.N)) != 0;

        wait_vmcnt<0>();        // the x tile has landed (this also waits for the slabs in flight: once per tile)
        ENC_STAMP(1)
        // B fragments of x are re-read from the staging tile where they are needed (q GEMM, every mlp.0 chunk) instead of
        // occupying 64 registers for the whole layer; the tile stays intact until the output overwrites it in place
        auto xfrags = [&](int s0, half8 (&fh)[4], half8 (&fl)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = s0 + i;
                const half4 a0 = *reinterpret_cast<const half4*>(stg + stg_off(tok, 2 * s) + 8 * half);
                const half4 a1 = *reinterpret_cast<const half4*>(stg + stg_off(tok, 2 * s + 1) + 8 * half);
                const half4 c0 = *reinterpret_cast<const half4*>(stg + 8192 + stg_off(tok, 2 * s) + 8 * half);
                const half4 c1 = *reinterpret_cast<const half4*>(stg + 8192 + stg_off(tok, 2 * s + 1) + 8 * half);
                fh[i] = half8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                fl[i] = half8{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            }
        };

        // KV^T fragments of the tile's first sequence: requested now, consumed after the q GEMM.  Fragment (b, t) is non-zero
        // only in the rows of head 2b + t (lanes with (tok >> 4) == t): the other lanes neither load nor are stored
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        half8 kvh[NKS], kvl[NKS];
        {
            const char* img = g.kvimg + (int64_t)min(n_first, g.N - 1) * KVIMG + lane * 16;
#pragma unroll
            for (int f = 0; f < NKS; ++f) {
                const bool on = (tok >> 4) == (f & 1);
                kvh[f] = on ? *reinterpret_cast<const half8*>(img + (f * 2 + 0) * 1024) : zero8;
                kvl[f] = on ? *reinterpret_cast<const half8*>(img + (f * 2 + 1) * 1024) : zero8;
            }
        }
        half8 ah[NKS], al[NKS];          // operand of the next GEMM: phi(q), then the message, then norm1(merge)
        // LayerNorm statistics of the 128 channels of this lane's token, straight from accumulator blocks (three cheap
        // passes over the accumulators instead of a 64-register copy of the row)
#define ENC_ROWSTATS(VAL, EPS, MEAN, RSTD)                                                                            \
        float MEAN, RSTD;                                                                                             \
        {                                                                                                             \
            float sum_ = 0.f;                                                                                         \
            _Pragma("unroll") for (int b = 0; b < NBLK; ++b) _Pragma("unroll") for (int r = 0; r < 16; ++r) sum_ += VAL(b, r); \
            sum_ = add_xor32(sum_);                                                                                   \
            MEAN = sum_ / (float)EC;                                                                                  \
            float sq_ = 0.f;                                                                                          \
            _Pragma("unroll") for (int b = 0; b < NBLK; ++b) _Pragma("unroll") for (int r = 0; r < 16; ++r)           \
                sq_ += (VAL(b, r) - MEAN) * (VAL(b, r) - MEAN);                                                       \
            sq_ = add_xor32(sq_);                                                                                     \
            RSTD = 1.f / sqrtf(sq_ / (float)EC + (EPS));                                                              \
        }
        float Z[2 * NBLK];
        // ---- S1: q = W_q x, phi(q), Z ------------------------------------------------------------------------------------
        {
            f32x16 am[NBLK], ax[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) am[b] = ax[b] = f32x16{0};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half8 th[4], tl[4];
                xfrags(4 * s2, th, tl);
                slab_mma<NBLK, 2>(ring, lane, am, ax, ax, th, tl);
                slab_mma<NBLK, 2>(ring, lane, am, ax, ax, th + 2, tl + 2);
            }
            ENC_STAMP(2)
            const float* ks = reinterpret_cast<const float*>(s_ks + (row >= bound ? 512 : 0));
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = am[b][r] + ax[b][r] * (1.f / 2048.f);
                if (g.dbg && g.dbg_stage == 1) dump(g, v, b, row, valid, half);
                float z0 = 0.f, z1 = 0.f;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 k4 = *reinterpret_cast<const f32x4*>(ks + 32 * b + 8 * q4 + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float f = phi_fast(v[4 * q4 + e]) * qm;
                        v[4 * q4 + e] = f;
                        if (q4 < 2) z0 += f * k4[e]; else z1 += f * k4[e];
                    }
                }
                z0 = add_xor32(z0);
                z1 = add_xor32(z1);
                Z[2 * b] = 1.f / (z0 + g.attn_eps);
                Z[2 * b + 1] = 1.f / (z1 + g.attn_eps);
                to_frags(v, ah[2 * b], al[2 * b], ah[2 * b + 1], al[2 * b + 1]);
            }
        }
        ENC_STAMP(3)
        // ---- S2: message^T = KV^T phi(q)^T per head (block-diagonal: block b holds heads 2b, 2b+1) ------------------------
        {
            f32x16 am[NBLK], ax[NBLK], ay[NBLK];      // a block's three products are consecutive here: three accumulators
#pragma unroll
            for (int b = 0; b < NBLK; ++b) am[b] = ax[b] = ay[b] = f32x16{0};
            {
                const bool mine = n_tok == min(n_first, g.N - 1);   // tokens of the other sequence contribute zero columns
#pragma unroll
                for (int f = 0; f < NKS; ++f) {
                    const half8 qh = mine ? ah[f] : zero8, ql = mine ? al[f] : zero8;
                    am[f >> 1] = mfma(kvh[f], qh, am[f >> 1]);
                    ax[f >> 1] = mfma(kvl[f], qh, ax[f >> 1]);
                    ay[f >> 1] = mfma(kvh[f], ql, ay[f >> 1]);
                }
            }
            if (two) {                                               // the tile's second sequence (uniform branch)
                const bool mine = n_tok == n_first + 1;
                const char* img = g.kvimg + (int64_t)(n_first + 1) * KVIMG + lane * 16;
#pragma unroll
                for (int f = 0; f < NKS; ++f) {
                    const bool on = (tok >> 4) == (f & 1);
                    const half8 kh = on ? *reinterpret_cast<const half8*>(img + (f * 2 + 0) * 1024) : zero8;
                    const half8 kl = on ? *reinterpret_cast<const half8*>(img + (f * 2 + 1) * 1024) : zero8;
                    const half8 qh = mine ? ah[f] : zero8, ql = mine ? al[f] : zero8;
                    am[f >> 1] = mfma(kh, qh, am[f >> 1]);
                    ax[f >> 1] = mfma(kl, qh, ax[f >> 1]);
                    ay[f >> 1] = mfma(kh, ql, ay[f >> 1]);
                }
            }
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    v[r] = ((am[b][r] + (ax[b][r] + ay[b][r]) * (1.f / 2048.f)) * Z[2 * b + (r >> 3)]) * Sf;
                if (g.dbg && g.dbg_stage == 2) dump(g, v, b, row, valid, half);
                to_frags(v, ah[2 * b], al[2 * b], ah[2 * b + 1], al[2 * b + 1]);
            }
        }
        ENC_STAMP(4)
        // ---- S3: merge, LayerNorm1 ------------------------------------------------------------------------------------------
        {
            f32x16 am[NBLK], ax[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) am[b] = ax[b] = f32x16{0};
#pragma unroll
            for (int s = 0; s < 4; ++s) slab_mma<NBLK, 2>(ring, lane, am, ax, ax, ah + 2 * s, al + 2 * s);
            ENC_STAMP(5)
#define ENC_V3(b, r) (am[b][r] + ax[b][r] * (1.f / 2048.f))
            ENC_ROWSTATS(ENC_V3, g.eps1, mean, rstd)
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                float v[16];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(s_ln + 32 * b + 8 * q4 + 4 * half);
                    const f32x4 bt = *reinterpret_cast<const f32x4*>(s_ln + EC + 32 * b + 8 * q4 + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * q4 + e] = (ENC_V3(b, 4 * q4 + e) - mean) * rstd * gm[e] + bt[e];
                }
                if (g.dbg && g.dbg_stage == 3) dump(g, v, b, row, valid, half);
                to_frags(v, ah[2 * b], al[2 * b], ah[2 * b + 1], al[2 * b + 1]);
            }
#undef ENC_V3
        }
        ENC_STAMP(6)
        // ---- S4: mlp.2(relu(mlp.0([x | m]))) in four 64-channel chunks of the hidden layer; S5: x + LayerNorm2(.) ----------
        {
            f32x16 om[NBLK], ox[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) om[b] = ox[b] = f32x16{0};
#pragma unroll 1
            for (int hc = 0; hc < 4; ++hc) {
                f32x16 hm[2], hx[2], hy[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) hm[b] = hx[b] = hy[b] = f32x16{0};
                {
                    half8 th[4], tl[4];
                    xfrags(0, th, tl);
                    slab_mma<2, 4>(ring, lane, hm, hx, hy, th, tl);          // k-steps 0-3: x channels 0-63
                    xfrags(4, th, tl);
                    slab_mma<2, 4>(ring, lane, hm, hx, hy, th, tl);          // 4-7
                }
                slab_mma<2, 4>(ring, lane, hm, hx, hy, ah, al);              // 8-11: norm1(merge) channels 0-63
                slab_mma<2, 4>(ring, lane, hm, hx, hy, ah + 4, al + 4);      // 12-15
                half8 hh[4], hl[4];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float hv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) hv[r] = fmaxf(hm[b][r] + (hx[b][r] + hy[b][r]) * (1.f / 2048.f), 0.f);
                    to_frags(hv, hh[2 * b], hl[2 * b], hh[2 * b + 1], hl[2 * b + 1]);
                }
                slab_mma<NBLK, 2>(ring, lane, om, ox, ox, hh, hl);
                slab_mma<NBLK, 2>(ring, lane, om, ox, ox, hh + 2, hl + 2);
            }
            ENC_STAMP(7)
#define ENC_V5(b, r) (om[b][r] + ox[b][r] * (1.f / 2048.f))
            ENC_ROWSTATS(ENC_V5, g.eps2, mean, rstd)
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                float v[16];
                if (g.dbg && g.dbg_stage == 4) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = ENC_V5(b, r);
                    dump(g, v, b, row, valid, half);
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(s_ln + 2 * EC + 32 * b + 8 * q4 + 4 * half);
                    const f32x4 bt = *reinterpret_cast<const f32x4*>(s_ln + 3 * EC + 32 * b + 8 * q4 + 4 * half);
                    // residual x: this lane's 4 channels of chunk 4b + q4 sit where its output goes (read, then overwritten below)
                    const half4 xrh = *reinterpret_cast<const half4*>(stg + stg_off(tok, 4 * b + q4) + 8 * half);
                    const half4 xrl = *reinterpret_cast<const half4*>(stg + 8192 + stg_off(tok, 4 * b + q4) + 8 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * q4 + e;
                        const float xr = (float)xrh[e] + (float)xrl[e] * (1.f / 2048.f);
                        v[r] = xr + ((ENC_V5(b, r) - mean) * rstd * gm[e] + bt[e]);
                    }
                }
                // split planes -> staging, in place of the x values just read (same swizzled row-major layout)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    half4 h4, l4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 a, c;
                        split_f32(v[4 * q4 + e], a, c);
                        h4[e] = a;
                        l4[e] = c;
                    }
                    const int o = stg_off(tok, 4 * b + q4) + 8 * half;
                    *reinterpret_cast<half4*>(stg + o) = h4;
                    *reinterpret_cast<half4*>(stg + 8192 + o) = l4;
                }
            }
#undef ENC_V5
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ENC_STAMP(8)
            {
                // whole-row stores; the fp32 form (last layer: features for the fine matcher, which splits them the same
                // way again) is the exact value of the planes, hi + lo / 2048.  The tile is read out of the staging first,
                // then the NEXT tile's x rows are requested into it, then the stores are issued: the load's latency runs
                // under the store issue instead of after it.
                const int fl = fresh_lane();
                const int trow = fl >> 4, p = fl & 15;
                uint4 dh[8], dl[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {       // lane p takes LOGICAL chunk p (physical p ^ (t & 15)): its store is then lane-linear
                    const int t = 4 * i + trow;
                    dh[i] = *reinterpret_cast<const uint4*>(stg + t * 256 + ((p ^ (t & 15)) << 4));
                    dl[i] = *reinterpret_cast<const uint4*>(stg + 8192 + t * 256 + ((p ^ (t & 15)) << 4));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (tile + (int)gridDim.x < g.ntiles) load_x(tile + (int)gridDim.x);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int t = 4 * i + trow;
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

// =====================================================================================================================
// enc_kv_kernel: persistent, one workgroup per CU walks the sequences; wave w takes token blocks w, w+4, ... of a sequence.
// Classic orientation: D[token][channel] = x[token][k] W[channel][k]: lane = channel, registers = 16 tokens, so that
// KV[d][d'] = sum_tokens phi(k)[token][d] v[token][d'] is an MFMA whose A and B fragments are the k / v accumulators.
// Weights: 8 slabs; slab 2 p + u = rows [k channels of head pair p | v channels of head pair p] (2 blocks) x k-steps 4u..4u+3.
// r06: the 128 KB of k|v weight fragments are RESIDENT in LDS for the whole launch (loaded once per workgroup) and the x
// fragments of a token block go global -> registers (lane = token, 16 bytes per k-step and plane), requested one block ahead.
// Before, the slabs cycled through the 4-deep ring of enc_common.h -- per token block eight barriers that kept the four waves in
// lock step, 32 LDS-DMA requests per wave, and a vmcnt(0) on the block's own x rows staged through LDS with nothing to overlap:
// a token block took ~28 K cycles for 6.9 K cycles of MFMA issue.  The arithmetic, the block -> wave assignment and the order of
// the cross-wave sums are unchanged (the image is bit-identical to the ring version's up to the 1 / S below).
// Measured on the same boxes and not kept (gpurun_out/r6q - r6w): (a) sched_group_barrier patterns "1 MFMA : 8 / 11 VALU" over the
// stream, +2 - 4 %; (b) the projection MFMAs in place by inline asm on two accumulators (main; both cross products), v unmasked, 1 / S
// on the finished sums: 1274 instead of 2302 VALU instructions per token block (900 of the 2302 are v_accvgpr moves of accumulators the
// allocator chains through copies) and NOT faster, 0.57 vs 0.55 ms -- and a trap on the way: an asm MFMA whose operand the compiler
// has just produced with a VALU instruction (v_accvgpr_read) needs two wait states the hazard recogniser cannot see (NaNs until every
// asm MFMA carried an s_nop 1); (c) fragment reads three k-steps ahead, equal; (d) diagnostic: x rows always re-read from a cache-hot
// address, -6 %: the block loop does not wait for its x rows.  s_memtime: a token block takes 16.3 K cycles (6.9 K of MFMA issue), the
// cross-wave reduction 6.6 K per sequence; wave 0's eighth block of a 900-token sequence (4 valid tokens) is 10 % of the launch.
// =====================================================================================================================
struct KvArgs {
    const _Float16 *xh, *xl;     // source rows as split planes
    int64_t ldx;
    unsigned xbytes;
    const char* wstream;         // 8 slabs
    const uint8_t* kvmask;       // [N][km_per_seq] or null
    int kv_group, km_per_seq;
    char* kvimg;                 // [N][KVIMG] out
    int S, N;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KV_AHEAD = 2, KV_NBUF = 4;       // k-steps a fragment pair is requested ahead; register ring (divides the 64-step stream)
static_assert(64 % KV_NBUF == 0 && KV_NBUF > KV_AHEAD && 2 * KV_AHEAD <= 15, "");
// LDS byte offset of the hi fragment of k-step q = 16 p + 8 b + k of a token block (p head pair, b 0 = k channels / 1 = v channels):
// slab 2 p + (k >> 2), fragment ((k & 3) * 2 + b) * 2; the lo fragment follows 1 KB later
constexpr int kv_frag_off(int q) { return (2 * (q >> 4) + ((q & 7) >> 2)) * SLAB + ((((q & 7) & 3) * 2 + ((q >> 3) & 1)) * 2) * 1024; }
// one k-step's fragment pair, requested by hand (ds_read's offset field is 16 bits: two base registers 64 KB apart)
#define KV_READ(Q)                                                                                                                      \
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                                                       \
                 : "=&v"(wfh[(Q) % KV_NBUF]), "=&v"(wfl[(Q) % KV_NBUF])                                                                  \
                 : "v"(kv_frag_off((Q) % 64) < 65536 ? wb0 : wb1), "n"(kv_frag_off((Q) % 64) & 65535), "n"((kv_frag_off((Q) % 64) & 65535) + 1024))
#define KV_WAIT(N, Q) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(wfh[(Q) % KV_NBUF]), "+v"(wfl[(Q) % KV_NBUF]) : "n"(N))

// end of block b of head pair p: k block -> phi(k) with the token mask (+ Ksum); v block -> v / S with the mask, then the pair's KV update
template <int P, int B>
__device__ __forceinline__ void kv_block_end(const f32x16& dm, const f32x16& dx, const f32x16& dy, const float (&tm)[16], float inv_S,
                                             float (&kf)[16], float (&vf)[16], float& ksum, f32x16& kvm, f32x16& kvx) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = dm[r] + (dx[r] + dy[r]) * (1.f / 2048.f);
        if (B == 0) {
            kf[r] = phi_fast(v) * tm[r];
            ksum += kf[r];
        } else {
            vf[r] = (v * tm[r]) * inv_S;
        }
    }
    if (B == 1) {
        half8 kh0, kl0, kh1, kl1, vh0, vl0, vh1, vl1;
        to_frags(kf, kh0, kl0, kh1, kl1);
        to_frags(vf, vh0, vl0, vh1, vl1);
        // KV[d][d'] += sum over the 16 + 16 tokens: A = phi(k) (lane = d), B = v (lane = d'); same token order on both
        kvm = mfma(kh0, vh0, kvm);
        kvx = mfma(kl0, vh0, kvx);
        kvm = mfma(kh1, vh1, kvm);
        kvx = mfma(kh0, vl0, kvx);
        kvx = mfma(kl1, vh1, kvx);
        kvx = mfma(kh1, vl1, kvx);
    }
}

__global__ __launch_bounds__(256) void enc_kv_kernel(KvArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    float* red = reinterpret_cast<float*>(smem + KV_W_BYTES);                 // [wave][r][lane]: one head pair's partial sums
    float* ksp = reinterpret_cast<float*>(smem + KV_W_BYTES + KV_RED);        // [wave][p][lane]
    const unsigned wb0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)lane * 16u, wb1 = wb0 + 65536u;

    {   // the weight fragments, once: 32 rounds of 4 KB (a wave moves 1 KB per request)
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)g.wstream, 0, KV_W_BYTES, 0x00020000);
#pragma unroll 4
        for (int i = 0; i < KV_W_BYTES / 4096; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(smem + i * 4096 + wave * 1024), 16,
                                                     (unsigned)(i * 4096 + wave * 1024 + lane * 16), 0, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)g.xh, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)g.xl, 0, g.xbytes, 0x00020000);
    // v / S as one multiplication by 1 / S (as encoder256.hip since r05: the IEEE division was ten VALU instructions per element of an
    // epilogue that runs beside nothing; the quotient differs by at most one rounding, inside the 2^-22 of the split it feeds)
    const float inv_S = 1.f / (float)g.S;
    const int nblocks = (g.S + 31) / 32;
    const __amdgpu_buffer_rsrc_t rmk = __builtin_amdgcn_make_buffer_rsrc((void*)g.kvmask, 0, g.kvmask ? (unsigned)g.N * (unsigned)g.km_per_seq : 0u, 0x00020000);

    // A fragments of x for token block blk of sequence n: lane = token, natural k order (16 bytes = channels 16 s + 8 half ..);
    // tokens past the sequence read zeros (out-of-range buffer offset), and so does "no next block"
    auto block_off = [&](int n, int blk) __attribute__((always_inline)) -> unsigned {
        const int s = blk * 32 + col;
        return (n < g.N && s < g.S) ? (unsigned)((((int64_t)n * g.S + s) * g.ldx + 8 * half) * 2) : g.xbytes;
    };
    // validity of token blk * 32 + (lane & 31) of sequence n: one byte load per lane and block, requested with the block's x rows (as byte
    // loads inside the block loop each of a lane's 16 entries was its own global load + wait); a ballot turns it into the block's bit mask
    auto block_mask = [&](int n, int blk) __attribute__((always_inline)) -> int {
        const int s = blk * 32 + col;
        if (!(n < g.N && s < g.S)) return 0;
        return g.kvmask ? (int)__builtin_amdgcn_raw_buffer_load_b8(rmk, (unsigned)n * (unsigned)g.km_per_seq + (unsigned)(s / g.kv_group), 0, 0) : 1;
    };
    // (every sequence has the same number of blocks: a wave without one never reads its x registers)
    u32x4 xrh[NKS], xrl[NKS];
    int mk = block_mask(blockIdx.x, wave);
    {
        const unsigned off = block_off(blockIdx.x, wave);
#pragma unroll
        for (int k = 0; k < NKS; ++k) {
            xrh[k] = __builtin_amdgcn_raw_buffer_load_b128(rxh, off, k * 32, 0);
            xrl[k] = __builtin_amdgcn_raw_buffer_load_b128(rxl, off, k * 32, 0);
        }
    }
    wait_vmcnt<2 * NKS>();                          // the weights were requested first and loads complete in order
    __syncthreads();

    for (int n = blockIdx.x; n < g.N; n += gridDim.x) {
        f32x16 kvm[NBLK], kvx[NBLK];                // KV of head pair p: rows = k channel, cols = v channel (lane)
        float ksum[NBLK];
#pragma unroll
        for (int p = 0; p < NBLK; ++p) {
            kvm[p] = kvx[p] = f32x16{0};
            ksum[p] = 0.f;
        }

        // (the first KV_AHEAD k-steps of a wave's first block of the sequence; later blocks find theirs requested by the block before)
        u32x4 wfh[KV_NBUF], wfl[KV_NBUF];
        if (wave < nblocks) {
            KV_READ(0);
            KV_READ(1);
        }
        for (int blk = wave; blk < nblocks; blk += 4) {
            // this wave's next block -- of this sequence, or its first one of the workgroup's next sequence: requested k-step by
            // k-step inside the LAST head pair, each into the registers whose fragment has just been used for the last time
            const bool more = blk + 4 < nblocks;
            const int nn = more ? n : n + (int)gridDim.x, nb = more ? blk + 4 : wave;
            const unsigned noff = block_off(nn, nb);
            // mask of the 16 tokens this lane's accumulator registers hold: token s0 + dch(r, half) (0 / 1 factors)
            const unsigned bits = (unsigned)__builtin_amdgcn_ballot_w64(mk != 0) >> (4 * half);
            mk = block_mask(nn, nb);
            float tm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) tm[r] = (float)((bits >> dch(r, 0)) & 1u);
            // The block's 64 k-steps as ONE stream q = 16 p + 8 b + k: head pair p, block b (0 = its k channels, 1 = its v channels; one
            // after the other on one accumulator triple -- both at once held 96 accumulator registers and left the epilogue of one
            // nothing to run beside), k-step k.  Its weight fragments are read KV_AHEAD k-steps ahead by hand: left to the compiler every
            // k-step was "two ds_read_b128, s_waitcnt lgkmcnt(0), three MFMAs" -- one exposed LDS latency per 96 cycles of MFMA issue.
            // (An inline-asm read is invisible to the compiler's own wait counting: every one is retired by a KV_WAIT that carries its
            // registers before they are used or die.)
            float kf[16], vf[16];
            f32x16 dm, dx, dy;
            const bool again = blk + 4 < nblocks;      // this wave has another block of this sequence: keep the fragment stream running
#define KV_STEP(Q)                                                                                                                      \
            {                                                                                                                           \
                constexpr int pq = (Q) >> 4, bq = ((Q) >> 3) & 1, kq = (Q) & 7;                                                         \
                if ((Q) + KV_AHEAD < 64) {                                                                                              \
                    KV_READ((Q) + KV_AHEAD);                                                                                            \
                    KV_WAIT(2 * KV_AHEAD, Q);                                                                                           \
                } else if (again) {                                                                                                     \
                    KV_READ((Q) + KV_AHEAD);                                                                                            \
                    KV_WAIT(2 * KV_AHEAD, Q);                                                                                           \
                } else {                                                                                                                \
                    KV_WAIT(2 * (63 - (Q)), Q);                                                                                         \
                }                                                                                                                       \
                const half8 wh = __builtin_bit_cast(half8, wfh[(Q) % KV_NBUF]), wl = __builtin_bit_cast(half8, wfl[(Q) % KV_NBUF]);     \
                const half8 xh = __builtin_bit_cast(half8, xrh[kq]), xl = __builtin_bit_cast(half8, xrl[kq]);                           \
                if (kq == 0) dm = dx = dy = f32x16{0};                                                                                  \
                dm = mfma(xh, wh, dm);                                                                                                  \
                dx = mfma(xh, wl, dx);                                                                                                  \
                dy = mfma(xl, wh, dy);                                                                                                  \
                if (pq == NBLK - 1 && bq == 1) {   /* last use of this x fragment: the next block's into the same registers */          \
                    xrh[kq] = __builtin_amdgcn_raw_buffer_load_b128(rxh, noff, kq * 32, 0);                                             \
                    xrl[kq] = __builtin_amdgcn_raw_buffer_load_b128(rxl, noff, kq * 32, 0);                                             \
                }                                                                                                                       \
                if (kq == 7) kv_block_end<pq, bq>(dm, dx, dy, tm, inv_S, kf, vf, ksum[pq], kvm[pq], kvx[pq]);                           \
            }
#define KV_STEP8(Q) KV_STEP(Q) KV_STEP((Q) + 1) KV_STEP((Q) + 2) KV_STEP((Q) + 3) KV_STEP((Q) + 4) KV_STEP((Q) + 5) KV_STEP((Q) + 6) KV_STEP((Q) + 7)
            KV_STEP8(0) KV_STEP8(8) KV_STEP8(16) KV_STEP8(24) KV_STEP8(32) KV_STEP8(40) KV_STEP8(48) KV_STEP8(56)
#undef KV_STEP8
#undef KV_STEP
        }
        // cross-wave sums, one head pair at a time through 16 KB: every wave parks its partial KV of pair p, wave p adds the four
        // in a fixed order (deterministic, independent of the batch) and writes the pair's part of the apply image
#pragma unroll
        for (int p = 0; p < NBLK; ++p) ksp[(wave * NBLK + p) * 64 + lane] = ksum[p];
        char* img = g.kvimg + (int64_t)n * KVIMG;
#pragma unroll
        for (int p = 0; p < NBLK; ++p) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = kvm[p][r] + kvx[p][r] * (1.f / 2048.f);
            __syncthreads();
            if (wave == p) {
                float kv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float a = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) a += red[(w * 16 + r) * 64 + lane];
                    kv[r] = a;
                }
                float ks = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) ks += ksp[(w * NBLK + p) * 64 + lane];
                ks = add_xor32(ks);                     // the two lane halves hold the two token halves of every block
                // apply image: fragment (b = p, t) of KV^T for enc_apply_kernel = this lane's registers 8t..8t+7, rows of the
                // other head zeroed (block-diagonal): lane = v channel d' (row of KV^T), slot j = k channel 16 t + dch(j, half)
                half8 h0, l0, h1, l1;
                to_frags(kv, h0, l0, h1, l1);
                if ((col >> 4) == 0) {                  // rows of head 2p: fragment t = 0; the other lanes' slots stay unwritten
                    *reinterpret_cast<half8*>(img + ((p * 2 + 0) * 2 + 0) * 1024 + lane * 16) = h0;  // (the consumer never loads them)
                    *reinterpret_cast<half8*>(img + ((p * 2 + 0) * 2 + 1) * 1024 + lane * 16) = l0;
                } else {                                // rows of head 2p + 1: fragment t = 1
                    *reinterpret_cast<half8*>(img + ((p * 2 + 1) * 2 + 0) * 1024 + lane * 16) = h1;
                    *reinterpret_cast<half8*>(img + ((p * 2 + 1) * 2 + 1) * 1024 + lane * 16) = l1;
                }
                if (half == 0) reinterpret_cast<float*>(img + 16384)[32 * p + col] = ks;
            }
            __syncthreads();
        }
    }
}

dfsfm::SmemAttr attr_apply, attr_kv;

}  // namespace

extern "C" int dfsfm_encoder_kv_f32(const void* src_hi, const void* src_lo, int64_t ld_src, int N, int S,
                                    const void* wstream_kv, const uint8_t* kv_mask, int kv_group, void* kv_image,
                                    void* stream_) {
    if (N == 0) return DFSFM_OK;
    if (!src_hi || !src_lo || !wstream_kv || !kv_image) return DFSFM_E_BADARG;
    if (N < 0 || S <= 0 || kv_group <= 0 || ld_src < EC) return DFSFM_E_BADARG;
    const int64_t span = ((int64_t)N * S - 1) * ld_src * 2 + EC * 2;
    if ((ld_src & 7) || span >= (int64_t)0xFFFFFFF0 || (reinterpret_cast<uintptr_t>(src_hi) & 15) ||
        (reinterpret_cast<uintptr_t>(src_lo) & 15) || (reinterpret_cast<uintptr_t>(wstream_kv) & 15) ||
        (reinterpret_cast<uintptr_t>(kv_image) & 15))
        return DFSFM_E_UNSUPPORTED;
    KvArgs g{};
    g.xh = static_cast<const _Float16*>(src_hi);
    g.xl = static_cast<const _Float16*>(src_lo);
    g.ldx = ld_src;
    g.xbytes = (unsigned)span;
    g.wstream = static_cast<const char*>(wstream_kv);
    g.kvmask = kv_mask;
    g.kv_group = kv_group;
    g.km_per_seq = (S + kv_group - 1) / kv_group;
    g.kvimg = static_cast<char*>(kv_image);
    g.S = S;
    g.N = N;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = N < cus ? N : cus;              // persistent: the weights are loaded into LDS once per workgroup
    attr_kv.ensure(reinterpret_cast<const void*>(&enc_kv_kernel), SMEM_KV);
    hipLaunchKernelGGL(enc_kv_kernel, dim3((unsigned)grid), dim3(256), SMEM_KV, static_cast<hipStream_t>(stream_), g);
    return dfsfm::check_launch("dfsfm_encoder_kv_f32");
}

extern "C" int dfsfm_encoder_apply_f32(const void* x_hi, const void* x_lo, int64_t ldx, int N, int L, int S,
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
    if (L < 32) return DFSFM_E_UNSUPPORTED;          // a 32-token tile may touch at most two sequences
    const int64_t M = (int64_t)N * L;
    const int64_t span = (M - 1) * ldx * 2 + EC * 2;
    if ((ldx & 7) || (ldo & 7) || (ldo32 & 3) || span >= (int64_t)0xFFFFFFF0) return DFSFM_E_UNSUPPORTED;
    for (const void* p : {x_hi, x_lo, wstream, kv_image, (const void*)out_hi, (const void*)out_lo, (const void*)out32,
                          (const void*)gamma1, (const void*)beta1, (const void*)gamma2, (const void*)beta2})
        if (reinterpret_cast<uintptr_t>(p) & 15) return DFSFM_E_UNSUPPORTED;
    ApplyArgs g{};
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
    g.ntiles = (int)((M + 127) / 128);
    g.dbg = debug;
    g.dbg_stage = debug_stage;
    g.stagger = g.ntiles > 4 * 256 ? 2 : 0;             // only worth it when every workgroup walks several tiles
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = g.ntiles < cus ? g.ntiles : cus;   // persistent: one workgroup per CU walks the tiles
    if ((int64_t)N * KVIMG >= (int64_t)0xFFFFFFF0) return DFSFM_E_UNSUPPORTED;      // 32-bit buffer offsets into the sequence images
    attr_apply.ensure(reinterpret_cast<const void*>(&enc_apply_kernel), SMEM_APPLY);
    hipLaunchKernelGGL(enc_apply_kernel, dim3((unsigned)grid), dim3(256), SMEM_APPLY, static_cast<hipStream_t>(stream_), g);
    return dfsfm::check_launch("dfsfm_encoder_apply_f32");
}
