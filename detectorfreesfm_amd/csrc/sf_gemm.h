// Shared device code of the fp16x2-split ("sf") MFMA kernels of libdfsfm_hip.so (gfx950): operand
// descriptor, LDS tile addressing, and the activation-reuse main loop used by the convolution / linear kernels
// (conv_gemm.hip) and the coarse-correlation kernels (coarse_match.hip).
#pragma once
#include "common.h"
#include <type_traits>

namespace dfsfm_sf {

using namespace dfsfm;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;           // K slab (fp16 elements): 64-byte LDS rows
constexpr int BM2 = 256;         // M tile of the 512-thread kernels

struct ConvArgs {
    const float* x;          // fp32 NHWC input: element (n,y,x,c) at n*sxn + y*sxh + x*ldx + c
    const _Float16* xh;      // ... or the same tensor pre-split into fp16 hi / lo planes (same strides,
    const _Float16* xl;      //     in elements); x = xh + xl / 2048
    const _Float16* wh;      // [Npad][Kpad]
    const _Float16* wl;
    const float* bias;       // [Cout] or null
    const float* res;        // residual [M][ldr] fp32, or split planes resh/resl, or none
    const _Float16* resh;
    const _Float16* resl;
    float* out;              // [M][ldo] fp32 and/or split planes outh/outl [M][ldo_s] (Cout_s channels,
    _Float16* outh;          //     channels >= Cout written as zeros)
    _Float16* outl;
    int64_t ldx, sxh, sxn, ldr, ldo, ldo_s, M;
    int H, W, Cin, Ho, Wo, Cout, Cout_s, kh, kw, stride, pad, K, Kpad, relu;
    unsigned xbytes;         // byte size of one input plane (buffer-descriptor bound, v2 kernel)
    unsigned wbytes;
    unsigned ntiles;         // M tiles x N tiles (v2 kernels; the grid is padded to a multiple of 8)
    const float* lng;        // LayerNorm fused into the epilogue (Cout == N tile): gamma, beta [Cout], eps;
    const float* lnb;        //     out = residual + LN(acc + bias) * gamma + beta
    float ln_eps;
};


// byte offset of (row, 16-byte k-slot) inside a [128][32] fp16 tile; XOR swizzle makes the
// ds_read_b128 fragment reads of 16 different rows conflict-free (64-byte rows, 4 rows per bank row)
__device__ __forceinline__ int tile_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// the same tile with the row swizzle of the 16 x 16 x 32 fragment pattern (16 rows x 4 slots per read instruction)
__device__ __forceinline__ int tile_off16(int row, int slot) { return row * 64 + ((slot ^ ((row >> 1) & 3)) << 4); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Workgroup b is dispatched to XCD b % 8 (round robin).  Each XCD gets a contiguous band of tiles, so the
// tiles that read the same activation rows (the ky halo of neighbouring M tiles, the N tiles of one M tile)
// run on one XCD at about the same time and meet in its L2 instead of each fetching from HBM.
__device__ __forceinline__ unsigned xcd_band_tile(unsigned b, unsigned per_xcd) {
    return (b & 7u) * per_xcd + (b >> 3);
}


// =================================================================================================
// "same" convolutions (stride 1, pad (KW-1)/2, KW x KW taps): tap-level reuse of the activation tile.
// The conv kernel above is bound by the global->LDS operand stream: every input pixel is
// fetched once per tap.  With flattened pixel indices the rows an M tile needs for the KW taps of one
// ky are ONE contiguous run of 256 + KW - 1 pixels, so the K loop is re-ordered to (ky, 32-channel
// chunk, kx): the run is DMA'd once per (ky, chunk) into a 2-stage A ring and the kx taps read it as
// row-shifted fragments (rows whose x + kx - pad leaves the image are zeroed in registers; rows
// whose y + ky - pad leaves it are zero-filled by the DMA).  Weights keep a 3-stage ring, one slab
// per tap.  A traffic drops KW-fold; everything else (3-MFMA split product, register double-buffered
// fragments, mid-slab barrier, pinned DMA/MFMA interleave, epilogue) is as in conv_gemm_sf_kernel.
// Weights are packed tap-padded: K = (ky, kx, ceil32(Cin)).
// =================================================================================================
// WM = waves along M: 4 -> 256 x BN tile (BN 64 / 128, waves 4 x 2); 2 -> 128 x 256 tile (waves 2 x 4, 1x1 only):
// a whole 256-channel row in one workgroup, for the LayerNorm-fused linear layers of the coarse transformer;
// 8 -> 512 x 64 tile (waves 8 x 1, 3x3 only): Cout <= 64 layers (S2DNet conv1_2: 12 M pixels) with the same 64 x 64 of
// output per wave -- and therefore the same MFMAs per fragment read -- as the 256 x 128 tile (on the 256 x 64 tile a wave
// owns 64 x 32 and the matrix pipe waits for LDS: 242 vs 357 TFLOP/s-effective).
template <int BN_, int KW, int WM = 4>
struct VS {
    static_assert(WM == 4 || (WM == 2 && KW == 1) || (WM == 8 && KW == 3 && BN_ == 64), "tile shapes");
    static constexpr int PAD = KW / 2;
    static constexpr int BM = WM * 64, WN = 8 / WM;
    static constexpr int AG = WM == 8 ? 33 : WM == 4 ? 17 : 8;   // 16-row groups per A stage (BM + KW - 1 rows)
    static constexpr int APW = WM == 8 ? 9 : WM == 4 ? 5 : 2;    // A pieces per wave and super-slab
    static constexpr int A_PLANE = AG * 1024;
    static constexpr int A_STAGE = 2 * A_PLANE;             // hi, lo
    static constexpr int B_PLANE = BN_ * 64;
    static constexpr int B_STAGE = 2 * B_PLANE;
    static constexpr int BN_I = (BN_ / 16) * 2 / 8;         // weight pieces per wave per slab (2 or 1)
    static constexpr int NA = KW == 1 ? 3 : 2;              // A ring depth (super-slabs)
    static constexpr int NB = 3;                            // B ring depth (slabs); 4 measured +-1 %
    static constexpr int OFF_B = NA * A_STAGE;
    static constexpr int OFF_DUMMY = OFF_B + NB * B_STAGE;  // 1 KB sink for the padding pieces
    static constexpr int RING = OFF_DUMMY + 1024;
    static constexpr int TILE_BYTES = BM * (BN_ + 4) * 4;
    static constexpr int SMEM = RING > TILE_BYTES ? RING : TILE_BYTES;
    static_assert(RING <= 160 * 1024, "LDS ring");
    // DMA pieces a wave issues in the load segment of tap kx: B(t+NB-1), and A(S+NA-1) spread over taps 0 (3
    // pieces) and 1 (2 pieces) -- all of them at tap 0 for a 1x1 kernel, three per tap for the 512-row tile
    static constexpr int CA(int kx) { return KW == 1 ? APW : WM == 8 ? (kx < 3 ? 3 : 0) : kx == 0 ? 3 : kx == 1 ? 2 : 0; }
    static constexpr int C(int kx) { return CA(kx) + BN_I; }
    // own pieces that may still be in flight when a wave publishes slab u = (S, kx) (the barrier that opens the
    // first load segment reading it): B(u) was the last piece of load segment u-NB+1, so everything issued in the
    // NB-2 segments since may be outstanding; with a 2-deep A ring A(S) was issued at taps 0/1 of S-1.
    static constexpr int NWAIT(int kx) {
        int n = 0;
        for (int d = 1; d <= NB - 2; ++d) n += C(((kx - d) % KW + KW) % KW);
        // slab (S, 0) also needs A(S): only what was issued after its last piece may still fly -- the B pieces of that tap
        // and every piece of the later taps
        if (KW > 1 && NA == 2 && kx == 0) {
            int last = 0, after = BN_I;
            for (int k = 0; k < KW; ++k) if (CA(k) > 0) last = k;
            for (int k = last + 1; k < KW; ++k) after += C(k);
            if (after < n) n = after;
        }
        return n;
    }
    static constexpr int NWAIT0 = APW * (NA - 2) + (NB - 2) * BN_I;    // prologue: A(0), B(0) landed
    static_assert(KW == 1 ? (NA == NB) : true, "1x1: A(t) and B(t) are issued in the same load segment");
};

// M16: the same wave tile on v_mfma_f32_16x16x32_f16 (4 x 2 NJ blocks of 16 x 16, one 32-wide k-step per slab) instead of
// v_mfma_f32_32x32x16_f16 (2 x NJ blocks of 32 x 32, two k-steps).  Same LDS fragment reads (a fragment is 1 KB either way), same
// accumulator registers, twice the MFMA instructions at half the cycles each -- and, in the POWER-limited regime these kernels run
// in, 15 % more sustained MFMA rate on random operands (tools/ubench/mfma_order.hip: 1.97 vs 1.72 PFLOP/s).  The fragment of a
// 16 x 16 x 32 MFMA is 16 rows x all four 16-byte slots of a 64-byte LDS row (lane = row + 16 slot), so the tiles use the row
// swizzle (r >> 1) & 3 -- conflict-free for that pattern at every tap shift -- instead of (r >> 2) & 3.
template <int BN_, int KW, int WM, bool M16, class AccArr>
__device__ __forceinline__ void sf_same_mainloop_impl(const ConvArgs& g, char* smem, AccArr& accm, AccArr& accx, int64_t m0, int n0) {
    using S_ = VS<BN_, KW, WM>;
    constexpr int WN = 8 / WM, PAD = KW / 2, BNI = S_::BN_I, NQ = WM == 8 ? 5 : WM == 4 ? 3 : 1;
    constexpr int NI = M16 ? 4 : 2;                          // row blocks per wave (16 / 32 rows)
    constexpr int NJ = M16 ? BN_ / (16 * WN) : BN_ / (32 * WN);
    constexpr int RB = M16 ? 16 : 32;                        // rows / columns of an MFMA block
    constexpr int NFULL = NQ == 1 ? 1 : NQ - 1;                 // full 16-row groups per wave (wave + 8 q); then the tail group
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kgrp = lane >> 5;
    const int row0 = (wave / WN) * 64;                       // first tile row / column of this wave's blocks
    const int col0 = (wave % WN) * (BN_ / WN);
    const int Cin_p = g.Kpad / (KW * KW);                    // tap-padded channel count (multiple of 32)
    const int nchunk = Cin_p / BK, nS = KW * nchunk;
    // the last 32-channel chunk of a tap holds <= 16 real channels (Cin = 196: 4): its second k-step is all padding and
    // is neither read nor multiplied (a fourteenth of the 196-channel layers' MFMAs; worth 2.5 % of their time -- the
    // loop is co-limited by its load segments)
    const bool tail_half = !M16 && g.Cin - (nchunk - 1) * BK <= 16;

    const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)g.xh, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)g.xl, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc((void*)g.wh, 0, g.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)g.wl, 0, g.wbytes, 0x00020000);

    const int lrow = lane >> 2;
    const int lslot = (lane & 3) ^ (M16 ? (lane >> 3) & 3 : (lane >> 4) & 3);   // logical slot of this lane's physical slot: the row swizzle
    // A rows of this lane: groups wave + 8 q and (waves 0/1 only: hi/lo plane of) the tail group 8 * NFULL
    int ay[5];                                                   // NQ used
    int64_t abase[5];
    bool aok[5];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int grp = q < NFULL ? wave + 8 * q : 8 * NFULL;
        const int64_t pix = m0 - PAD + grp * 16 + lrow;      // flattened pixel of LDS row grp*16 + lrow
        aok[q] = pix >= 0 && pix < g.M && (q < NFULL || wave < 2);
        const int64_t pp = aok[q] ? pix : 0;
        const int ox = (int)(pp % g.W);
        const int64_t t = pp / g.W;
        ay[q] = (int)(t % g.H);
        abase[q] = (t / g.H) * g.sxn + (int64_t)ay[q] * g.sxh + (int64_t)ox * g.ldx + lslot * 8;
    }
    const bool cok_lane = true;
    (void)cok_lane;
    unsigned offA[5];                                            // NQ used
    auto addrA = [&](int Sn) __attribute__((always_inline)) {   // offsets of super-slab Sn = (ky, chunk)
        const int ky = Sn / nchunk, chunk = Sn - ky * nchunk;
        const bool in = Sn < nS && chunk * BK + lslot * 8 < g.Cin;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int iy = ay[q] + ky - PAD;
            const bool ok = aok[q] & in & (iy >= 0) & (iy < g.H);
            const int64_t off = (abase[q] + (int64_t)(ky - PAD) * g.sxh + chunk * BK) * 2;
            offA[q] = ok ? (unsigned)off : g.xbytes;
        }
    };
    const unsigned bbase = (unsigned)(((int64_t)(n0 + lrow) * g.Kpad + lslot * 8) * 2);
    unsigned offB[4];                                            // BNI used (a dependent bound here drops the host stub)
    auto addrB = [&](int t) __attribute__((always_inline)) {    // offsets of slab t = (S, kx) -> (ky, chunk, kx)
        const int Sn = t / KW, kx = t - Sn * KW;
        const int ky = Sn / nchunk, chunk = Sn - ky * nchunk;
        const unsigned koff = (unsigned)(((ky * KW + kx) * Cin_p + chunk * BK) * 2);
#pragma unroll
        for (int j = 0; j < BNI; ++j) {
            const int grp = (wave + 8 * j) % (BN_ / 16);
            offB[j] = Sn < nS ? bbase + koff + (unsigned)grp * 16u * (unsigned)g.Kpad * 2u : g.wbytes;
        }
    };
    // A piece p of a super-slab: 2q, 2q+1 = group wave + 8q (hi, lo); the last = the tail group (wave 0: hi,
    // wave 1: lo, other waves: an out-of-range piece into the 1-KB sink so every wave issues the same count)
#define SDMA_A(p, astage)                                                                                          \
    do {                                                                                                            \
        if ((p) < 2 * NFULL) {                                                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(((p) & 1) ? rxl : rxh,                                         \
                                                     (lds_void*)(smem + (astage) * S_::A_STAGE + ((p) & 1) * S_::A_PLANE + \
                                                                 (wave + 8 * ((p) >> 1)) * 1024),                   \
                                                     16, offA[(p) >> 1], 0, 0, 0);                                  \
        } else {                                                                                                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wave == 1 ? rxl : rxh,                                         \
                                                     (lds_void*)(wave < 2 ? smem + (astage) * S_::A_STAGE + wave * S_::A_PLANE + 8 * NFULL * 1024 \
                                                                          : smem + S_::OFF_DUMMY),                  \
                                                     16, offA[NFULL], 0, 0, 0);                                     \
        }                                                                                                           \
    } while (0)
#define SDMA_B(j, bstage)                                                                                          \
    do {                                                                                                            \
        const int ib_ = wave + 8 * (j);                                                                             \
        const int plane_ = ib_ / (BN_ / 16), grp_ = ib_ % (BN_ / 16);                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(plane_ ? rwl : rwh,                                                \
                                                 (lds_void*)(smem + S_::OFF_B + (bstage) * S_::B_STAGE +            \
                                                             plane_ * S_::B_PLANE + grp_ * 1024),                   \
                                                 16, offB[j], 0, 0, 0);                                             \
    } while (0)

#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            using V = std::remove_reference_t<decltype(accm[0][0])>;
            accm[i][j] = V{0};
            accx[i][j] = V{0};
        }
    // x coordinate of this lane's output rows (for the left/right border of the kx taps)
    int oxr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) oxr[i] = (int)((m0 + row0 + i * RB + (M16 ? lane & 15 : col)) % g.W);

    // ---- ping-pong schedule ------------------------------------------------------------------------------
    // Waves w and w+4 share a SIMD.  The two halves of the workgroup run the same slab sequence half a slab
    // apart: while waves 0-3 are in the COMPUTE segment of slab t (24 back-to-back MFMAs on register fragments),
    // waves 4-7 are in its LOAD segment (16 fragment reads for the whole slab, this wave's DMA pieces for slab
    // t+NB-1, address arithmetic, border masks), and vice versa, with one s_barrier per segment.  The matrix
    // pipe of every SIMD always has exactly one wave feeding it and never sees LDS / VMEM issue in its stream.
    //   barrier k:        b0    b1    b2    b3    b4
    //   waves 0-3:           L0    C0    L1    C1   ...
    //   waves 4-7:           --    L0    C0    L1   ...
    // Slab u is read in the two segments after barrier b(2u); every wave waits for its own pieces of slab u
    // (counted vmcnt) just before that barrier: waves 0-3 at the end of C(u-1), waves 4-7 at the end of L(u-1).
    const int grp = wave >> 2;
    constexpr int NKS = M16 ? 1 : 2;                                 // k-steps per slab
    half8 fah[NKS][NI], fal[NKS][NI], fbh[NKS][NJ], fbl[NKS][NJ];   // [k-step][block] fragments of one slab
    auto read_slab = [&](int astage, int shift, int bstage, bool halfc) __attribute__((always_inline)) {
        const char* sa = smem + astage * S_::A_STAGE;
        const char* sb = smem + S_::OFF_B + bstage * S_::B_STAGE;
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int off = tile_off16(row0 + i * 16 + (lane & 15) + shift, lane >> 4);
                fah[0][i] = *reinterpret_cast<const half8*>(sa + off);
                fal[0][i] = *reinterpret_cast<const half8*>(sa + S_::A_PLANE + off);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int off = tile_off16(col0 + j * 16 + (lane & 15), lane >> 4);
                fbh[0][j] = *reinterpret_cast<const half8*>(sb + off);
                fbl[0][j] = *reinterpret_cast<const half8*>(sb + S_::B_PLANE + off);
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks == 1 && halfc) break;                // wave-uniform
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int off = tile_off(row0 + i * 32 + col + shift, ks * 2 + kgrp);
                fah[ks][i] = *reinterpret_cast<const half8*>(sa + off);
                fal[ks][i] = *reinterpret_cast<const half8*>(sa + S_::A_PLANE + off);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int off = tile_off(col0 + j * 32 + col, ks * 2 + kgrp);
                fbh[ks][j] = *reinterpret_cast<const half8*>(sb + off);
                fbl[ks][j] = *reinterpret_cast<const half8*>(sb + S_::B_PLANE + off);
            }
        }
    };
    auto mask_slab = [&](int shift) __attribute__((always_inline)) {   // tap leaves the image row: contributes zero
        if (shift == PAD) return;
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool out = (unsigned)(oxr[i] + shift - PAD) >= (unsigned)g.W;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                fah[ks][i] = out ? z : fah[ks][i];
                fal[ks][i] = out ? z : fal[ks][i];
            }
        }
    };
    auto compute_slab = [&](bool halfc) __attribute__((always_inline)) {
        // 3 products per block pair; consecutive MFMAs never share an accumulator
        if constexpr (M16) {
            // accumulate IN PLACE (inline asm ties destination and addend): left to the register allocator, the 16 x 16 x 32
            // builtin got a destination different from its addend across the unrolled taps and the kernel spilled 44 registers.
            // Consecutive MFMAs never share an accumulator (reuse distance 16), so no software wait states are needed here.  The
            // compiler's hazard recogniser does not look inside the asm: the accumulators are zeroed in the prologue, a whole ring
            // fill (hundreds of cycles, with s_waitcnt + s_barrier in between) before the first MFMA reads them, and the K loop
            // ends with an explicit s_nop run before any VALU instruction may read the results (end of this function).
#define SF_MFMA16(ACC, A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A_), "v"(B_))
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) SF_MFMA16(accm[i][j], fah[0][i], fbh[0][j]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) SF_MFMA16(accx[i][j], fah[0][i], fbl[0][j]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) SF_MFMA16(accx[i][j], fal[0][i], fbh[0][j]);
#undef SF_MFMA16
        } else {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks == 1 && halfc) break;                // wave-uniform: one scalar branch per slab
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[ks][i], fbh[ks][j], accm[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[ks][i], fbl[ks][j], accx[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[ks][i], fbh[ks][j], accx[i][j], 0, 0, 0);
        }
        }
    };

    // prologue: A(0) B(0) A(1).. B(1)..  (A(0..NA-2), B(0..NB-2))
#pragma unroll
    for (int t = 0; t < (S_::NA > S_::NB ? S_::NA : S_::NB) - 1; ++t) {
        if (t < S_::NA - 1) {
            addrA(t);
            SDMA_A(0, t); SDMA_A(1, t);
            if constexpr (S_::APW >= 5) { SDMA_A(2, t); SDMA_A(3, t); SDMA_A(4, t); }
            if constexpr (S_::APW == 9) { SDMA_A(5, t); SDMA_A(6, t); SDMA_A(7, t); SDMA_A(8, t); }
        }
        if (t < S_::NB - 1) {
            addrB(t);
#pragma unroll
            for (int j = 0; j < BNI; ++j) SDMA_B(j, t);
        }
    }
    wait_vmcnt<S_::NWAIT0>();                                         // own pieces of slab 0 have landed
    __builtin_amdgcn_s_barrier();                                     // b0: slab 0 published
    if (grp == 1) __builtin_amdgcn_s_barrier();                       // waves 4-7 sit out the first segment
    int bst = 0;                                                      // B stage of the current slab (t % NB)
    int ast = 0;                                                      // A stage of the current super-slab (S % NA)
    int chunk = 0;                                                    // S % nchunk
    for (int S = 0; S < nS; ++S) {
        const bool half = tail_half && chunk == nchunk - 1;           // only k-step 0 of this super-slab's slabs is live
        chunk = chunk == nchunk - 1 ? 0 : chunk + 1;
        const int anx = ast == S_::NA - 1 ? 0 : ast + 1;              // stage of A(S+1)
        const int adm = ast == 0 ? S_::NA - 1 : ast - 1;              // stage A(S+NA-1) is DMA'd into (held A(S-1))
        auto tap = [&](auto kxc, bool halfc) __attribute__((always_inline)) {
            constexpr int kx = decltype(kxc)::value;
            constexpr int kxn = (kx + 1) % KW;                        // tap of the next slab
            const int t = S * KW + kx;
            const int bdm = bst == 0 ? S_::NB - 1 : bst - 1;          // stage of slab t+NB-1 (held slab t-1)
            // ---- LOAD segment of slab t ----
            read_slab(ast, kx, bst, halfc);
            if (kx == 0) addrA(S + S_::NA - 1);
            addrB(t + S_::NB - 1);
            if (kx == 0) { SDMA_A(0, adm); SDMA_A(1, adm); }
            if constexpr (S_::APW == 5) {
                if (kx == 0) { SDMA_A(2, adm); }
                if (KW == 1 || kx == 1) { SDMA_A(3, adm); SDMA_A(4, adm); }
            }
            if constexpr (S_::APW == 9) {                              // three pieces per tap
                if (kx == 0) { SDMA_A(2, adm); }
                if (kx == 1) { SDMA_A(3, adm); SDMA_A(4, adm); SDMA_A(5, adm); }
                if (kx == 2) { SDMA_A(6, adm); SDMA_A(7, adm); SDMA_A(8, adm); }
            }
#pragma unroll
            for (int j = 0; j < BNI; ++j) SDMA_B(j, bdm);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mask_slab(kx);
            if (grp == 1) wait_vmcnt<S_::NWAIT(kxn)>();               // waves 4-7 publish slab t+1 here
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- COMPUTE segment of slab t ----
            __builtin_amdgcn_s_setprio(1);
            compute_slab(halfc);
            __builtin_amdgcn_s_setprio(0);
            if (grp == 0) wait_vmcnt<S_::NWAIT(kxn)>();               // waves 0-3 publish slab t+1 here
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            bst = bst == S_::NB - 1 ? 0 : bst + 1;
        };
        tap(std::integral_constant<int, 0>{}, half);
        if constexpr (KW > 1) {
            tap(std::integral_constant<int, 1>{}, half);
            tap(std::integral_constant<int, 2>{}, half);
        }
        if constexpr (KW > 3) {
            tap(std::integral_constant<int, 3>{}, half);
            tap(std::integral_constant<int, 4>{}, half);
        }
        ast = anx;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                       // pairs with the extra barrier of waves 4-7
    wait_vmcnt<0>();
    // The inline-asm MFMAs are invisible to the compiler's hazard recogniser: close the region explicitly.  An XDL write of a
    // 16 x 16 x 32 MFMA (4 passes) must be 7 wait states ahead of a VALU read / overlapped write of its destination on gfx950
    // (passes + 3; tools/studies/mfma_hazard_scan.py); a barrier every wave has already reached is one issue slot, not a delay.
    // The s_nop alone is not enough: asm volatile orders memory, not registers, and the compiler hoisted the epilogue's first
    // accumulator reads ABOVE it (r05, seen in the ISA).  Every accumulator therefore passes through an empty asm as "+v" after the
    // nop: volatile asms keep their order, and a value redefined there cannot be read before it.  8 wait states per TILE.
    if constexpr (M16) {
        asm volatile("s_nop 7" ::: "memory");
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                asm volatile("" : "+v"(accm[i][j]));
                asm volatile("" : "+v"(accx[i][j]));
            }
    }
#undef SDMA_A
#undef SDMA_B
}

template <int BN_, int KW, int WM = 4>
__device__ __forceinline__ void sf_same_mainloop(const ConvArgs& g, char* smem, f32x16 (&accm)[2][BN_ * WM / 256],
                                                 f32x16 (&accx)[2][BN_ * WM / 256], int64_t m0, int n0) {
    sf_same_mainloop_impl<BN_, KW, WM, false>(g, smem, accm, accx, m0, n0);
}
// the 16 x 16 x 32 form: 4 x (BN_ WM / 128) accumulator blocks of 4 registers
template <int BN_, int KW, int WM = 4>
__device__ __forceinline__ void sf_same_mainloop16(const ConvArgs& g, char* smem, f32x4 (&accm)[4][BN_ * WM / 128],
                                                   f32x4 (&accx)[4][BN_ * WM / 128], int64_t m0, int n0) {
    sf_same_mainloop_impl<BN_, KW, WM, true>(g, smem, accm, accx, m0, n0);
}

// =================================================================================================
// Second schedule of the same product: 128 x 128 tile, 256 threads (4 waves as 2 x 2, 64 x 64 each), <= 80 KB of LDS, so
// that TWO workgroups share a CU and one's epilogue (statistics / LayerNorm / split / stores) runs under the other's main
// loop -- what linear_gemm_sf_kernel does for 1x1 layers (conv_gemm.hip), here with the activation reuse across the kx taps
// for the stride-1 "same" 3x3 convolutions and as the main loop of the 128 x 128 coarse-correlation kernel.
// One barrier per slab: [wait own pieces of slab t] [barrier] [DMA B(t+1); at kx == 0 also A(S+1)] [16 fragment reads]
// [24 MFMAs].  B is issued before A, so the pieces that may stay in flight at the next wait are exactly the A pieces.
// =================================================================================================
template <int KW>
struct V2S {
    static constexpr int PAD = KW / 2;
    static constexpr int BM = 128, BN = 128, NT = 256;
    static constexpr int AG = KW == 1 ? 8 : 9;               // 16-row groups per A stage (128 + KW - 1 <= 144)
    static constexpr int APW = KW == 1 ? 4 : 5;              // A pieces per wave and super-slab (group 8: waves 0 / 1, else a sink)
    static constexpr int A_PLANE = AG * 1024;
    static constexpr int A_STAGE = 2 * A_PLANE;
    static constexpr int NA = KW == 1 ? 3 : 2;
    static constexpr int B_PLANE = 128 * 64;
    static constexpr int B_STAGE = 2 * B_PLANE;
    static constexpr int NB = 2;
    static constexpr int OFF_B = NA * A_STAGE;
    static constexpr int OFF_DUMMY = OFF_B + NB * B_STAGE;
    static constexpr int RING = OFF_DUMMY + (KW == 1 ? 0 : 1024);   // 1 KB sink for the padding pieces of group 8
    static constexpr int TILE_BYTES = BM * (BN + 4) * 4;
    static constexpr int SMEM = RING > TILE_BYTES ? RING : TILE_BYTES;
    static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");
};

template <int KW>
__device__ __forceinline__ void sf2_mainloop(const ConvArgs& g, char* smem, f32x16 (&accm)[2][2], f32x16 (&accx)[2][2],
                                             int64_t m0, int n0) {
    using T = V2S<KW>;
    constexpr int PAD = T::PAD;
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kgrp = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int Cin_p = g.Kpad / (KW * KW);
    const int nchunk = Cin_p / BK, nS = KW * nchunk;          // super-slabs (ky, chunk); slabs t = S * KW + kx

    const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)g.xh, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)g.xl, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc((void*)g.wh, 0, g.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)g.wl, 0, g.wbytes, 0x00020000);

    const int lrow = lane >> 2;
    const int lslot = (lane & 3) ^ ((lane >> 4) & 3);
    // A rows of this lane: groups wave, wave + 4 and (KW > 1, waves 0 / 1 only: hi / lo plane of) group 8
    constexpr int NQ = KW == 1 ? 2 : 3;
    int ay[3];
    int64_t abase[3];
    bool aok[3];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int grp = q < 2 ? wave + 4 * q : 8;
        const int64_t pix = m0 - PAD + grp * 16 + lrow;
        aok[q] = pix >= 0 && pix < g.M && (q < 2 || wave < 2);
        const int64_t pp = aok[q] ? pix : 0;
        const int ox = (int)(pp % g.W);
        const int64_t t = pp / g.W;
        ay[q] = (int)(t % g.H);
        abase[q] = (t / g.H) * g.sxn + (int64_t)ay[q] * g.sxh + (int64_t)ox * g.ldx + lslot * 8;
    }
    unsigned offA[3];
    auto addrA = [&](int Sn) __attribute__((always_inline)) {
        const int ky = Sn / nchunk, chunk = Sn - ky * nchunk;
        const bool in = Sn < nS && chunk * BK + lslot * 8 < g.Cin;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int iy = ay[q] + ky - PAD;
            const bool ok = aok[q] & in & (iy >= 0) & (iy < g.H);
            const int64_t off = (abase[q] + (int64_t)(ky - PAD) * g.sxh + chunk * BK) * 2;
            offA[q] = ok ? (unsigned)off : g.xbytes;
        }
    };
    const unsigned bbase = (unsigned)(((int64_t)(n0 + lrow) * g.Kpad + lslot * 8) * 2);
    unsigned offB[2];
    auto addrB = [&](int t) __attribute__((always_inline)) {
        const int Sn = t / KW, kx = t - Sn * KW;
        const int ky = Sn / nchunk, chunk = Sn - ky * nchunk;
        const unsigned koff = (unsigned)(((ky * KW + kx) * Cin_p + chunk * BK) * 2);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            offB[q] = Sn < nS ? bbase + koff + (unsigned)(wave + 4 * q) * 16u * (unsigned)g.Kpad * 2u : g.wbytes;
    };
    auto dmaA = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            char* d = smem + stage * T::A_STAGE + (wave + 4 * q) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_void*)d, 16, offA[q], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (lds_void*)(d + T::A_PLANE), 16, offA[q], 0, 0, 0);
        }
        if constexpr (KW > 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wave == 1 ? rxl : rxh,
                                                     (lds_void*)(wave < 2 ? smem + stage * T::A_STAGE + wave * T::A_PLANE + 8 * 1024
                                                                          : smem + T::OFF_DUMMY),
                                                     16, offA[2], 0, 0, 0);
    };
    auto dmaB = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            char* d = smem + T::OFF_B + stage * T::B_STAGE + (wave + 4 * q) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, (lds_void*)d, 16, offB[q], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, (lds_void*)(d + T::B_PLANE), 16, offB[q], 0, 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            accm[i][j] = f32x16{0};
            accx[i][j] = f32x16{0};
        }
    int oxr[2];                                               // x coordinate of this lane's two output rows (kx borders)
#pragma unroll
    for (int i = 0; i < 2; ++i) oxr[i] = (int)((m0 + wr * 64 + i * 32 + col) % g.W);

    // prologue: B(0), then A(0) .. A(NA-2)
    addrB(0);
    dmaB(0);
#pragma unroll
    for (int Sn = 0; Sn < T::NA - 1; ++Sn) {
        addrA(Sn);
        dmaA(Sn);
    }
    int ast = 0, bst = 0;
    for (int S = 0; S < nS; ++S) {
        auto tap = [&](auto kxc) __attribute__((always_inline)) {
            constexpr int kx = decltype(kxc)::value;
            const int t = S * KW + kx;
            // own pieces that may still fly: the A super-slab issued after this slab's B pieces (KW == 1: A(t+1), one slab
            // ahead of its use; KW > 1: A(S+1), issued during tap 0, may still fly at tap 1)
            if constexpr (KW == 1) wait_vmcnt<(T::NA - 2) * T::APW>();
            else if constexpr (kx == 1) wait_vmcnt<T::APW>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                     // slab t published; every wave is done with slab t - 1
            __builtin_amdgcn_sched_barrier(0);
            addrB(t + 1);
            dmaB(bst ^ 1);
            if constexpr (kx == 0) {
                const int adm = ast == 0 ? T::NA - 1 : ast - 1;   // stage of A(S-1) = where A(S+NA-1) goes
                addrA(S + T::NA - 1);
                dmaA(adm);
            }
            const char* sa = smem + ast * T::A_STAGE;
            const char* sb = smem + T::OFF_B + bst * T::B_STAGE;
            half8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int off = tile_off(wr * 64 + i * 32 + col + kx, ks * 2 + kgrp);
                    ah[ks][i] = *reinterpret_cast<const half8*>(sa + off);
                    al[ks][i] = *reinterpret_cast<const half8*>(sa + T::A_PLANE + off);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int off = tile_off(wc * 64 + j * 32 + col, ks * 2 + kgrp);
                    bh[ks][j] = *reinterpret_cast<const half8*>(sb + off);
                    bl[ks][j] = *reinterpret_cast<const half8*>(sb + T::B_PLANE + off);
                }
            }
            if constexpr (KW > 1 && kx != PAD) {              // tap leaves the image row: contributes zero
                const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool out = (unsigned)(oxr[i] + kx - PAD) >= (unsigned)g.W;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        ah[ks][i] = out ? z : ah[ks][i];
                        al[ks][i] = out ? z : al[ks][i];
                    }
                }
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
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
            }
            __builtin_amdgcn_s_setprio(0);
            bst ^= 1;
        };
        tap(std::integral_constant<int, 0>{});
        if constexpr (KW > 1) {
            tap(std::integral_constant<int, 1>{});
            tap(std::integral_constant<int, 2>{});
        }
        ast = ast == T::NA - 1 ? 0 : ast + 1;
    }
    wait_vmcnt<0>();                                          // the zero-fill tail pieces, before LDS is reused
}

// =================================================================================================
// Third schedule, for the LayerNorm-fused 256-channel linears of the coarse transformer: a 160 x 256 tile, 512 threads,
// wave w owns the 32 output columns [32w, 32w + 32) of all 160 rows (5 x 1 MFMA blocks, 160 accumulator registers).
// Why 160 rows: a cross-attention layer works on 8 x 4800 = 38 400 rows -- 300 tiles of 128 rows on 256 CUs are two rounds for
// 1.17 rounds of work, 240 tiles of 160 rows are ONE round (a self layer: 480 tiles = two rounds instead of three).  Same slab
// order and the same per-accumulator MFMA order as the other schedules: identical results.
// One barrier per slab like sf2_mainloop: [wait own pieces of slab t] [barrier] [DMA B(t+1), A(t+2)] [2 x (12 reads, 15 MFMAs)].
// Per slab a wave issues 4 B pieces (weight row groups w, w + 8; hi, lo) and 3 A pieces (row group w hi, lo; waves 0-3 one
// plane of row groups 8, 9; waves 4-7 an out-of-range piece into a 1 KB sink, so that the counted wait is uniform).
// =================================================================================================
struct V160 {
    static constexpr int BM = 160, BN = 256, NT = 512;
    static constexpr int AG = 10;                            // 16-row groups per A stage
    static constexpr int APW = 3, BPW = 4;
    static constexpr int A_PLANE = AG * 1024;
    static constexpr int A_STAGE = 2 * A_PLANE;
    static constexpr int NA = 3;
    static constexpr int B_PLANE = BN * 64;
    static constexpr int B_STAGE = 2 * B_PLANE;
    static constexpr int NB = 2;
    static constexpr int OFF_B = NA * A_STAGE;
    static constexpr int OFF_DUMMY = OFF_B + NB * B_STAGE;
    static constexpr int RING = OFF_DUMMY + 1024;
    static constexpr int EPI_ROWS = 96;                      // rows per epilogue pass (96, 64)
    static constexpr int TILE_BYTES = EPI_ROWS * (BN + 4) * 4;
    static constexpr int SMEM = RING > TILE_BYTES ? RING : TILE_BYTES;
    static_assert(SMEM <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void ln160_mainloop(const ConvArgs& g, char* smem, f32x16 (&accm)[5], f32x16 (&accx)[5], int64_t m0) {
    using T = V160;
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kgrp = lane >> 5;
    const int nS = g.Kpad / BK;                               // slabs (1x1: one 32-channel chunk each)

    const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)g.xh, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)g.xl, 0, g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc((void*)g.wh, 0, g.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)g.wl, 0, g.wbytes, 0x00020000);

    const int lrow = lane >> 2;
    const int lslot = (lane & 3) ^ ((lane >> 4) & 3);
    int64_t abase[2];
    bool aok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int grp = q == 0 ? wave : 8 + (wave >> 1);
        const int64_t pix = m0 + grp * 16 + lrow;
        aok[q] = pix < g.M && (q == 0 || wave < 4);
        const int64_t pp = aok[q] ? pix : 0;
        const int ox = (int)(pp % g.W);
        const int64_t t = pp / g.W;
        abase[q] = (t / g.H) * g.sxn + (t % g.H) * g.sxh + (int64_t)ox * g.ldx + lslot * 8;
    }
    unsigned offA[2];
    auto addrA = [&](int Sn) __attribute__((always_inline)) {
        const bool in = Sn < nS && Sn * BK + lslot * 8 < g.Cin;
#pragma unroll
        for (int q = 0; q < 2; ++q) offA[q] = (aok[q] & in) ? (unsigned)((abase[q] + Sn * BK) * 2) : g.xbytes;
    };
    const unsigned bbase = (unsigned)(((int64_t)lrow * g.Kpad + lslot * 8) * 2);
    unsigned offB[2];
    auto addrB = [&](int Sn) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            offB[q] = Sn < nS ? bbase + (unsigned)(Sn * BK * 2) + (unsigned)(wave + 8 * q) * 16u * (unsigned)g.Kpad * 2u : g.wbytes;
    };
    auto dmaA_own = [&](int stage) __attribute__((always_inline)) {         // row group `wave`, both planes
        char* d = smem + stage * T::A_STAGE + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_void*)d, 16, offA[0], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (lds_void*)(d + T::A_PLANE), 16, offA[0], 0, 0, 0);
    };
    auto dmaA_extra = [&](int stage) __attribute__((always_inline)) {       // waves 0-3: one plane of row groups 8, 9; else the sink
        __builtin_amdgcn_raw_ptr_buffer_load_lds((wave & 1) ? rxl : rxh,
                                                 (lds_void*)(wave < 4 ? smem + stage * T::A_STAGE + (wave & 1) * T::A_PLANE +
                                                                            (8 + (wave >> 1)) * 1024
                                                                      : smem + T::OFF_DUMMY),
                                                 16, offA[1], 0, 0, 0);
    };
    auto dmaA = [&](int stage) __attribute__((always_inline)) {
        dmaA_own(stage);
        dmaA_extra(stage);
    };
    auto dmaB_q = [&](int stage, int q) __attribute__((always_inline)) {
        char* d = smem + T::OFF_B + stage * T::B_STAGE + (wave + 8 * q) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, (lds_void*)d, 16, offB[q], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, (lds_void*)(d + T::B_PLANE), 16, offB[q], 0, 0, 0);
    };
    auto dmaB = [&](int stage) __attribute__((always_inline)) {
        dmaB_q(stage, 0);
        dmaB_q(stage, 1);
    };
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        accm[i] = f32x16{0};
        accx[i] = f32x16{0};
    }
    // prologue: B(0), then A(0) .. A(NA-2)
    addrB(0);
    dmaB(0);
#pragma unroll
    for (int Sn = 0; Sn < T::NA - 1; ++Sn) {
        addrA(Sn);
        dmaA(Sn);
    }
    int ast = 0, bst = 0;
    for (int S = 0; S < nS; ++S) {
        wait_vmcnt<(T::NA - 2) * T::APW>();                   // B(S), A(S) landed; A(S+1) may still fly
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                         // slab S published; every wave is done with slab S - 1
        __builtin_amdgcn_sched_barrier(0);
        // the refill -- B(S+1) then A(S+2): 4 + 3 requests, in this order for the counted wait -- is spread between the slab's
        // MFMA groups instead of standing in front of its first MFMA (positions pinned by sched_barrier)
        addrB(S + 1);
        addrA(S + T::NA - 1);
        const int adm = ast == 0 ? T::NA - 1 : ast - 1;       // the stage of A(S-1)
        const char* sa = smem + ast * T::A_STAGE;
        const char* sb = smem + T::OFF_B + bst * T::B_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 ah[5], al[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int off = tile_off(i * 32 + col, ks * 2 + kgrp);
                ah[i] = *reinterpret_cast<const half8*>(sa + off);
                al[i] = *reinterpret_cast<const half8*>(sa + T::A_PLANE + off);
            }
            const int offb = tile_off(wave * 32 + col, ks * 2 + kgrp);
            const half8 bh = *reinterpret_cast<const half8*>(sb + offb);
            const half8 bl = *reinterpret_cast<const half8*>(sb + T::B_PLANE + offb);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 5; ++i) accm[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, accm[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) dmaB_q(bst ^ 1, 0); else dmaA_own(adm);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 5; ++i) accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, accx[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) dmaB_q(bst ^ 1, 1); else dmaA_extra(adm);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 5; ++i) accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, accx[i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        bst ^= 1;
        ast = ast == T::NA - 1 ? 0 : ast + 1;
    }
    wait_vmcnt<0>();                                          // the zero-fill tail pieces, before LDS is reused
}

}  // namespace dfsfm_sf
