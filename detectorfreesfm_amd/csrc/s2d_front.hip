// K9 front end -- S2DNet conv1_1 -> ReLU -> conv1_2 -> ReLU -> { centre crop, MaxPool2d(3, 2, 1) } in ONE launch, gfx950 (MI355X).
//
// Replaces, for the 35 x 35 RGB patches of the refinement head,
//   VGG16 features[0..4] with the substituted pooling layer   src/MultiviewMatcher/backbone/S2DNet/s2dnet.py:86-92, 127-160
//   (conv1_1 3->64, ReLU, conv1_2 64->64, ReLU, MaxPool2d(3, stride 2, padding 1))                 backbone/S2DNet/vggnet.py:12-44
// and hands out exactly the two things the rest of the network reads of relu1_2: the centre window that adaptation layer 0 is
// evaluated on (s2dnet.py:164-175; refine.py evaluates it only where the W x W output needs it) and the pooled 18 x 18 map that
// conv2_1 reads.  As three launches (direct_conv, conv_gemm_sf_same<64,3,8>, maxpool) this front end moved 13 GB per 10 000
// patches -- conv1_1 wrote the 3.1 GB fp16x2 map that conv1_2 read back, conv1_2 wrote another 3.1 GB of which the pool re-read
// all and kept a quarter -- and took 5.6 of the 34 ms refinement step (profiles/r05_refine_step_kernel_stats.csv) at a third of
// the HBM rate and a third of the matrix rate at once.  Here neither map exists in HBM: 14.7 KB of RGB in, 175 KB of split
// planes out per patch (1.9 GB per 10 000).
//
// One workgroup (4 waves) per patch, TWO workgroups per CU (78.9 KB of LDS each), a patch in five bands of seven output rows:
//   rgb     the band's 11 input rows (zero-padded to 37 columns, (r, g, b, 0) per pixel) -> LDS as fp16 hi / lo planes
//   conv1_1 for one 32-channel half of the band's 9 relu1_1 rows, as ONE more MFMA k-step (K = 27 taps padded to 32) of the same
//           fp16x2-split product: the weights are the A operand (host-built fragments), a pixel's im2col row is gathered from the
//           rgb planes in LDS two bytes at a time; bias, ReLU, split
//           (v = hi + lo/2048) and an 8-byte store per plane into [9 x 37 rows][32 ch] -- zero rows / columns where conv1_2 pads
//           (r06 first version: 27-tap FMA chains in fp32 on the VALU -- half of the kernel's VALU instructions, which do not hide
//           under the other workgroup's MFMAs)
//   conv1_2 9 taps x that half = 9 slabs of K = 32 on v_mfma_f32_16x16x32_f16 (three products per slab: hi*hi, hi*lo, lo*hi):
//           a wave owns 64 of the band's 245 output pixels x 64 channels (4 x 4 blocks); a tap is an LDS ROW OFFSET of the
//           A fragment (ky * 37 + kx: the zero columns make every tap a plain shift, no masks); the weights stream L2 -> LDS
//           through a 3-stage ring of 8-KB slabs (buffer_load ... lds, counted vmcnt, one barrier per slab)
//   ... the other half, same accumulators ...
//   out     accumulators -> fp32 staging tile (over the dead operand buffers) -> bias, ReLU, fp16x2 split -> the rows of the
//           crop window and the pooled rows this band completes; a pooled row that straddles two bands is carried as a 4.6-KB
//           partial maximum (max commutes with + bias, ReLU and the split, all monotone: pooled bytes equal pool-after-split)
// Two workgroups per CU: 3.24 ms per 10 000 patches against 4.6 - 4.9 ms with one (and 4.8 - 5.0 ms for the three launches).  What
// bounds it (profiles/r06_s2d_front.txt: timing-only ablations at one / two workgroups per CU, SQ counters, a residency census): the
// matrix pipe is busy 38 % of the kernel and VALU instructions issue 50 % of it, and the two barely overlap -- fp32 VALU work of a
// wave runs at HALF rate while its SIMD partner issues back-to-back MFMAs (tools/ubench/mfma_valu_overlap.hip: x2.0 - 2.2 for
// v_fma / v_pk_fma, x1.15 - 1.3 for integer work, the MFMAs unaffected), so a second workgroup hides latency, not VALU time; wave
// priority and a start skew between the two workgroups of a CU measured +-1 %.  786 M VALU instructions per launch for 173 M MFMAs:
// half of them were conv1_1's FMA chains -- now an MFMA k-step (above).
// MFMA work per patch: 5 bands x 256 rows x 64 x 576 x 2 x 3 = 283 MFLOP for 271 MFLOP of split product (245 of 256 rows live).
#include "common.h"
#include "sf_gemm.h"

namespace {

using namespace dfsfm;
using dfsfm_sf::half8;
using dfsfm_sf::tile_off16;
using dfsfm_sf::wait_vmcnt;

constexpr int P = 35;                             // patch side
constexpr int PW = P + 2;                         // band row in LDS: one zero column on each side
constexpr int R = 7;                              // output rows per band
constexpr int NBAND = P / R;
constexpr int LR = R + 2;                         // relu1_1 rows of a band
constexpr int AROWS = 336;                        // LR * PW = 333 rows, rounded
constexpr int MROWS = R * P;                      // 245 output pixels per band
constexpr int A_PLANE = AROWS * 64;               // [row][32 channels] fp16
constexpr int B_PLANE = 64 * 64;                  // [cout][32 k] fp16
constexpr int B_STAGE = 2 * B_PLANE;
constexpr int NBST = 3;
constexpr int OFF_B = 2 * A_PLANE;
constexpr int RING = OFF_B + NBST * B_STAGE;      // 67 584
constexpr int TILE_LD = 68;                       // fp32 staging row (floats)
constexpr int TILE_BYTES = 256 * TILE_LD * 4;     // 69 632: all 256 rows of the MFMA tile, so that staging needs no row test
constexpr int UNION = RING > TILE_BYTES ? RING : TILE_BYTES;
constexpr int RGB_ROWS = R + 4;                   // input rows 7 b - 2 .. 7 b + 8
constexpr int OFF_RGB = UNION;
constexpr int RGB_PLANE = RGB_ROWS * PW * 8;      // fp16 (r, g, b, 0) per pixel
constexpr int RGB_BYTES = (2 * RGB_PLANE + 15) & ~15;
constexpr int PO = (P + 1) / 2;                   // pooled side: 18
constexpr int OFF_CARRY = OFF_RGB + RGB_BYTES;
constexpr int CARRY_BYTES = PO * 64 * 4;
constexpr int SMEM = OFF_CARRY + CARRY_BYTES;     // 80 752
constexpr int NSLAB = 18;                         // 2 channel halves x 9 taps
static_assert(LR * PW <= AROWS && 2 * SMEM <= 160 * 1024 && P % R == 0, "two workgroups per CU");

struct FrontArgs {
    const float* x;                               // [n][P][P][3] normalised RGB patches (NHWC)
    const _Float16* w1f;                          // conv1_1 weights as MFMA A fragments [2 halves][2 x 16 channels][hi, lo][64 lanes][8]
    const float* b1;                              // [64]
    const _Float16 *w2h, *w2l;                    // conv1_2 weights, tap-padded split planes [>= 64][kpad], k = tap * 64 + ci
    const float* b2;                              // [64]
    _Float16 *crh, *crl;                          // centre window of relu1_2: [n][c1 - c0][c1 - c0][64] split planes
    _Float16 *poh, *pol;                          // pooled relu1_2: [n][18][18][64] split planes
    int c0, c1, kpad;
    unsigned w2bytes;
};

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(256, 2) void s2d_front_kernel(FrontArgs g) {
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t patch = blockIdx.x;
    const float* xp = g.x + patch * (P * P * 3);
    dfsfm_half4* rgb_h = reinterpret_cast<dfsfm_half4*>(smem + OFF_RGB);                 // [RGB_ROWS * PW] pixels x (r, g, b, 0) fp16, hi plane
    dfsfm_half4* rgb_l = rgb_h + RGB_ROWS * PW;                                          // lo plane
    float* carry = reinterpret_cast<float*>(smem + OFF_CARRY);
    float* tile = reinterpret_cast<float*>(smem);
    const int CW = g.c1 - g.c0;

    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc((void*)g.w2h, 0, g.w2bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)g.w2l, 0, g.w2bytes, 0x00020000);

    // ---- lane constants of the MFMA loop -------------------------------------------------------------------------------------
    int arow[4];                                   // LDS row of tap (0, 0) for this lane's pixel of row block i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = wave * 64 + i * 16 + (lane & 15);
        const int mm = m < MROWS ? m : 0;          // the 11 pad rows of the tile multiply pixel 0; nobody reads their results
        const int y = mm / P;
        arow[i] = y * PW + (mm - y * P);
    }
    const int kslot = lane >> 4;
    // weight slab DMA: a wave's piece is 16 couts x 64 B of one plane; lane -> (row lane >> 2, physical slot lane & 3)
    const unsigned bbase = (unsigned)(((wave * 16 + (lane >> 2)) * g.kpad + (((lane & 3) ^ ((lane >> 3) & 3)) * 8)) * 2);
    auto dma_b = [&](int t) __attribute__((always_inline)) {          // slab t = half * 9 + tap -> stage t % 3
        const int hf = t >= 9 ? 1 : 0, tap = t - 9 * hf;
        const unsigned off = t < NSLAB ? bbase + (unsigned)((tap * 64 + hf * 32) * 2) : g.w2bytes;   // past the end: zero fill
        char* d = smem + OFF_B + (t % NBST) * B_STAGE + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, (lds_void*)d, 16, off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, (lds_void*)(d + B_PLANE), 16, off, 0, 0, 0);
    };

    // ---- conv1_1 as ONE MFMA k-step per product (K = 27 taps -> 32), transposed: relu1_1^T[channel][pixel] = W1[channel][k] X^T[k][pixel].
    // A operand = the weights of 16 channels (host-built fragments, lane = channel + 16 kslot), B operand = the im2col row of a pixel
    // (lane = pixel + 16 kslot: k = 8 kslot .. + 7, k = 3 tap + ci), gathered from the band's rgb planes in LDS two bytes at a time and
    // packed in pairs.  A result block has
    // lane = pixel, registers = 4 consecutive channels -> one 8-byte store per plane into the A planes of conv1_2.
    // Wave w takes the 16-pixel blocks w, w + 4, ... of the band's 9 x 35 = 315 pixels (20 blocks).
    unsigned koff[8];                              // byte offset of k = 8 kslot + j inside an rgb plane, relative to the pixel's tap (0, 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * kslot + j;
        const int tap = k < 27 ? k / 3 : 0, ci = k < 27 ? k - 3 * tap : 3;          // k >= 27: the zero fourth element of a pixel
        const int ky = tap / 3, kx = tap - 3 * ky;
        koff[j] = (unsigned)(((ky * PW + kx) * 4 + ci) * 2);
    }
    float bias2[8];                                // every output item of this thread has c8 = tid & 7 (items are strided by 256)
#pragma unroll
    for (int q = 0; q < 8; ++q) bias2[q] = g.b2[(tid & 7) * 8 + q];
    for (int band = 0; band < NBAND; ++band) {
        const int y0 = band * R;
        // ---- rgb rows y0 - 2 .. y0 + 8, columns -1 .. 35 ---------------------------------------------------------------------
        for (int e = tid; e < RGB_ROWS * PW; e += 256) {
            const int ry = e / PW, cx = e - ry * PW;
            const int iy = y0 - 2 + ry, ix = cx - 1;
            const bool ok = iy >= 0 && iy < P && ix >= 0 && ix < P;
            const float* p = xp + (ok ? (iy * P + ix) * 3 : 0);
            const float r0 = ok ? p[0] : 0.f, r1 = ok ? p[1] : 0.f, r2 = ok ? p[2] : 0.f;
            dfsfm_half4 h = {0, 0, 0, 0}, l = {0, 0, 0, 0};
            _Float16 a, b;
            split_f32(r0, a, b); h[0] = a; l[0] = b;
            split_f32(r1, a, b); h[1] = a; l[1] = b;
            split_f32(r2, a, b); h[2] = a; l[2] = b;
            rgb_h[e] = h;
            rgb_l[e] = l;
        }
        wait_vmcnt<0>();                            // also the previous band's output stores: the counted waits below see only DMA
        dma_b(0);
        dma_b(1);
        lds_barrier();

        f32x4 accm[4][4], accx[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accm[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }

#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {
            // ---- conv1_1, channels hf * 32 .. + 31 of the band's 9 rows (three MFMAs per 16 channels x 16 pixels) -------------------------
            {
                const half8* wf = reinterpret_cast<const half8*>(g.w1f) + hf * 4 * 64 + lane;       // [half][cb][plane][lane] fragments
                const half8 wh0 = wf[0], wl0 = wf[64], wh1 = wf[128], wl1 = wf[192];
                const f32x4 bv0 = *reinterpret_cast<const f32x4*>(g.b1 + hf * 32 + 4 * kslot);
                const f32x4 bv1 = *reinterpret_cast<const f32x4*>(g.b1 + hf * 32 + 16 + 4 * kslot);
                const unsigned rgb_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + OFF_RGB);   // LDS byte address of the hi plane
#pragma unroll 1                 // a rolled loop: one block's fragments and accumulators live at a time
                for (int it = 0; it < 5; ++it) {
                    const int q = (wave + 4 * it) * 16 + (lane & 15);          // this lane's pixel of the 9 x 35 band
                    const bool pix_ok = q < LR * P;
                    const int qq = pix_ok ? q : 0;
                    const int ly = qq / P;
                    const int pix_rb = ly * PW + (qq - ly * P);                // rgb pixel index of tap (0, 0); A row = pix_rb + 1
                    // im2col fragment of this lane's pixel: 8 fp16 of each plane, two bytes per LDS read, into register halves
                    // (ds_read_u16_d16 / _d16_hi into register halves would save the packing, but with SRAM ECC on -- this part -- a d16 load
                    // clears the other half of its destination: measured, the low halves came back 0)
                    unsigned th[8], tl[8];
                    const unsigned pa = rgb_base + (unsigned)pix_rb * 8u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned ad = pa + koff[j];
                        asm volatile("ds_read_u16 %0, %1" : "=v"(th[j]) : "v"(ad));
                        asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(tl[j]) : "v"(ad), "n"(RGB_PLANE));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(th[0]), "+v"(th[1]), "+v"(th[2]), "+v"(th[3]), "+v"(th[4]), "+v"(th[5]), "+v"(th[6]),
                                 "+v"(th[7]), "+v"(tl[0]), "+v"(tl[1]), "+v"(tl[2]), "+v"(tl[3]), "+v"(tl[4]), "+v"(tl[5]), "+v"(tl[6]), "+v"(tl[7])::"memory");
                    unsigned xh[4], xl[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        xh[i] = th[2 * i] | (th[2 * i + 1] << 16);
                        xl[i] = tl[2 * i] | (tl[2 * i + 1] << 16);
                    }
                    half8 bh, bl;
                    __builtin_memcpy(&bh, xh, 16);
                    __builtin_memcpy(&bl, xl, 16);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, bh, z, 0, 0, 0);
                    const f32x4 m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, bh, z, 0, 0, 0);
                    f32x4 x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, bl, z, 0, 0, 0);
                    f32x4 x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, bl, z, 0, 0, 0);
                    x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0, bh, x0, 0, 0, 0);
                    x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1, bh, x1, 0, 0, 0);
                    const int iy = y0 - 1 + ly;
                    const bool live = iy >= 0 && iy < P;            // rows outside the image are conv1_2's zero padding
                    dfsfm_half4 h0, l0, h1, l1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        _Float16 a, b;
                        split_f32(live ? fmaxf(m0[r] + x0[r] * (1.f / 2048.f) + bv0[r], 0.f) : 0.f, a, b);
                        h0[r] = a; l0[r] = b;
                        split_f32(live ? fmaxf(m1[r] + x1[r] * (1.f / 2048.f) + bv1[r], 0.f) : 0.f, a, b);
                        h1[r] = a; l1[r] = b;
                    }
                    if (pix_ok) {                  // channels 4 kslot .. + 3 of block cb: logical slot 2 cb + (kslot >> 1), half (kslot & 1)
                        const int row = pix_rb + 1;
                        const int o0 = tile_off16(row, kslot >> 1) + (kslot & 1) * 8, o1 = tile_off16(row, 2 + (kslot >> 1)) + (kslot & 1) * 8;
                        *reinterpret_cast<dfsfm_half4*>(smem + o0) = h0;
                        *reinterpret_cast<dfsfm_half4*>(smem + A_PLANE + o0) = l0;
                        *reinterpret_cast<dfsfm_half4*>(smem + o1) = h1;
                        *reinterpret_cast<dfsfm_half4*>(smem + A_PLANE + o1) = l1;
                    }
                }
                if (tid < 2 * LR * 4) {                             // the zero columns x' = 0 and x' = 36 of every band row
                    const int ly = tid >> 3, side = (tid >> 2) & 1;
                    const int off = tile_off16(ly * PW + side * (PW - 1), tid & 3);
                    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    *reinterpret_cast<half8*>(smem + off) = z;
                    *reinterpret_cast<half8*>(smem + A_PLANE + off) = z;
                }
            }
            // ---- conv1_2: 9 taps of this half ------------------------------------------------------------------------------------
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int t = hf * 9 + tap;
                wait_vmcnt<2>();                    // own pieces of slab t landed (slab t + 1 may still fly)
                lds_barrier();                      // slab t published, every wave is done with slab t - 1; tap 0: the A planes written
                dma_b(t + 2);
                const int ky = tap / 3, kx = tap - 3 * ky;
                const char* sb = smem + OFF_B + (t % NBST) * B_STAGE;
                // fragment reads in the order the MFMA groups need them (A hi, B hi | B lo | A lo): the compiler's counted lgkmcnt waits
                // let the first 16 MFMAs start when half of the reads have landed
                half8 ah[4], al[4], bh[4], bl[4];
                int aoff[4], boff[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) aoff[i] = tile_off16(arow[i] + ky * PW + kx, kslot);
#pragma unroll
                for (int j = 0; j < 4; ++j) boff[j] = tile_off16(j * 16 + (lane & 15), kslot);
#pragma unroll
                for (int i = 0; i < 4; ++i) ah[i] = *reinterpret_cast<const half8*>(smem + aoff[i]);
#pragma unroll
                for (int j = 0; j < 4; ++j) bh[j] = *reinterpret_cast<const half8*>(sb + boff[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) bl[j] = *reinterpret_cast<const half8*>(sb + B_PLANE + boff[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) al[i] = *reinterpret_cast<const half8*>(smem + A_PLANE + aoff[i]);
                __builtin_amdgcn_sched_barrier(0);
                // in place (destination tied to the addend, see sf_gemm.h); consecutive MFMAs never share an accumulator
#define S2D_MFMA(ACC, A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A_), "v"(B_))
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) S2D_MFMA(accm[i][j], ah[i], bh[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) S2D_MFMA(accx[i][j], ah[i], bl[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) S2D_MFMA(accx[i][j], al[i], bh[j]);
#undef S2D_MFMA
            }
            lds_barrier();                          // every wave has read its last A fragments of this half
        }
        wait_vmcnt<0>();                            // the two zero-fill tail slabs, before the ring becomes the staging tile
        lds_barrier();
        // the inline-asm MFMAs are invisible to the hazard recogniser: 16 x 16 x 32 results need passes + 3 = 7 wait states before
        // a VALU read (sf_gemm.h); the accumulators pass through an empty asm so that no read is hoisted above the nop
        asm volatile("s_nop 7" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("" : "+v"(accm[i][j]));
                asm volatile("" : "+v"(accx[i][j]));
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wave * 64 + i * 16 + 4 * (lane >> 4) + r;
                    tile[row * TILE_LD + j * 16 + (lane & 15)] = accm[i][j][r] + accx[i][j][r] * (1.f / 2048.f);
                }
        lds_barrier();

        // ---- outputs of the band: 8 channels per thread ------------------------------------------------------------------------
        auto finish = [&](const float (&v)[8], int c8, _Float16* oh, _Float16* ol, int64_t o) __attribute__((always_inline)) {
            half8 h, l;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                _Float16 a, b;
                split_f32(fmaxf(v[q] + bias2[q], 0.f), a, b);
                h[q] = a;
                l[q] = b;
            }
            *reinterpret_cast<half8*>(oh + o) = h;
            *reinterpret_cast<half8*>(ol + o) = l;
        };
        {   // centre window rows of this band
            const int ya = y0 > g.c0 ? y0 : g.c0, yb = y0 + R < g.c1 ? y0 + R : g.c1;
            const int n = yb > ya ? (yb - ya) * CW * 8 : 0;
            for (int e = tid; e < n; e += 256) {
                const int c8 = e & 7, px = e >> 3;
                const int yy = px / CW, xx = px - yy * CW;
                const int y = ya + yy, x = g.c0 + xx;
                const float* tp = tile + ((y - y0) * P + x) * TILE_LD + c8 * 8;
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(tp), t1 = *reinterpret_cast<const f32x4*>(tp + 4);
                const float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                finish(v, c8, g.crh, g.crl, ((patch * CW + (y - g.c0)) * CW + xx) * 64 + c8 * 8);
            }
        }
        // pooled rows: j covers rows 2j-1 .. 2j+1 (inside the image); it is finished by the band that holds its last row
        const int ylast = y0 + R - 1;
        const int jlo = y0 / 2;                                     // first pooled row with a row in this band (2j + 1 >= y0)
        const int jhi = (ylast + 1) / 2 < PO - 1 ? (ylast + 1) / 2 : PO - 1;
        auto pooled = [&](int j, int jx, int c8, bool with_carry, float (&v)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = -INFINITY;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int y = 2 * j + dy;
                if (y < y0 || y > ylast || y >= P) continue;        // uniform per (j, band) except through j: cheap either way
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int x = 2 * jx + dx;
                    if (x < 0 || x >= P) continue;
                    const float* tp = tile + ((y - y0) * P + x) * TILE_LD + c8 * 8;
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(tp), t1 = *reinterpret_cast<const f32x4*>(tp + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[q] = fmaxf(v[q], t0[q]);
                        v[4 + q] = fmaxf(v[4 + q], t1[q]);
                    }
                }
            }
            if (with_carry) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], carry[jx * 64 + c8 * 8 + q]);
            }
        };
        {
            const int n = (jhi - jlo + 1) * PO * 8;
            for (int e = tid; e < n; e += 256) {
                const int c8 = e & 7, px = e >> 3;
                const int jj = px / PO, jx = px - jj * PO;
                const int j = jlo + jj;
                const int last = 2 * j + 1 < P ? 2 * j + 1 : P - 1;
                if (last > ylast) continue;                         // finished by the next band (its partial maximum: below)
                float v[8];
                pooled(j, jx, c8, 2 * j - 1 < y0 && y0 > 0, v);
                finish(v, c8, g.poh, g.pol, ((patch * PO + j) * PO + jx) * 64 + c8 * 8);
            }
        }
        lds_barrier();                              // the old partial maxima have been read
        {
            const int last = 2 * jhi + 1 < P ? 2 * jhi + 1 : P - 1;
            if (last > ylast) {                     // uniform: jhi straddles into the next band
                for (int e = tid; e < PO * 8; e += 256) {
                    const int c8 = e & 7, jx = e >> 3;
                    float v[8];
                    pooled(jhi, jx, c8, false, v);
#pragma unroll
                    for (int q = 0; q < 8; ++q) carry[jx * 64 + c8 * 8 + q] = v[q];
                }
            }
        }
        lds_barrier();                              // tile and rgb are free for the next band
    }
}

}  // namespace

extern "C" int dfsfm_s2d_front_f32(const float* patches, int64_t n_patches, int patch, const void* w1_frag, const float* b1,
                                   const void* w2_hi, const void* w2_lo, int64_t w2_rows, int64_t kpad, const float* b2, int c0,
                                   int c1, void* crop_hi, void* crop_lo, void* pool_hi, void* pool_lo, void* stream_) {
    if (!patches || !w1_frag || !b1 || !w2_hi || !w2_lo || !b2 || !crop_hi || !crop_lo || !pool_hi || !pool_lo) return DFSFM_E_BADARG;
    if (n_patches < 0 || c0 < 0 || c1 <= c0 || c1 > patch || w2_rows < 64) return DFSFM_E_BADARG;
    if (n_patches == 0) return DFSFM_OK;
    if (patch != P || kpad != 9 * 64 || n_patches > 0x7fffffff) return DFSFM_E_UNSUPPORTED;   // callers keep the three-launch path
    const uintptr_t al = reinterpret_cast<uintptr_t>(w2_hi) | reinterpret_cast<uintptr_t>(w2_lo) | reinterpret_cast<uintptr_t>(crop_hi) |
                         reinterpret_cast<uintptr_t>(crop_lo) | reinterpret_cast<uintptr_t>(pool_hi) | reinterpret_cast<uintptr_t>(pool_lo);
    if ((al & 15) || (reinterpret_cast<uintptr_t>(w1_frag) & 15) || (reinterpret_cast<uintptr_t>(b1) & 15) ||
        (reinterpret_cast<uintptr_t>(patches) & 3))
        return DFSFM_E_UNSUPPORTED;
    FrontArgs g{};
    g.x = patches; g.w1f = static_cast<const _Float16*>(w1_frag); g.b1 = b1;
    g.w2h = static_cast<const _Float16*>(w2_hi); g.w2l = static_cast<const _Float16*>(w2_lo); g.b2 = b2;
    g.crh = static_cast<_Float16*>(crop_hi); g.crl = static_cast<_Float16*>(crop_lo);
    g.poh = static_cast<_Float16*>(pool_hi); g.pol = static_cast<_Float16*>(pool_lo);
    g.c0 = c0; g.c1 = c1; g.kpad = (int)kpad;
    g.w2bytes = (unsigned)(64 * kpad * 2);
    static dfsfm::SmemAttr attr;
    attr.ensure(reinterpret_cast<const void*>(&s2d_front_kernel), SMEM);
    hipLaunchKernelGGL(s2d_front_kernel, dim3((unsigned)n_patches), dim3(256), SMEM, static_cast<hipStream_t>(stream_), g);
    return dfsfm::check_launch("dfsfm_s2d_front_f32");
}
