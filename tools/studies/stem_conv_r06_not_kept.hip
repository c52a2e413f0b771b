// K6, the stem -- ResNetFPN_8_2.conv1 (7 x 7, stride 2, pad 3, 1 -> 128 channels) + folded eval BatchNorm + ReLU for gfx950 (MI355X):
//   third_party/LoFTR/src/loftr/backbone/resnet_fpn.py:100-108  (self.conv1 / self.bn1 / self.relu on the grey frame)
// as ONE launch that reads the fp32 frame and writes the fp16 hi / lo split planes the next convolution DMAs.
//
// Why a kernel of its own (r06).  On the implicit-GEMM kernel (conv_gemm_kernel<false>: K = 49 flattened taps, every A element a
// scalar gather through a (ky, kx) table) the layer took 0.29 ms per 16 frames of 480 x 640 for 0.65 GB of traffic -- 2.2 TB/s: the
// gather and the 128 x 128 tile's epilogue, not the bytes, set the time.  Here a workgroup stages the (2 R + 5) x (2 C + 5) input
// pixels of an R x C = 4 x 64 output tile ONCE in LDS as fp16 hi / lo planes (7 KB), the weights sit in LDS as ready-made MFMA
// fragments for the whole launch (32 KB, persistent workgroups), and the convolution runs TRANSPOSED on v_mfma_f32_32x32x16_f16:
//   D[channel][pixel] = sum_k W[channel][k] * patch[k][pixel],   k = ky * 7 + kx  (49 taps, zero-padded to 64 = 4 k-steps)
// so that a lane owns ONE output pixel (lane & 31) and 16 channels of a 32-channel block per accumulator -- four runs of four
// consecutive channels, i.e. 8-byte stores into the pixel's 256-byte row of each plane, no staging tile, no cross-lane traffic.
// The B fragment (8 taps of a lane's pixel per k-step) is gathered from the LDS patch with 2-byte reads at compile-time offsets.  Arithmetic as everywhere in this library: fp16x2 split operands, three MFMA products (hi hi, lo hi, hi lo), fp32
// accumulation; bias (folded BN) and ReLU in fp32; outputs split by split_f32.
#include "common.h"
#include <cstdlib>

namespace {

using namespace dfsfm;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int KH = 7, TAPS = 49, KSTEPS = 4;          // 49 taps in 4 k-steps of 16
constexpr int COUT = 128, NBLK = COUT / 32;
constexpr int TR = 4, TC = 64;                        // output tile of a workgroup: one row per wave, two 32-pixel runs
constexpr int PR = 2 * TR + 5, PC = 2 * TC + 5;       // input patch 13 x 133
constexpr int PW = PC + 1;                            // patch row stride in halves (even)
constexpr int PLANE = PR * PW;                        // halves per plane
constexpr int W_BYTES = KSTEPS * NBLK * 2 * 1024;     // weight fragments [k-step][block][hi, lo][64 lanes][8 halves]
constexpr int OFF_PATCH = W_BYTES;                    // two planes of PLANE halves (+ one zero half for the padding taps)
constexpr int OFF_BIAS = OFF_PATCH + ((2 * PLANE + 1) * 2 + 15) / 16 * 16;
constexpr int SMEM = OFF_BIAS + COUT * 4;

struct StemArgs {
    const float* x;              // [N][H][W] fp32 (row stride sxh, image stride sxn, in floats)
    int64_t sxn, sxh;
    int N, H, W, Ho, Wo;
    const char* wfrag;           // W_BYTES
    const float* bias;           // [128] or null
    int relu;
    _Float16 *oh, *ol;           // [N * Ho * Wo][ldo] split planes
    int64_t ldo;
    int tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ f32x16 mfma(const half8 a, const half8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void stem7x7s2_kernel(StemArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, half = lane >> 5;
    _Float16* patch = reinterpret_cast<_Float16*>(smem + OFF_PATCH);
    float* s_bias = reinterpret_cast<float*>(smem + OFF_BIAS);
    // weights and bias once per workgroup
    for (int i = tid; i < W_BYTES / 16; i += 256)
        reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(g.wfrag)[i];
    if (tid < COUT) s_bias[tid] = g.bias ? g.bias[tid] : 0.f;
    if (tid == 0) patch[2 * PLANE] = (_Float16)0.f;                       // what the padding taps 49..63 read (their weights are 0)
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int n = tile / (g.tiles_x * g.tiles_y), r = tile - n * g.tiles_x * g.tiles_y;
        const int oy0 = (r / g.tiles_x) * TR, ox0 = (r % g.tiles_x) * TC;
        __syncthreads();                                                  // the previous tile's readers are done (first tile: weights landed)
        // ---- the tile's input patch, split into hi / lo planes (zeros outside the frame = the convolution's padding) ----
        {
            const float* img = g.x + (int64_t)n * g.sxn;
            const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
            for (int i = tid; i < PR * PC; i += 256) {
                const int ry = i / PC, rx = i - ry * PC;
                const int iy = iy0 + ry, ix = ix0 + rx;
                float v = 0.f;
                if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) v = img[(int64_t)iy * g.sxh + ix];
                _Float16 h, l;
                split_f32(v, h, l);
                patch[ry * PW + rx] = h;
                patch[PLANE + ry * PW + rx] = l;
            }
        }
        __syncthreads();
        const int oy = oy0 + wave;
#pragma unroll 1
        for (int run = 0; run < TC / 32; ++run) {
            const int ox = ox0 + run * 32 + px;
            const int pbase = (2 * wave) * PW + 2 * (run * 32 + px);      // patch position of tap (0, 0) of this lane's pixel
            f32x16 am[NBLK], ax[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) am[b] = ax[b] = f32x16{0};
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                asm volatile("" ::: "memory");      // keeps the LDS reads of later k-steps out of this one's registers (128 accumulators are live)
                // B fragment: the lane's 8 taps of this k-step, both planes (2-byte LDS reads, packed by hand)
                // tap t = 16 s + 8 half + j sits (t / 7) * PW + t % 7 halves behind the pixel's tap (0, 0); taps 49..63 (zero weights) read
                // the zero slot.  Both halves' offsets are compile-time constants: one select per read instead of 32 live registers
                unsigned short th[8], tl[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t0 = 16 * s + j, t1 = t0 + 8;
                    const int o0 = t0 < TAPS ? pbase + (t0 / KH) * PW + t0 % KH : 2 * PLANE;
                    const int o1 = t1 < TAPS ? pbase + (t1 / KH) * PW + t1 % KH : 2 * PLANE;
                    const int l0 = t0 < TAPS ? o0 + PLANE : 2 * PLANE, l1 = t1 < TAPS ? o1 + PLANE : 2 * PLANE;
                    th[j] = reinterpret_cast<const unsigned short*>(patch)[half ? o1 : o0];
                    tl[j] = reinterpret_cast<const unsigned short*>(patch)[half ? l1 : l0];
                }
                unsigned ph[4], pl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ph[i] = (unsigned)th[2 * i] | ((unsigned)th[2 * i + 1] << 16);
                    pl[i] = (unsigned)tl[2 * i] | ((unsigned)tl[2 * i + 1] << 16);
                }
                half8 bh, bl;
                __builtin_memcpy(&bh, ph, 16);
                __builtin_memcpy(&bl, pl, 16);
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    const half8 wh = *reinterpret_cast<const half8*>(smem + ((s * NBLK + b) * 2 + 0) * 1024 + lane * 16);
                    const half8 wl = *reinterpret_cast<const half8*>(smem + ((s * NBLK + b) * 2 + 1) * 1024 + lane * 16);
                    am[b] = mfma(wh, bh, am[b]);
                    ax[b] = mfma(wl, bh, ax[b]);
                    ax[b] = mfma(wh, bl, ax[b]);
                }
            }
            // ---- epilogue: bias (folded BN), ReLU, split; the lane's pixel row gets four 8-byte pieces per block and plane ----
            if (oy < g.Ho && ox < g.Wo) {
                const int64_t row = ((int64_t)n * g.Ho + oy) * g.Wo + ox;
                _Float16* rh = g.oh + row * g.ldo;
                _Float16* rl = g.ol + row * g.ldo;
#pragma unroll
                for (int b = 0; b < NBLK; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c0 = 32 * b + 8 * q + 4 * half;
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(s_bias + c0);
                        half4 h4, l4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = am[b][4 * q + e] + ax[b][4 * q + e] * (1.f / 2048.f) + bv[e];
                            if (g.relu) v = fmaxf(v, 0.f);
                            _Float16 h, l;
                            split_f32(v, h, l);
                            h4[e] = h;
                            l4[e] = l;
                        }
                        *reinterpret_cast<half4*>(rh + c0) = h4;
                        *reinterpret_cast<half4*>(rl + c0) = l4;
                    }
            }
        }
    }
}

}  // namespace

// x: fp32 frames [N][H][W] (one channel; sxn / sxh = image / row strides in floats); w_frag: the 32-KB fragment image of the folded
// 7 x 7 weights (ops.StemWeights); bias[128] or NULL; out_hi / out_lo: split planes [N * Ho * Wo][ldo] halves, Ho = (H - 1) / 2 + 1.
extern "C" int dfsfm_stem7x7s2_f32(const float* x, int64_t sxn, int64_t sxh, int N, int H, int W, const void* w_frag, const float* bias,
                                   int relu, void* out_hi, void* out_lo, int64_t ldo, void* stream_) {
    if (N == 0) return DFSFM_OK;
    if (!x || !w_frag || !out_hi || !out_lo) return DFSFM_E_BADARG;
    if (N < 0 || H <= 0 || W <= 0 || sxh < W || ldo < COUT) return DFSFM_E_BADARG;
    if ((ldo & 3) || (reinterpret_cast<uintptr_t>(out_hi) & 7) || (reinterpret_cast<uintptr_t>(out_lo) & 7) ||
        (reinterpret_cast<uintptr_t>(w_frag) & 15) || H > 32768 || W > 32768)
        return DFSFM_E_UNSUPPORTED;
    StemArgs g{};
    g.x = x; g.sxn = sxn; g.sxh = sxh;
    g.N = N; g.H = H; g.W = W;
    g.Ho = (H - 1) / 2 + 1;
    g.Wo = (W - 1) / 2 + 1;
    g.wfrag = static_cast<const char*>(w_frag);
    g.bias = bias;
    g.relu = relu;
    g.oh = static_cast<_Float16*>(out_hi);
    g.ol = static_cast<_Float16*>(out_lo);
    g.ldo = ldo;
    g.tiles_x = (g.Wo + TC - 1) / TC;
    g.tiles_y = (g.Ho + TR - 1) / TR;
    const int64_t nt = (int64_t)N * g.tiles_x * g.tiles_y;
    if (nt > 0x7fffffff) return DFSFM_E_UNSUPPORTED;
    g.ntiles = (int)nt;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = g.ntiles < 2 * cus ? g.ntiles : 2 * cus;          // persistent: the weight fragments are loaded once per workgroup
    hipLaunchKernelGGL(stem7x7s2_kernel, dim3((unsigned)grid), dim3(256), SMEM, static_cast<hipStream_t>(stream_), g);
    return dfsfm::check_launch("dfsfm_stem7x7s2_f32");
}
