// Baseline JPEG decode on the device -- gfx950 (MI355X).  The last host-side stage of SURVEY.md 8(f) rank 1 ("image feeding").
//
// Replaces, for baseline Huffman JPEGs (SOF0 / SOF1, 8 bit, grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0, restart markers or not):
//     cv2.imread(path, cv2.IMREAD_GRAYSCALE)        src/dataset/utils.py:127, 183   -> the luma plane
//     cv2.imread(path, cv2.IMREAD_COLOR) + BGR2RGB  src/dataset/utils.py:86-92      -> RGB
// i.e. libjpeg-turbo's default decompression path (jdhuff.c, jidctint.c, jdsample.c, jdcolor.c), byte for byte
// (oracle/jpeg_baseline.c is the sequential restatement; tests pin both to the library itself).
//
// The entropy-coded scan is ONE bit string with no index: symbol k+1 starts where symbol k ends.  What makes it parallel is
// that Huffman streams self-synchronise -- a decoder started at a wrong bit falls into step with the true one after a few
// symbols -- so the (compacted) scan is cut into fixed chunks (128 bytes by default) with one thread each:
//   unstuff   the scan is compacted once (stuffed zeros, restart markers and fill bytes out), so that bit positions are plain
//             offsets and the decoder's inner loop has no byte-level special cases
//   init      every chunk's exit state := "next chunk starts on its first byte, at the DC of block 0"
//   sweep x n (each launch: rounds inside every workgroup of 256 chunks until none of them changes, csrc/jpeg_core.h)
//             chunk c decodes from its predecessor's current exit state (position, block-in-MCU, coefficient index) and
//             publishes its own; chunks whose entry did not change since their last decode do nothing.  The first chunk
//             of a segment starts from the truth, so this is a fixed-point iteration that is exact when a sweep decodes
//             nothing, and self-synchronisation makes that happen after a few sweeps instead of nchunks
//             (Weissenberger & Schmidt's scheme for GPU Huffman / JPEG decoding, restated for 64-lane waves: no
//             intra-block phases, the relaxation runs in place in global memory, 8-byte states are single transactions).
//             Restart markers (DRI) cut the scan into independent segments: more starting points that are true.
//   scan      exclusive prefix of the blocks each chunk completes -> the block every chunk starts in
//   write     decode once more, now storing coefficients (zig-zag undone, DC as differences) into coef[block][64]
//   dc        per-component prefix sum of the DC differences in scan order (reset at restarts): 3 small launches
//   idct      one thread per 8x8 block: dequantise + jpeg_idct_islow; the luma plane goes straight to the output
//   colour    (RGB only) fancy upsampling of Cb / Cr + YCbCr -> RGB per output pixel
// All integer / byte work, and latency-bound: a thread's next lookup depends on its last.  So the chain is kept short -- one
// 32-bit window per symbol covers the code AND its value bits (one lookup, one shift), the window refills from aligned words
// of the compacted scan ONE WORD AHEAD of its use (some lane of a wave refills in almost every iteration, and a load the wave
// waits for on the spot stalls all 64), DC and AC symbols share one path (no divergence between lanes at different
// coefficients), and the Huffman tables live in 9.5 KB of LDS per workgroup (10-bit direct table + canonical search for
// the rare long codes): no vector-memory instruction in the symbol loop but that prefetch.  v1 of this file decoded the
// stuffed bytes in place at 1.5 us per byte and thread, v2 (compaction, one window per symbol) at 0.7
// (profiles/r05_jpeg_v{1,2}_kernel_stats.csv).
//
// The marker segments (tables, frame header, restart positions) are parsed on the host (detectorfreesfm_amd/jpeg.py): a few
// hundred bytes of control data.  Progressive, arithmetic-coded, 12-bit, CMYK and multi-scan files are refused there
// (DFSFM_E_UNSUPPORTED at this level), and the caller decides who decodes them.
#include "common.h"
#include "jpeg_core.h"
#include "jpeg_host.h"
#include <mutex>
#include <vector>
#include <algorithm>
#include <cstring>

namespace {

using jd::Params;
using jd::Layout;
using jd::derive;
using jd::layout_of;

// Every stage is written once as a body over (parameter block, block index) and launched in two forms: for one file (the
// parameter block is the kernel argument) and for a BATCH of files (r06: grid.y -- grid.z for the colour stage -- = file; the
// parameter blocks of up to JD_BATCH files travel as one kernel argument, indexed with the block id: scalar loads with a computed
// offset, pointers still known to be global).  A scene's files then cost one set of launches per JD_BATCH files instead of one
// per file; blocks past a file's own extent leave at once.
constexpr int JD_BATCH = 7;
struct Batch {
    Params p[JD_BATCH];
};
static_assert(sizeof(Batch) + 16 <= 4096, "kernel argument segment");

__device__ __forceinline__ void unstuff_body(const Params& P, int bx, uint32_t* cnt, uint32_t* grp) {
    if ((int64_t)bx * jd::UNSTUFF_BLOCK >= P.scan_bytes) return;          // uniform per workgroup
    uint64_t w0, w1;
    const uint32_t keep = jd::unstuff_mask(P, bx, threadIdx.x, w0, w1);
    cnt[threadIdx.x] = __popc(keep);
    __syncthreads();
    if (threadIdx.x < jd::UNSTUFF_G) jd::unstuff_scan_b1(cnt, grp, threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) jd::unstuff_scan_b2(grp);
    __syncthreads();
    jd::unstuff_store(P, bx, keep, w0, w1, cnt[threadIdx.x] + grp[threadIdx.x / (jd::UNSTUFF_T / jd::UNSTUFF_G)]);
}
__device__ __forceinline__ void init_body(const Params& P, int bx) {
    const int c = bx * 256 + threadIdx.x;
    if (c < P.nchunks) jd::init_thread(P, c);
}
__device__ __forceinline__ void sweep_body(const Params& P, int bx, int sweep, uint32_t* tab) {
    if (bx * jd::SWEEP_WG >= P.nchunks) return;
    if (sweep > 0 && P.work[sweep - 1] == 0) return;         // the launch before this one decoded nothing: the fixed point is reached
    const int c = bx * jd::SWEEP_WG + threadIdx.x;
    bool loaded = false;
    for (int round = 0; round < jd::SWEEP_ROUNDS; ++round) {
        uint64_t entry = 0;
        const bool need = c < P.nchunks && jd::sweep_needs(P, c, entry);
        if (!__syncthreads_or(need)) return;                 // a settled stretch of the scan: nothing (more) to look up
        if (!loaded) {
            for (int i = threadIdx.x; i < jd::TAB_WORDS; i += jd::SWEEP_WG) tab[i] = P.tab[i];
            loaded = true;
        }
        __syncthreads();
        if (need) jd::sweep_thread(P, c, sweep, entry, tab);
        __syncthreads();                                     // exit states are agent-scope stores / loads: the neighbours see them
    }
}
__device__ __forceinline__ void scan_body(const Params& P, int32_t* part, int32_t* grp) {
    jd::scan_phase_a(P, threadIdx.x, part);
    __syncthreads();
    if (threadIdx.x < jd::SCAN_G) jd::scan_phase_b1(part, grp, threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) jd::scan_phase_b2(grp);
    __syncthreads();
    jd::scan_phase_b3(part, grp, threadIdx.x);
    __syncthreads();
    jd::scan_phase_c(P, threadIdx.x, part);
}
__device__ __forceinline__ void write_body(const Params& P, int bx, uint32_t* tab) {
    if (bx * 256 >= P.nchunks) return;
    for (int i = threadIdx.x; i < jd::TAB_WORDS; i += 256) tab[i] = P.tab[i];
    __syncthreads();
    const int c = bx * 256 + threadIdx.x;
    if (c < P.nchunks) jd::write_thread(P, c, tab);
}
__device__ __forceinline__ void dc_scan_body(const Params& P, int32_t* part, int32_t* grp) {
    jd::dc_scan_phase_a(P, threadIdx.x, part);
    __syncthreads();
    if (threadIdx.x < jd::SCAN_G) jd::dc_scan_phase_b1(part, grp, threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) jd::dc_scan_phase_b2(grp);
    __syncthreads();
    jd::dc_scan_phase_b3(part, grp, threadIdx.x);
    __syncthreads();
    jd::dc_scan_phase_c(P, threadIdx.x, part);
}
__device__ __forceinline__ void status_body(const Params& P, int sweeps) {
    P.status[0] = P.work[sweeps - 1];
    int used = 0;
    for (int i = 0; i < sweeps; ++i)
        if (P.work[i]) used = i + 1;
    P.status[3] = used;                                          // sweeps of this call that still decoded something
}
// batch form of the three memsets: work[64] + status[4] (first launch of a call) and the coefficient array (before the write stage)
__device__ __forceinline__ void zero_body(const Params& P, int bx, int nbx, int what) {
    if (what == 0) {
        if (bx == 0 && threadIdx.x < 64) P.work[threadIdx.x] = 0;
        if (bx == 0 && threadIdx.x < 4) P.status[threadIdx.x] = 0;
        return;
    }
    uint4* c = reinterpret_cast<uint4*>(P.coef);
    const int64_t n = (int64_t)P.nblocks * 8;                    // 128 bytes per block
    for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < n; i += (int64_t)nbx * 256) c[i] = uint4{0, 0, 0, 0};
}

#define JD_FORMS(NAME, BOUNDS, SHARED, CALL1, CALLB)                                                                     \
    __global__ __launch_bounds__(BOUNDS) void NAME(Params P) { SHARED CALL1; }                                            \
    __global__ __launch_bounds__(BOUNDS) void NAME##_b(Batch B) { SHARED const Params& P = B.p[blockIdx.y]; CALLB; }

JD_FORMS(jd_unstuff_kernel, jd::UNSTUFF_T, __shared__ uint32_t cnt[jd::UNSTUFF_T]; __shared__ uint32_t grp[jd::UNSTUFF_G];,
         unstuff_body(P, blockIdx.x, cnt, grp), unstuff_body(P, blockIdx.x, cnt, grp))
JD_FORMS(jd_init_kernel, 256, , init_body(P, blockIdx.x), init_body(P, blockIdx.x))
JD_FORMS(jd_scan_kernel, jd::SCAN_T, __shared__ int32_t part[jd::SCAN_T]; __shared__ int32_t grp[jd::SCAN_G];, scan_body(P, part, grp),
         scan_body(P, part, grp))
JD_FORMS(jd_write_kernel, 256, __shared__ uint32_t tab[jd::TAB_WORDS];, write_body(P, blockIdx.x, tab), write_body(P, blockIdx.x, tab))
JD_FORMS(jd_dc_scan_kernel, jd::SCAN_T, __shared__ int32_t part[4 * jd::SCAN_T]; __shared__ int32_t grp[4 * jd::SCAN_G];,
         dc_scan_body(P, part, grp), dc_scan_body(P, part, grp))
__global__ __launch_bounds__(jd::SWEEP_WG) void jd_sweep_kernel(Params P, int sweep) {
    __shared__ uint32_t tab[jd::TAB_WORDS];
    sweep_body(P, blockIdx.x, sweep, tab);
}
__global__ __launch_bounds__(jd::SWEEP_WG) void jd_sweep_kernel_b(Batch B, int sweep) {
    __shared__ uint32_t tab[jd::TAB_WORDS];
    sweep_body(B.p[blockIdx.y], blockIdx.x, sweep, tab);
}
__global__ __launch_bounds__(256) void jd_dc_sum_kernel(Params P) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < jd::dc_ngroups(P)) jd::dc_sum_thread(P, g);
}
__global__ __launch_bounds__(256) void jd_dc_sum_kernel_b(Batch B) {
    const Params& P = B.p[blockIdx.y];
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < jd::dc_ngroups(P)) jd::dc_sum_thread(P, g);
}
__global__ __launch_bounds__(256) void jd_dc_apply_kernel(Params P) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < jd::dc_ngroups(P)) jd::dc_apply_thread(P, g);
}
__global__ __launch_bounds__(256) void jd_dc_apply_kernel_b(Batch B) {
    const Params& P = B.p[blockIdx.y];
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < jd::dc_ngroups(P)) jd::dc_apply_thread(P, g);
}
__global__ __launch_bounds__(64) void jd_idct_kernel(Params P) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b < P.nblocks) jd::idct_thread(P, b);
}
__global__ __launch_bounds__(64) void jd_idct_kernel_b(Batch B) {
    const Params& P = B.p[blockIdx.y];
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b < P.nblocks) jd::idct_thread(P, b);
}
__global__ __launch_bounds__(256) void jd_color_kernel(Params P) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < P.width && y < P.height) jd::color_thread(P, x, y);
}
__global__ __launch_bounds__(256) void jd_color_kernel_b(Batch B) {
    const Params& P = B.p[blockIdx.z];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < P.width && y < P.height) jd::color_thread(P, x, y);
}
__global__ void jd_status_kernel(Params P, int sweeps) { status_body(P, sweeps); }
__global__ void jd_status_kernel_b(Batch B, int sweeps) { status_body(B.p[blockIdx.y], sweeps); }
__global__ __launch_bounds__(256) void jd_zero_kernel_b(Batch B, int what) { zero_body(B.p[blockIdx.y], blockIdx.x, gridDim.x, what); }

}  // namespace

extern "C" size_t dfsfm_jpeg_decode_workspace(const dfsfm_jpeg_frame* frame_host, int64_t scan_bytes, int out_channels) {
    if (!frame_host || scan_bytes <= 0 || scan_bytes >= (1ll << 31) || (out_channels != 1 && out_channels != 3)) return 0;
    Params P{};
    if (!derive(*frame_host, P)) return 0;
    return layout_of(P, scan_bytes, out_channels).total;
}

extern "C" int dfsfm_jpeg_decode_u8(const uint8_t* scan, int64_t scan_bytes, const dfsfm_jpeg_frame* frame_host,
                                    const uint32_t* huff_tab, const uint16_t* qt, const uint32_t* block_base,
                                    const uint32_t* seg_beg, const uint32_t* seg_end, const int32_t* seg_chunk0, const int32_t* chunk_seg,
                                    uint8_t* out, int64_t out_stride, int out_channels, int sweeps, int resume,
                                    int32_t* status, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!scan || !frame_host || !huff_tab || !qt || !block_base || !seg_beg || !seg_end || !seg_chunk0 || !chunk_seg || !out || !status ||
        !workspace)
        return DFSFM_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(scan) & 15) != 0) return DFSFM_E_BADARG;     // the compaction pass loads aligned 16-byte pieces
    if (scan_bytes <= 0 || scan_bytes >= (1ll << 31) || (out_channels != 1 && out_channels != 3) || sweeps < 1 || sweeps > 64)
        return DFSFM_E_BADARG;
    Params P{};
    if (!derive(*frame_host, P)) return DFSFM_E_UNSUPPORTED;
    if (out_stride < (int64_t)P.width * out_channels) return DFSFM_E_BADARG;
    const Layout L = layout_of(P, scan_bytes, out_channels);
    if (workspace_bytes < L.total) return DFSFM_E_WORKSPACE;
    jd::bind(P, L, static_cast<char*>(workspace), scan, scan_bytes, huff_tab, qt, block_base, seg_beg, seg_end, seg_chunk0, chunk_seg, out, out_stride,
             out_channels, status);
    hipStream_t stream = static_cast<hipStream_t>(stream_);

    const unsigned gc = (unsigned)((P.nchunks + jd::SWEEP_WG - 1) / jd::SWEEP_WG);
    (void)hipMemsetAsync(P.work, 0, 64 * 4, stream);
    (void)hipMemsetAsync(status, 0, 4 * 4, stream);
    if (!resume) {
        const unsigned gu = (unsigned)((scan_bytes + jd::UNSTUFF_BLOCK - 1) / jd::UNSTUFF_BLOCK);
        hipLaunchKernelGGL(jd_unstuff_kernel, dim3(gu), dim3(jd::UNSTUFF_T), 0, stream, P);
        hipLaunchKernelGGL(jd_init_kernel, dim3(gc), dim3(256), 0, stream, P);
    }
    for (int s = 0; s < sweeps; ++s) hipLaunchKernelGGL(jd_sweep_kernel, dim3(gc), dim3(jd::SWEEP_WG), 0, stream, P, s);
    hipLaunchKernelGGL(jd_status_kernel, dim3(1), dim3(1), 0, stream, P, sweeps);
    hipLaunchKernelGGL(jd_scan_kernel, dim3(1), dim3(jd::SCAN_T), 0, stream, P);
    (void)hipMemsetAsync(P.coef, 0, (size_t)P.nblocks * 128, stream);
    hipLaunchKernelGGL(jd_write_kernel, dim3(gc), dim3(256), 0, stream, P);
    const unsigned gg = (unsigned)((jd::dc_ngroups(P) + 255) / 256);
    hipLaunchKernelGGL(jd_dc_sum_kernel, dim3(gg), dim3(256), 0, stream, P);
    hipLaunchKernelGGL(jd_dc_scan_kernel, dim3(1), dim3(jd::SCAN_T), 0, stream, P);
    hipLaunchKernelGGL(jd_dc_apply_kernel, dim3(gg), dim3(256), 0, stream, P);
    hipLaunchKernelGGL(jd_idct_kernel, dim3((unsigned)((P.nblocks + 63) / 64)), dim3(64), 0, stream, P);
    if (out_channels == 3)
        hipLaunchKernelGGL(jd_color_kernel, dim3((unsigned)((P.width + 63) / 64), (unsigned)((P.height + 3) / 4)), dim3(256), 0,
                           stream, P);
    return dfsfm::check_launch("dfsfm_jpeg_decode_u8");
}

namespace {
constexpr int JD_SIDE = 4;
struct SideStreams {
    hipStream_t s[JD_SIDE];
    hipEvent_t fork, join[JD_SIDE];
};
// one pool per device, created on first use and kept for the life of the process (a handful of handles)
SideStreams* side_streams() {
    static SideStreams pool[64];
    static bool made[64] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!made[dev]) {
        SideStreams& p = pool[dev];
        for (int k = 0; k < JD_SIDE; ++k) {
            if (hipStreamCreateWithFlags(&p.s[k], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&p.join[k], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        made[dev] = true;
    }
    return &pool[dev];
}
}  // namespace

extern "C" int dfsfm_jpeg_decode_batch_u8(const dfsfm_jpeg_job* jobs_host, int n_jobs, int out_channels, int sweeps, int resume,
                                          void* stream_) {
    if (!jobs_host || n_jobs < 0) return DFSFM_E_BADARG;
    if ((out_channels != 1 && out_channels != 3) || sweeps < 1 || sweeps > 64) return DFSFM_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // every job is checked before anything is launched
    for (int i = 0; i < n_jobs; ++i) {
        const dfsfm_jpeg_job& j = jobs_host[i];
        if (!j.scan || !j.frame_host || !j.huff_tab || !j.qt || !j.block_base || !j.seg_beg || !j.seg_end || !j.seg_chunk0 || !j.chunk_seg ||
            !j.out || !j.status || !j.workspace)
            return DFSFM_E_BADARG;
        if ((reinterpret_cast<uintptr_t>(j.scan) & 15) != 0 || j.scan_bytes <= 0 || j.scan_bytes >= (1ll << 31)) return DFSFM_E_BADARG;
        Params P{};
        if (!derive(*j.frame_host, P)) return DFSFM_E_UNSUPPORTED;
        if (j.out_stride < (int64_t)P.width * out_channels) return DFSFM_E_BADARG;
        if (j.workspace_bytes < layout_of(P, j.scan_bytes, out_channels).total) return DFSFM_E_WORKSPACE;
    }
    // Groups of JD_BATCH files are independent chains of small, latency-bound launches (a group's sweep launch is ~180 workgroups on
    // 256 CUs): they run on up to JD_SIDE internal streams that fork from `stream` and join it again, so that several groups share
    // the chip (measured: one group at a time 0.19 ms per 1600 x 1200 file on the device, profiles/r06_jpeg_batch.txt).
    const int ngroups = (n_jobs + JD_BATCH - 1) / JD_BATCH;
    SideStreams* side = ngroups > 1 ? side_streams() : nullptr;
    if (side) {
        (void)hipEventRecord(side->fork, stream);
        for (int k = 0; k < JD_SIDE && k < ngroups; ++k) (void)hipStreamWaitEvent(side->s[k], side->fork, 0);
    }
    for (int i0 = 0, grp = 0; i0 < n_jobs; i0 += JD_BATCH, ++grp) {
        const int nb = n_jobs - i0 < JD_BATCH ? n_jobs - i0 : JD_BATCH;
        hipStream_t gs = side ? side->s[grp % JD_SIDE] : stream;
        Batch B{};
        unsigned gu = 1, gc = 1, gg = 1, gi = 1, gz = 1, gcx = 1, gcy = 1;        // grid extents: the largest of the group
        for (int k = 0; k < nb; ++k) {
            const dfsfm_jpeg_job& j = jobs_host[i0 + k];
            Params& P = B.p[k];
            derive(*j.frame_host, P);
            const Layout L = layout_of(P, j.scan_bytes, out_channels);
            jd::bind(P, L, static_cast<char*>(j.workspace), j.scan, j.scan_bytes, j.huff_tab, j.qt, j.block_base, j.seg_beg, j.seg_end, j.seg_chunk0,
                     j.chunk_seg, j.out, j.out_stride, out_channels, j.status);
            auto up = [](unsigned& g, int64_t v) { if (v > (int64_t)g) g = (unsigned)v; };
            up(gu, (j.scan_bytes + jd::UNSTUFF_BLOCK - 1) / jd::UNSTUFF_BLOCK);
            up(gc, (P.nchunks + jd::SWEEP_WG - 1) / jd::SWEEP_WG);
            up(gg, (jd::dc_ngroups(P) + 255) / 256);
            up(gi, (P.nblocks + 63) / 64);
            up(gz, ((int64_t)P.nblocks * 8 + 256 * 8 - 1) / (256 * 8));
            up(gcx, (P.width + 63) / 64);
            up(gcy, (P.height + 3) / 4);
        }
        for (int k = nb; k < JD_BATCH; ++k) B.p[k] = B.p[0];                       // never indexed: grid.y = nb
        const unsigned ny = (unsigned)nb;
        hipLaunchKernelGGL(jd_zero_kernel_b, dim3(1, ny), dim3(256), 0, gs, B, 0);
        if (!resume) {
            hipLaunchKernelGGL(jd_unstuff_kernel_b, dim3(gu, ny), dim3(jd::UNSTUFF_T), 0, gs, B);
            hipLaunchKernelGGL(jd_init_kernel_b, dim3(gc, ny), dim3(256), 0, gs, B);
        }
        for (int s = 0; s < sweeps; ++s) hipLaunchKernelGGL(jd_sweep_kernel_b, dim3(gc, ny), dim3(jd::SWEEP_WG), 0, gs, B, s);
        hipLaunchKernelGGL(jd_status_kernel_b, dim3(1, ny), dim3(1), 0, gs, B, sweeps);
        hipLaunchKernelGGL(jd_scan_kernel_b, dim3(1, ny), dim3(jd::SCAN_T), 0, gs, B);
        hipLaunchKernelGGL(jd_zero_kernel_b, dim3(gz, ny), dim3(256), 0, gs, B, 1);
        hipLaunchKernelGGL(jd_write_kernel_b, dim3(gc, ny), dim3(256), 0, gs, B);
        hipLaunchKernelGGL(jd_dc_sum_kernel_b, dim3(gg, ny), dim3(256), 0, gs, B);
        hipLaunchKernelGGL(jd_dc_scan_kernel_b, dim3(1, ny), dim3(jd::SCAN_T), 0, gs, B);
        hipLaunchKernelGGL(jd_dc_apply_kernel_b, dim3(gg, ny), dim3(256), 0, gs, B);
        hipLaunchKernelGGL(jd_idct_kernel_b, dim3(gi, ny), dim3(64), 0, gs, B);
        if (out_channels == 3) hipLaunchKernelGGL(jd_color_kernel_b, dim3(gcx, gcy, ny), dim3(256), 0, gs, B);
    }
    if (side) {
        for (int k = 0; k < JD_SIDE && k < ngroups; ++k) {
            (void)hipEventRecord(side->join[k], side->s[k]);
            (void)hipStreamWaitEvent(stream, side->join[k], 0);
        }
    }
    return dfsfm::check_launch("dfsfm_jpeg_decode_batch_u8");
}

// ---- the colour stage alone (r06): multi-scan sequential files ---------------------------------------------------------------
// A sequential file may carry each component in a scan of its own (T.81 A.2.2; one block per MCU over the component's own block
// grid).  jpeg.plan_components turns such a file into three grey frames of the components' real samples, each decoded by
// dfsfm_jpeg_decode_u8; this is what is left: libjpeg-turbo's upsampling (jdsample.c) + ycc_rgb_convert (jdcolor.c) per output pixel.
extern "C" int dfsfm_jpeg_ycc_planes_to_rgb_u8(const uint8_t* y, int64_t y_stride, const uint8_t* cb, const uint8_t* cr, int64_t c_stride,
                                               int width, int height, int h0, int v0, uint8_t* out, int64_t out_stride, void* stream_) {
    if (!y || !cb || !cr || !out || out_stride < 3 * (int64_t)width) return DFSFM_E_BADARG;
    jd::Params P;
    if (!jd::planes_params(P, y, y_stride, cb, cr, c_stride, width, height, h0, v0, out, out_stride)) return DFSFM_E_UNSUPPORTED;
    hipLaunchKernelGGL(jd_color_kernel, dim3((unsigned)((P.width + 63) / 64), (unsigned)((P.height + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), P);
    return dfsfm::check_launch("dfsfm_jpeg_ycc_planes_to_rgb_u8");
}

// ---- host side: the index of a scan (r06) -------------------------------------------------------------------------------
// What jpeg.plan needs to know about the entropy-coded bytes before anything is uploaded: where the scan ends (the first marker that
// is not RSTn), where the restart markers are, and how many bytes the device's compaction pass keeps in front of every 4096-byte
// block / restart interval (kept = not a stuffed zero, not the FF of a marker or fill byte, not a restart marker's code).  The
// numpy formulation of the same rules (jpeg._scan_index_numpy, the specification the tests hold this to) costs 0.33 ms per 0.84-MB
// file and holds the interpreter lock for part of it; this is one memchr walk over the ~1/256 of the bytes that are 0xFF, outside
// the lock (ctypes), so that helper threads really parse in parallel.  Plain host code: no device is touched.
extern "C" int64_t dfsfm_jpeg_scan_index(const uint8_t* scan, int64_t n_avail, uint32_t* block_base, int64_t block_cap, uint32_t* seg_beg,
                                         uint32_t* seg_end, int64_t seg_cap, int64_t* n_rst_out) {
    if (!scan || n_avail < 0 || !block_base || !seg_beg || !seg_end || !n_rst_out || block_cap < 0 || seg_cap < 1) return DFSFM_E_BADARG;
    if (n_avail >= (1ll << 31)) return DFSFM_E_UNSUPPORTED;
    std::vector<int64_t> drops, rst;
    int64_t scan_len = n_avail;
    for (int64_t p = 0; p < n_avail;) {
        const void* hit = memchr(scan + p, 0xFF, (size_t)(n_avail - p));
        if (!hit) break;
        p = static_cast<const uint8_t*>(hit) - scan;
        const unsigned c = p + 1 < n_avail ? scan[p + 1] : 0xD9u;        // what follows the data is a marker
        if (c == 0) {
            drops.push_back(p + 1);                                        // the stuffed zero behind a data FF
            p += 2;
        } else if (c == 0xFF) {
            drops.push_back(p);                                            // a fill byte; the next FF is looked at on its own
            p += 1;
        } else if (c >= 0xD0 && c <= 0xD7) {
            drops.push_back(p);
            drops.push_back(p + 1);                                        // RSTn: both bytes
            rst.push_back(p);
            p += 2;
        } else {
            scan_len = p;                                                  // any other marker ends the scan
            break;
        }
    }
    // an FF in the last byte of the scan is followed by the end marker: dropped as a marker byte, never a restart marker (the
    // walk above classified it from the byte that follows, which IS that marker's FF or nothing: same result)
    while (!rst.empty() && rst.back() + 1 >= scan_len) {                   // (cannot happen: an RSTn pair lies inside the scan)
        rst.pop_back();
    }
    *n_rst_out = (int64_t)rst.size();
    const int64_t nblocks = (scan_len + jd::UNSTUFF_BLOCK - 1) / jd::UNSTUFF_BLOCK;
    if (nblocks > block_cap || (int64_t)rst.size() + 1 > seg_cap) return scan_len;      // the caller sees the counts and raises
    size_t k = 0;
    for (int64_t b = 0; b < nblocks; ++b) {                                // kept bytes in front of block b = start - #drops before it
        const int64_t x = b * jd::UNSTUFF_BLOCK;
        while (k < drops.size() && drops[k] < x) ++k;
        block_base[b] = (uint32_t)(x - (int64_t)k);
    }
    auto clean = [&](int64_t x) {                                          // x is increasing over the calls below per array
        const size_t lo = std::lower_bound(drops.begin(), drops.end(), x) - drops.begin();
        return (uint32_t)(x - (int64_t)lo);
    };
    seg_beg[0] = clean(0);
    for (size_t i = 0; i < rst.size(); ++i) {
        seg_end[i] = clean(rst[i]);
        seg_beg[i + 1] = clean(rst[i] + 2);
    }
    seg_end[rst.size()] = clean(scan_len);
    return scan_len;
}

