// CPU lane model of the device JPEG decoder: the SAME thread functions as csrc/jpeg_decode.hip (csrc/jpeg_core.h) and the same
// host geometry (csrc/jpeg_host.h), compiled with g++; every kernel launch becomes a loop over its threads, in the launch
// order of dfsfm_jpeg_decode_u8.  `order` permutes the workgroup order of a sweep launch and the thread order inside a round (0 ascending, 1 descending, 2 a fixed
// shuffle of the workgroups): the relaxation must reach the same fixed point whatever the hardware's scheduling does.
// Test infrastructure (tests/test_jpeg_cpu.py builds it into tests/_build/); not part of the product library.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "jpeg_core.h"
#include "jpeg_host.h"

extern "C" size_t jd_emul_workspace(const dfsfm_jpeg_frame* f, int64_t scan_bytes, int out_channels) {
    jd::Params P{};
    if (!jd::derive(*f, P)) return 0;
    return jd::layout_of(P, scan_bytes, out_channels).total;
}

// returns 0, or the DFSFM_E_* code the real entry point would; sweeps_used (may be null) receives, per sweep, the chunks decoded
extern "C" int jd_emul_decode(const uint8_t* scan, int64_t scan_bytes, const dfsfm_jpeg_frame* f, const uint32_t* tab,
                              const uint16_t* qt, const uint32_t* block_base, const uint32_t* seg_beg, const uint32_t* seg_end,
                              const int32_t* seg_chunk0,
                              const int32_t* chunk_seg, uint8_t* out, int64_t out_stride, int out_channels, int sweeps,
                              int resume, int32_t* status, void* workspace, size_t workspace_bytes, int order,
                              int32_t* work_out) {
    jd::Params P{};
    if (!jd::derive(*f, P)) return DFSFM_E_UNSUPPORTED;
    if (out_stride < (int64_t)P.width * out_channels || sweeps < 1 || sweeps > 64) return DFSFM_E_BADARG;
    const jd::Layout L = jd::layout_of(P, scan_bytes, out_channels);
    if (workspace_bytes < L.total) return DFSFM_E_WORKSPACE;
    jd::bind(P, L, static_cast<char*>(workspace), scan, scan_bytes, tab, qt, block_base, seg_beg, seg_end, seg_chunk0, chunk_seg, out, out_stride,
             out_channels, status);
    std::memset(P.work, 0, 64 * 4);
    std::memset(status, 0, 16);
    std::vector<uint32_t> lds(P.tab, P.tab + jd::TAB_WORDS);  // the kernels' LDS copy of the Huffman tables
    if (!resume) {
        const int nb = (int)((scan_bytes + jd::UNSTUFF_BLOCK - 1) / jd::UNSTUFF_BLOCK);
        for (int blk = 0; blk < nb; ++blk) {                  // one workgroup of jd_unstuff_kernel
            uint32_t cnt[jd::UNSTUFF_T], grp[jd::UNSTUFF_G], keep[jd::UNSTUFF_T];
            uint64_t w0[jd::UNSTUFF_T], w1[jd::UNSTUFF_T];
            for (int t = 0; t < jd::UNSTUFF_T; ++t) {
                keep[t] = jd::unstuff_mask(P, blk, t, w0[t], w1[t]);
                cnt[t] = (uint32_t)__builtin_popcount(keep[t]);
            }
            for (int g = 0; g < jd::UNSTUFF_G; ++g) jd::unstuff_scan_b1(cnt, grp, g);
            jd::unstuff_scan_b2(grp);
            for (int t = 0; t < jd::UNSTUFF_T; ++t)
                jd::unstuff_store(P, blk, keep[t], w0[t], w1[t], cnt[t] + grp[t / (jd::UNSTUFF_T / jd::UNSTUFF_G)]);
        }
        for (int c = 0; c < P.nchunks; ++c) jd::init_thread(P, c);
    }
    // a sweep launch = every workgroup of SWEEP_WG chunks runs rounds until none of its chunks changes; within a round every thread
    // looks at its entry BEFORE any thread of the workgroup decodes (the kernel's __syncthreads_or).  The workgroups of a launch run
    // in the order `order` gives (ascending = each sees the final result of the one before it: the best case, descending = the worst)
    const int nwg = (P.nchunks + jd::SWEEP_WG - 1) / jd::SWEEP_WG;
    std::vector<int> wperm(nwg);
    for (int w = 0; w < nwg; ++w) wperm[w] = order == 1 ? nwg - 1 - w : w;
    if (order == 2) {
        uint32_t s2 = 54321;
        for (int w = nwg - 1; w > 0; --w) {
            s2 = s2 * 1664525u + 1013904223u;
            std::swap(wperm[w], wperm[(s2 >> 8) % (uint32_t)(w + 1)]);
        }
    }
    for (int s = 0; s < sweeps; ++s) {
        if (s > 0 && P.work[s - 1] == 0) break;
        for (int wi = 0; wi < nwg; ++wi) {
            const int c0 = wperm[wi] * jd::SWEEP_WG, c1 = c0 + jd::SWEEP_WG < P.nchunks ? c0 + jd::SWEEP_WG : P.nchunks;
            for (int round = 0; round < jd::SWEEP_ROUNDS; ++round) {
                std::vector<int> todo;
                std::vector<uint64_t> ent;
                for (int c = c0; c < c1; ++c) {
                    uint64_t entry = 0;
                    if (jd::sweep_needs(P, c, entry)) { todo.push_back(c); ent.push_back(entry); }
                }
                if (todo.empty()) break;
                for (size_t k = 0; k < todo.size(); ++k) {
                    const size_t kk = order == 1 ? todo.size() - 1 - k : k;
                    jd::sweep_thread(P, todo[kk], s, ent[kk], lds.data());
                }
            }
        }
    }
    status[0] = P.work[sweeps - 1];
    for (int i = 0; i < sweeps; ++i)
        if (P.work[i]) status[3] = i + 1;
    if (work_out) std::memcpy(work_out, P.work, 64 * 4);
    {
        std::vector<int32_t> part(jd::SCAN_T), grp(jd::SCAN_G);
        for (int t = 0; t < jd::SCAN_T; ++t) jd::scan_phase_a(P, t, part.data());
        for (int g = 0; g < jd::SCAN_G; ++g) jd::scan_phase_b1(part.data(), grp.data(), g);
        jd::scan_phase_b2(grp.data());
        for (int t = 0; t < jd::SCAN_T; ++t) jd::scan_phase_b3(part.data(), grp.data(), t);
        for (int t = 0; t < jd::SCAN_T; ++t) jd::scan_phase_c(P, t, part.data());
    }
    std::memset(P.coef, 0, (size_t)P.nblocks * 128);
    for (int c = 0; c < P.nchunks; ++c) jd::write_thread(P, c, lds.data());
    const int ng = jd::dc_ngroups(P);
    for (int g = 0; g < ng; ++g) jd::dc_sum_thread(P, g);
    {
        std::vector<int32_t> part(4 * jd::SCAN_T), grp(4 * jd::SCAN_G);
        for (int t = 0; t < jd::SCAN_T; ++t) jd::dc_scan_phase_a(P, t, part.data());
        for (int g = 0; g < jd::SCAN_G; ++g) jd::dc_scan_phase_b1(part.data(), grp.data(), g);
        jd::dc_scan_phase_b2(grp.data());
        for (int t = 0; t < jd::SCAN_T; ++t) jd::dc_scan_phase_b3(part.data(), grp.data(), t);
        for (int t = 0; t < jd::SCAN_T; ++t) jd::dc_scan_phase_c(P, t, part.data());
    }
    for (int g = 0; g < ng; ++g) jd::dc_apply_thread(P, g);
    for (int b = 0; b < P.nblocks; ++b) jd::idct_thread(P, b);
    if (out_channels == 3)
        for (int y = 0; y < P.height; ++y)
            for (int x = 0; x < P.width; ++x) jd::color_thread(P, x, y);
    return 0;
}

// the colour stage alone on three host planes (dfsfm_jpeg_ycc_planes_to_rgb_u8's kernel as a loop)
extern "C" int jd_emul_planes_to_rgb(const uint8_t* y, int64_t y_stride, const uint8_t* cb, const uint8_t* cr, int64_t c_stride, int width,
                                     int height, int h0, int v0, uint8_t* out, int64_t out_stride) {
    jd::Params P;
    if (!jd::planes_params(P, y, y_stride, cb, cr, c_stride, width, height, h0, v0, out, out_stride)) return -3;
    for (int yy = 0; yy < P.height; ++yy)
        for (int x = 0; x < P.width; ++x) jd::color_thread(P, x, yy);
    return 0;
}
