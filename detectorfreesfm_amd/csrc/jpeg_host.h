// Host-side geometry of the device JPEG decoder: what the frame header implies (derive), where the workspace arrays live
// (layout_of) and the binding of both into the kernels' parameter block.  Plain C++: shared by csrc/jpeg_decode.hip and by the
// CPU lane model (tests/jpeg_emul.cpp).
#pragma once
#include <stddef.h>
#include "dfsfm_hip.h"
#include "jpeg_core.h"

namespace jd {

struct Layout {
    size_t clean, exit_state, last_entry, nblk, blk0, work, coef, dc_part, dc_base, plane[3], total;
};

// geometry the frame header implies; false = not a frame this decoder takes
inline bool derive(const dfsfm_jpeg_frame& f, Params& P) {
    if (f.width <= 0 || f.height <= 0 || f.width > 65535 || f.height > 65535) return false;
    if (f.ncomp != 1 && f.ncomp != 3) return false;
    P.ncomp = f.ncomp;
    P.width = f.width;
    P.height = f.height;
    int hmax = 1, vmax = 1;
    if (f.ncomp == 1) {
        P.comp_h[0] = P.comp_v[0] = 1;                          // a single-component scan is never interleaved
    } else {
        for (int c = 0; c < 3; ++c) { P.comp_h[c] = f.h[c]; P.comp_v[c] = f.v[c]; }
        if (f.h[1] != 1 || f.v[1] != 1 || f.h[2] != 1 || f.v[2] != 1) return false;
        if (!((f.h[0] == 1 && f.v[0] == 1) || (f.h[0] == 2 && f.v[0] == 1) || (f.h[0] == 2 && f.v[0] == 2) ||
              (f.h[0] == 1 && f.v[0] == 2) || (f.h[0] == 4 && f.v[0] == 1)))       // 4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1
            return false;
        hmax = f.h[0];
        vmax = f.v[0];
    }
    P.hmax = hmax;
    P.vmax = vmax;
    P.mcux = (f.width + 8 * hmax - 1) / (8 * hmax);
    const int mcuy = (f.height + 8 * vmax - 1) / (8 * vmax);
    P.nmcu = P.mcux * mcuy;
    int j = 0;
    for (int c = 0; c < f.ncomp; ++c) {
        if (f.dc_slot[c] < 0 || f.dc_slot[c] > 3 || f.ac_slot[c] < 0 || f.ac_slot[c] > 3) return false;
        P.comp_j0[c] = j;
        for (int by = 0; by < P.comp_v[c]; ++by)
            for (int bx = 0; bx < P.comp_h[c]; ++bx, ++j) {
                P.blk_comp[j] = c; P.blk_bx[j] = bx; P.blk_by[j] = by;
                P.blk_dc[j] = f.dc_slot[c]; P.blk_ac[j] = f.ac_slot[c];
            }
        P.plane_w[c] = P.mcux * P.comp_h[c] * 8;
        P.plane_h[c] = mcuy * P.comp_v[c] * 8;
        P.real_w[c] = (f.width * P.comp_h[c] + hmax - 1) / hmax;
        P.real_h[c] = (f.height * P.comp_v[c] + vmax - 1) / vmax;
    }
    P.nb = j;
    P.dc_pack = P.ac_pack = P.comp_pack = 0;
    for (int i = 0; i < j; ++i) {
        P.dc_pack |= (uint32_t)P.blk_dc[i] << (4 * i);
        P.ac_pack |= (uint32_t)P.blk_ac[i] << (4 * i);
        P.comp_pack |= (uint32_t)P.blk_comp[i] << (4 * i);
    }
    P.nblocks = P.nmcu * P.nb;
    P.restart = f.restart;
    P.nseg = f.nseg;
    P.nchunks = f.nchunks;
    P.chunk_bytes = f.chunk_bytes;
    if (f.restart < 0 || f.nseg < 1 || f.nchunks < f.nseg || f.chunk_bytes < 16) return false;
    if (f.restart == 0 ? f.nseg != 1 : f.nseg != (P.nmcu + f.restart - 1) / f.restart) return false;
    return true;
}

// the colour stage alone, on three component planes that calls of their own decoded (multi-scan sequential files): the fields
// color_thread / chroma_at read.  false = a sampling the decoder does not take
inline bool planes_params(Params& P, const uint8_t* y, int64_t y_stride, const uint8_t* cb, const uint8_t* cr, int64_t c_stride, int width,
                          int height, int h0, int v0, uint8_t* out, int64_t out_stride) {
    if (width <= 0 || height <= 0 || width > 65535 || height > 65535) return false;
    if (!((h0 == 1 && v0 == 1) || (h0 == 2 && v0 == 1) || (h0 == 2 && v0 == 2) || (h0 == 1 && v0 == 2) || (h0 == 4 && v0 == 1))) return false;
    if (y_stride < width || c_stride < (width + h0 - 1) / h0 || y_stride > 0x7fffffff || c_stride > 0x7fffffff) return false;
    P = Params{};
    P.ncomp = 3;
    P.width = width;
    P.height = height;
    P.hmax = h0;
    P.vmax = v0;
    P.comp_h[0] = h0; P.comp_v[0] = v0;
    P.comp_h[1] = P.comp_h[2] = P.comp_v[1] = P.comp_v[2] = 1;
    P.plane[0] = const_cast<uint8_t*>(y);
    P.plane[1] = const_cast<uint8_t*>(cb);
    P.plane[2] = const_cast<uint8_t*>(cr);
    P.plane_w[0] = (int32_t)y_stride;
    P.plane_w[1] = P.plane_w[2] = (int32_t)c_stride;
    for (int c = 0; c < 3; ++c) {
        P.real_w[c] = (width * P.comp_h[c] + h0 - 1) / h0;
        P.real_h[c] = (height * P.comp_v[c] + v0 - 1) / v0;
        P.plane_h[c] = P.real_h[c];
    }
    P.out_channels = 3;
    P.out = out;
    P.out_stride = out_stride;
    return true;
}

inline Layout layout_of(const Params& P, int64_t scan_bytes, int out_channels) {
    Layout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) / 256 * 256; return at; };
    L.clean = take((size_t)scan_bytes + 64);                   // compacted scan + words the bit reader may touch past its end
    L.exit_state = take((size_t)P.nchunks * 8);
    L.last_entry = take((size_t)P.nchunks * 8);
    L.nblk = take((size_t)P.nchunks * 4);
    L.blk0 = take((size_t)P.nchunks * 4);
    L.work = take(64 * 4 + 4 * 4);                              // work[64] + status scratch
    L.coef = take((size_t)P.nblocks * 128);
    const size_t ng = (size_t)(P.nmcu + jd::DC_GROUP - 1) / jd::DC_GROUP;
    L.dc_part = take(ng * 16);
    L.dc_base = take(ng * 16);
    for (int c = 0; c < 3; ++c)
        L.plane[c] = (out_channels == 3 && c < P.ncomp) ? take((size_t)P.plane_w[c] * P.plane_h[c]) : 0;
    L.total = o;
    return L;
}


inline void bind(Params& P, const Layout& L, char* ws, const uint8_t* scan, int64_t scan_bytes, const uint32_t* huff_tab,
                 const uint16_t* qt, const uint32_t* block_base, const uint32_t* seg_beg, const uint32_t* seg_end, const int32_t* seg_chunk0, const int32_t* chunk_seg, uint8_t* out,
                 int64_t out_stride, int out_channels, int32_t* status) {
    P.scan = scan;
    P.scan_bytes = (int32_t)scan_bytes;
    P.block_base = block_base;
    P.clean = reinterpret_cast<uint8_t*>(ws + L.clean);
    P.tab = huff_tab;
    P.qt = qt;
    P.seg_beg = seg_beg;
    P.seg_end = seg_end;
    P.seg_chunk0 = seg_chunk0;
    P.chunk_seg = chunk_seg;
    P.out_channels = out_channels;
    P.exit_state = reinterpret_cast<uint64_t*>(ws + L.exit_state);
    P.last_entry = reinterpret_cast<uint64_t*>(ws + L.last_entry);
    P.nblk = reinterpret_cast<int32_t*>(ws + L.nblk);
    P.blk0 = reinterpret_cast<int32_t*>(ws + L.blk0);
    P.work = reinterpret_cast<int32_t*>(ws + L.work);
    P.status = status;
    P.coef = reinterpret_cast<int16_t*>(ws + L.coef);
    P.dc_part = reinterpret_cast<int32_t*>(ws + L.dc_part);
    P.dc_base = reinterpret_cast<int32_t*>(ws + L.dc_base);
    for (int c = 0; c < 3; ++c) P.plane[c] = L.plane[c] ? reinterpret_cast<uint8_t*>(ws + L.plane[c]) : nullptr;
    P.out = out;
    P.out_stride = out_stride;
}

}  // namespace jd
