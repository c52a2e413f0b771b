// Thread-level pieces of the device JPEG decoder (csrc/jpeg_decode.hip) -- gfx950 (MI355X).
//
// Every function here is what ONE thread of a kernel does, written against plain pointers, so the same text also compiles
// with g++ as the CPU lane model of the decoder (tests/jpeg_emul.cpp: the kernels' grids become loops; tests/test_jpeg_cpu.py
// holds it to the oracle and to libjpeg-turbo before anything runs on a GPU).  See jpeg_decode.hip for the algorithm.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define JD_FN __device__ __forceinline__
#define JD_HD __host__ __device__ __forceinline__
#define JD_UNROLL _Pragma("unroll")
#else
#define JD_UNROLL
#define JD_FN static inline
#define JD_HD static inline
#endif

namespace jd {

constexpr int MAX_BLK = 6;                 // blocks per MCU (4:2:0: Y Y Y Y Cb Cr; 4:1:1 the same count, 4:4:0 four)
constexpr int SCAN_T = 1024;               // threads of the single-workgroup scans
constexpr int DC_GROUP = 4;                // MCUs per thread of the DC prediction passes
constexpr uint64_t NO_STATE = ~0ull;

struct Params {
    // ---- the scan and its tables (device) ------------------------------------------------------------------------
    const uint8_t* scan;                   // entropy-coded bytes as in the file: stuffed zeros and RSTn markers still in place
    int32_t scan_bytes;
    const uint32_t* block_base;            // [ceil(scan_bytes / UNSTUFF_BLOCK)] entropy bytes in front of each block of `scan`
    uint8_t* clean;                        // workspace: the scan without stuffed zeros, markers and fill bytes (+ 16 bytes of pad)
    const uint32_t* tab;                   // [4][TAB_SLOT_WORDS] decoding tables of the (up to) four Huffman codes, see below
    const uint16_t* qt;                    // [3][64] quantisation steps of each component, natural (row-major) order
    const uint32_t* seg_beg;               // [nseg] byte range of each restart segment in `clean`
    const uint32_t* seg_end;
    const int32_t* seg_chunk0;             // [nseg] first chunk of the segment
    const int32_t* chunk_seg;              // [nchunks]
    int32_t nseg, nchunks, chunk_bytes;
    // ---- frame geometry ------------------------------------------------------------------------------------------
    int32_t nb;                            // blocks per MCU
    uint32_t dc_pack, ac_pack, comp_pack;  // 4 bits per block-in-MCU: LUT slot of its DC / AC table, its component
    int32_t blk_comp[MAX_BLK], blk_bx[MAX_BLK], blk_by[MAX_BLK], blk_dc[MAX_BLK], blk_ac[MAX_BLK];
    int32_t ncomp, comp_h[3], comp_v[3], comp_j0[3];
    int32_t hmax, vmax;
    int32_t mcux, nmcu, restart;           // MCUs per row, in the frame, per restart interval (0 = none)
    int32_t nblocks;
    int32_t width, height;
    int32_t plane_w[3], plane_h[3];        // padded component planes (samples)
    int32_t real_w[3], real_h[3];          // ceil(width h / hmax), ceil(height v / vmax): the samples that exist
    int32_t out_channels;                  // 1: luma plane, 3: RGB
    // ---- workspace (device) --------------------------------------------------------------------------------------
    uint64_t* exit_state;                  // [nchunks] decoder state at the first symbol that starts at / after the chunk's end
    uint64_t* last_entry;                  // [nchunks] entry state the stored exit was computed from
    int32_t* nblk;                         // [nchunks] blocks completed by symbols that start inside the chunk
    int32_t* blk0;                         // [nchunks] exclusive prefix of nblk
    int32_t* work;                         // [64] chunks decoded by sweep i (0 = fixed point reached)
    int32_t* status;                       // [4] work of the last sweep, invalid codes, segments with a wrong block count, sweeps used
    int16_t* coef;                         // [nblocks][64] in zig-zag (scan) order; DC differences until the prediction pass
    int32_t* dc_part;                      // [ngroups][4] (sum Y, Cb, Cr since the last reset in the group; has reset)
    int32_t* dc_base;                      // [ngroups][4]
    uint8_t* plane[3];                     // colour output only
    uint8_t* out;
    int64_t out_stride;
};

JD_FN uint64_t pack_state(uint64_t pos, uint32_t b, uint32_t z) { return (pos << 16) | ((uint64_t)b << 8) | z; }

// ---- pass 0: the scan without its byte stuffing -----------------------------------------------------------------------
// In the file every data byte 0xFF is followed by a stuffed 0x00, restart intervals end in FF D0..D7 (possibly after FF fill
// bytes).  A decoder that has to step over those while it tracks bit positions spends more instructions on that than on
// Huffman codes, so the scan is compacted once: a byte stays unless it is a stuffed zero (00 after FF), the FF of a marker /
// fill byte (FF not followed by 00) or a marker's code (D0..D7 after FF).  A workgroup takes UNSTUFF_BLOCK bytes, 16 per
// thread; the number of kept bytes in front of each block comes from the host, which has seen every FF anyway (jpeg.plan).
constexpr int UNSTUFF_BLOCK = 4096;
JD_FN uint32_t unstuff_mask(const Params& P, int blk, int t, uint64_t& w0, uint64_t& w1) {   // bit j set: byte 16 t + j of the block stays
    const int64_t r0 = (int64_t)blk * UNSTUFF_BLOCK + 16 * t;
    w0 = w1 = 0;
    if (r0 >= P.scan_bytes) return 0;
    const uint64_t* src = reinterpret_cast<const uint64_t*>(P.scan + r0);     // `scan` is 16-byte aligned and readable to the next multiple of 16
    w0 = src[0];
    w1 = src[1];
    uint32_t keep = 0;
    uint32_t prev = r0 > 0 ? P.scan[r0 - 1] : 0;
    uint32_t cur = (uint32_t)w0 & 0xFF;
    for (int j = 0; j < 16; ++j) {
        const int64_t i = r0 + j;
        if (i >= P.scan_bytes) break;
        uint32_t next;
        if (j < 15) next = (uint32_t)((j + 1 < 8 ? w0 >> (8 * (j + 1)) : w1 >> (8 * (j - 7))) & 0xFF);
        else next = i + 1 < P.scan_bytes ? P.scan[i + 1] : 0xD9;
        if (i + 1 >= P.scan_bytes) next = 0xD9;               // what follows the scan is a marker
        const bool drop = (prev == 0xFF && (cur == 0 || (cur >= 0xD0 && cur <= 0xD7))) || (cur == 0xFF && next != 0);
        if (!drop) keep |= 1u << j;
        prev = cur;
        cur = next;
    }
    return keep;
}
JD_FN void unstuff_store(const Params& P, int blk, uint32_t keep, uint64_t w0, uint64_t w1, uint32_t before) {
    uint8_t* o = P.clean + P.block_base[blk] + before;      // `before`: kept bytes of the block in front of this thread
    for (int j = 0; j < 16; ++j)
        if (keep >> j & 1) *o++ = (uint8_t)((j < 8 ? w0 >> (8 * j) : w1 >> (8 * (j - 8))) & 0xFF);
}
// exclusive prefix of the 256 per-thread counts of a block, the same three phases as the chunk scan below
constexpr int UNSTUFF_T = UNSTUFF_BLOCK / 16, UNSTUFF_G = 16;
JD_FN void unstuff_scan_b1(uint32_t* cnt, uint32_t* grp, int g) {
    uint32_t run = 0;
    for (int t = g * (UNSTUFF_T / UNSTUFF_G); t < (g + 1) * (UNSTUFF_T / UNSTUFF_G); ++t) {
        const uint32_t c = cnt[t];
        cnt[t] = run;
        run += c;
    }
    grp[g] = run;
}
JD_FN void unstuff_scan_b2(uint32_t* grp) {
    uint32_t run = 0;
    for (int g = 0; g < UNSTUFF_G; ++g) {
        const uint32_t c = grp[g];
        grp[g] = run;
        run += c;
    }
}

// ---- bit reader over the compacted scan --------------------------------------------------------------------------------
// 64-bit window, left aligned, refilled 32 bits at a time from aligned big-endian words: at least 32 valid bits at the start
// of every symbol cover the longest code (16) plus the longest value (15).  Positions are bits of `clean`.  Past a segment's
// end the reader sees the next segment (or the pad): every loop is bounded by positions, not by what the bits say.
struct Reader {
    const uint32_t* d;
    uint32_t i;                            // next word to load (the word before it waits in `ahead`)
    uint64_t acc;
    int32_t nbits;
    uint32_t ahead;                        // the next word, loaded one refill early: its latency hides behind a whole refill period
};                                         //   (a lane refills every ~6 symbols, but SOME lane of the wave does in almost every
                                           //   iteration, and a load the wave has to wait for on the spot costs every lane ~250 cycles)
JD_FN uint32_t be32(uint32_t w) { return (w >> 24) | ((w >> 8) & 0xFF00u) | ((w << 8) & 0xFF0000u) | (w << 24); }
JD_FN void reader_init(Reader& r, const uint8_t* clean, uint64_t pos) {
    r.d = reinterpret_cast<const uint32_t*>(clean);
    r.i = (uint32_t)(pos >> 5);
    const uint32_t sh = (uint32_t)(pos & 31);
    r.acc = (uint64_t)be32(r.d[r.i]) << (32 + sh);
    r.nbits = 32 - (int32_t)sh;
    r.ahead = r.d[r.i + 1];                                 // as loaded: the byte swap waits until the word is used
    r.i += 2;
}
JD_FN void reader_fill(Reader& r) {                      // afterwards nbits >= 32
    if (r.nbits < 32) {
        r.acc |= (uint64_t)be32(r.ahead) << (32 - r.nbits);
        r.nbits += 32;
        r.ahead = r.d[r.i];
        r.i += 1;
    }
}
JD_FN uint32_t reader_peek32(const Reader& r) { return (uint32_t)(r.acc >> 32); }
JD_FN void reader_skip(Reader& r, uint32_t n) {          // n <= 32 bits that were valid
    r.acc <<= n;
    r.nbits -= (int32_t)n;
}
JD_FN uint64_t reader_pos(const Reader& r) { return (uint64_t)(r.i - 1) * 32 - (uint64_t)r.nbits; }

JD_FN int extend(uint32_t v, uint32_t s) { return s && v < (1u << (s - 1)) ? (int)v - (int)(1u << s) + 1 : (int)v; }

#if defined(__HIPCC__)
#define JD_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define JD_LOAD64(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define JD_STORE64(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define JD_ATOMIC_ADD(p, v) (*(p) += (v))
#define JD_LOAD64(p) (*(p))
#define JD_STORE64(p, v) (*(p) = (v))
#endif

// ---- chunk geometry ---------------------------------------------------------------------------------------------
struct Chunk {
    int32_t seg;
    uint32_t beg, end, lim;                // bytes [beg, end) of `clean`; lim = end of the segment
    bool first, last;                      // of its segment
};
JD_FN Chunk chunk_of(const Params& P, int c) {
    Chunk k;
    k.seg = P.chunk_seg[c];
    const int i = c - P.seg_chunk0[k.seg];
    k.lim = P.seg_end[k.seg];
    k.beg = P.seg_beg[k.seg] + (uint32_t)i * (uint32_t)P.chunk_bytes;
    const uint32_t e = k.beg + (uint32_t)P.chunk_bytes;
    k.end = e < k.lim ? e : k.lim;
    k.first = i == 0;
    k.last = k.end == k.lim;
    return k;
}
// the state a decoder is ASSUMED to be in at the first byte of a chunk before anything is known: on a byte boundary, at
// the DC coefficient of the MCU's first block
JD_FN uint64_t default_entry(const Chunk& k) { return pack_state((uint64_t)k.beg * 8, 0, 0); }

// ---- Huffman lookup: everything in LDS ---------------------------------------------------------------------------------
// Per code (slot) the host builds (jpeg.py huffman_table), and every workgroup copies into LDS:
//   l1[1024] u16    (length << 8) | symbol for every 10-bit prefix whose code is at most 10 bits long, else 0
//   long[6][3] u32  for lengths 11..16: limit (first left-aligned 16-bit value past the codes of that length), base (the
//                   first code of that length, left aligned), valoff (index of its symbol in vals) -- the canonical code of
//                   ITU T.81 Annex C: codes of one length are consecutive, longer codes follow numerically
//   vals[256] u8    HUFFVAL
// Almost every symbol of a photograph is answered by l1; the few long codes take a six-step search in LDS.  No global memory
// in the symbol loop: a vector-memory load there would make the loop wait for the scan prefetch as well (one counter).
constexpr int L1_BITS = 10;
constexpr int TAB_L1_BYTES = 2 << L1_BITS, TAB_LONG_BYTES = 6 * 3 * 4, TAB_VALS_BYTES = 256;
constexpr int TAB_SLOT_BYTES = (TAB_L1_BYTES + TAB_LONG_BYTES + TAB_VALS_BYTES + 15) / 16 * 16;      // 2384
constexpr int TAB_SLOT_WORDS = TAB_SLOT_BYTES / 4, TAB_WORDS = 4 * TAB_SLOT_WORDS;
JD_FN uint32_t huff_lookup(const uint32_t* tab, uint32_t slot, uint32_t pk32) {        // (length << 8) | symbol, 0 = no such code
    const uint8_t* t = reinterpret_cast<const uint8_t*>(tab) + slot * TAB_SLOT_BYTES;
    uint32_t e = reinterpret_cast<const uint16_t*>(t)[pk32 >> (32 - L1_BITS)];
    if (e == 0) {
        const uint32_t code = pk32 >> 16;
        const uint32_t* lg = reinterpret_cast<const uint32_t*>(t + TAB_L1_BYTES);
        const uint8_t* vals = t + TAB_L1_BYTES + TAB_LONG_BYTES;
        for (uint32_t l = 11; l <= 16; ++l) {
            const uint32_t* q = lg + 3 * (l - 11);
            if (code < q[0]) {
                e = (l << 8) | vals[q[2] + ((code - q[1]) >> (16 - l))];
                break;
            }
        }
    }
    return e;
}

// ---- the Huffman decoder of one chunk ----------------------------------------------------------------------------
// Decodes the symbols that START inside [entry position, chunk end): DC size + difference when z == 0, AC run / size (EOB,
// ZRL) otherwise (ITU T.81 F.2.2; libjpeg-turbo jdhuff.c decode_mcu).  WRITE: coefficients go to coef[block][natural index]
// (DC as its difference), blocks from blk on, at most up to blk_limit.  An invalid prefix costs one bit and is counted:
// on a mis-synchronised path that is routine, on the final path it means a corrupt file.
struct ChunkResult {
    uint64_t exit;
    int32_t nblk, nbad;
};
template <bool WRITE>
JD_FN ChunkResult decode_chunk(const Params& P, const Chunk& k, uint64_t entry, int32_t blk, int32_t blk_limit, const uint32_t* tab) {
    Reader r;
    reader_init(r, P.clean, entry >> 16);
    uint32_t b = (uint32_t)(entry >> 8) & 0xFF, z = (uint32_t)entry & 0xFF;
    const uint64_t end_pos = (uint64_t)k.end * 8;
    ChunkResult res;
    res.nblk = 0;
    res.nbad = 0;
    while (reader_pos(r) < end_pos && (!WRITE || blk < blk_limit)) {
        reader_fill(r);
        const uint32_t pk = reader_peek32(r);
        const bool is_dc = z == 0;
        const uint32_t slot = ((is_dc ? P.dc_pack : P.ac_pack) >> (4 * b)) & 15;
        const uint32_t e = huff_lookup(tab, slot, pk);
        const uint32_t len = e >> 8;
        if (len == 0) {                                        // no code has this prefix
            reader_skip(r, 1);
            res.nbad += 1;
            continue;
        }
        const uint32_t sym = e & 0xFF;
        if (is_dc && sym > 15) res.nbad += 1;
        const uint32_t s = sym & 15, run = is_dc ? 0 : sym >> 4;
        const uint32_t v = (uint32_t)(((uint64_t)(pk << len)) >> (32 - s));     // the s bits after the code (0 for s = 0)
        reader_skip(r, len + s);                               // <= 16 + 15 of the >= 32 valid bits
        if (!is_dc && s == 0) {
            z = run == 15 ? z + 16 : 64;                       // ZRL / EOB
        } else {
            z += run;
            if (WRITE && z < 64) P.coef[(int64_t)blk * 64 + z] = (int16_t)extend(v, s);      // zig-zag order; the IDCT undoes it
            z += 1;
        }
        if (z >= 64) {
            z = 0;
            b = b + 1 == (uint32_t)P.nb ? 0 : b + 1;
            res.nblk += 1;
            blk += 1;
        }
    }
    res.exit = pack_state(reader_pos(r), b, z);
    return res;
}

// ---- kernels, one thread each --------------------------------------------------------------------------------------
JD_FN void init_thread(const Params& P, int c) {
    uint64_t e = NO_STATE;
    if (c + 1 < P.nchunks && P.chunk_seg[c + 1] == P.chunk_seg[c]) e = default_entry(chunk_of(P, c + 1));
    P.exit_state[c] = e;
    P.last_entry[c] = NO_STATE;
    P.nblk[c] = 0;
}

// One relaxation ROUND: chunk c decodes from the exit state its predecessor currently reports (in the same launch that may
// be the old or the new one: both are proposals) unless that is the entry its stored result was computed from.  Chunk 0 of a
// segment starts from the truth, so after round i the first i + 1 chunks of every segment are final; because Huffman
// streams self-synchronise, a wrong entry usually leads to the right exit within the chunk and the fixed point arrives
// after a handful of rounds.  A round that decodes nothing IS the fixed point (every stored result matches its entry).
// r06: a sweep LAUNCH runs rounds inside each workgroup (SWEEP_WG consecutive chunks) until none of its chunks changes (at most
// SWEEP_ROUNDS): where the stream does NOT self-synchronise -- a flat area is one short code repeated, a decoder that enters it
// out of phase stays out of phase -- the truth still advances one chunk per round, and r05 paid one 54-us launch per round for
// that (a 12-megapixel frame with a blown-out sky: > 400 launches, then CorruptJpeg).  Now a launch settles every workgroup whose
// first entry is true, so the number of launches is bounded by the workgroups a segment spans, + 1 (flat_launch_bound).
constexpr int SWEEP_WG = 256;              // chunks per workgroup of the sweep kernel
constexpr int SWEEP_ROUNDS = 256;          // rounds per launch: enough to carry a true entry through the whole workgroup
JD_FN bool sweep_needs(const Params& P, int c, uint64_t& entry) {
    const Chunk k = chunk_of(P, c);
    entry = k.first ? pack_state((uint64_t)k.beg * 8, 0, 0) : JD_LOAD64(&P.exit_state[c - 1]);
    return entry != P.last_entry[c];
}
JD_FN void sweep_thread(const Params& P, int c, int sweep, uint64_t entry, const uint32_t* tab) {
    const Chunk k = chunk_of(P, c);
    const ChunkResult res = decode_chunk<false>(P, k, entry, 0, 0, tab);
    JD_STORE64(&P.exit_state[c], res.exit);
    P.last_entry[c] = entry;
    P.nblk[c] = res.nblk;
    JD_ATOMIC_ADD(&P.work[sweep], 1);
}

// Exclusive prefix of nblk over all chunks, one workgroup: spans, scan of the span sums, spans again.
JD_FN int32_t scan_span(const Params& P) { return (P.nchunks + SCAN_T - 1) / SCAN_T; }
JD_FN void scan_phase_a(const Params& P, int t, int32_t* part) {
    const int span = scan_span(P);
    int32_t s = 0;
    for (int c = t * span; c < (t + 1) * span && c < P.nchunks; ++c) s += P.nblk[c];
    part[t] = s;
}
// the SCAN_T span sums -> their exclusive prefix, in three short phases instead of one thread walking 1024 LDS words:
// b1 thread g < SCAN_G scans its SCAN_T / SCAN_G entries in place and leaves the group total in grp[g]; b2 thread 0 scans the
// SCAN_G totals; b3 every thread adds its group's offset.
constexpr int SCAN_G = 32;
JD_FN void scan_phase_b1(int32_t* part, int32_t* grp, int g) {
    int32_t run = 0;
    for (int t = g * (SCAN_T / SCAN_G); t < (g + 1) * (SCAN_T / SCAN_G); ++t) {
        const int32_t s = part[t];
        part[t] = run;
        run += s;
    }
    grp[g] = run;
}
JD_FN void scan_phase_b2(int32_t* grp) {                    // thread 0
    int32_t run = 0;
    for (int g = 0; g < SCAN_G; ++g) {
        const int32_t s = grp[g];
        grp[g] = run;
        run += s;
    }
}
JD_FN void scan_phase_b3(int32_t* part, const int32_t* grp, int t) { part[t] += grp[t / (SCAN_T / SCAN_G)]; }
JD_FN void scan_phase_c(const Params& P, int t, const int32_t* part) {
    const int span = scan_span(P);
    int32_t run = part[t];
    for (int c = t * span; c < (t + 1) * span && c < P.nchunks; ++c) {
        P.blk0[c] = run;
        run += P.nblk[c];
    }
}

JD_FN void write_thread(const Params& P, int c, const uint32_t* tab) {
    const Chunk k = chunk_of(P, c);
    const uint64_t entry = k.first ? pack_state((uint64_t)k.beg * 8, 0, 0) : P.exit_state[c - 1];
    const int32_t per_seg = P.restart ? P.restart * P.nb : P.nblocks;
    const int32_t seg_blk0 = k.seg * per_seg;
    int32_t limit = seg_blk0 + per_seg;
    if (limit > P.nblocks) limit = P.nblocks;
    const int32_t blk = seg_blk0 + P.blk0[c] - P.blk0[P.seg_chunk0[k.seg]];
    if (blk >= limit && !k.last) return;
    const ChunkResult res = decode_chunk<true>(P, k, entry, blk, limit, tab);
    if (res.nbad) JD_ATOMIC_ADD(&P.status[1], res.nbad);
    if (k.last && blk + res.nblk != limit) JD_ATOMIC_ADD(&P.status[2], 1);
}

// DC prediction (T.81 F.2.1.3.1: DC = previous DC of the same component + difference, reset to 0 at every restart):
// pass 1 sums the differences of DC_GROUP MCUs per component, a single-workgroup scan turns the sums into the prediction
// each group starts from, pass 3 rewrites the differences as values.
JD_HD int32_t dc_ngroups(const Params& P) { return (P.nmcu + DC_GROUP - 1) / DC_GROUP; }
JD_FN bool mcu_resets(const Params& P, int m) { return m == 0 || (P.restart && m % P.restart == 0); }
JD_FN void dc_sum_thread(const Params& P, int g) {
    int32_t sum[3] = {0, 0, 0};
    int32_t reset = 0;
    for (int m = g * DC_GROUP; m < (g + 1) * DC_GROUP && m < P.nmcu; ++m) {
        if (mcu_resets(P, m)) {
            sum[0] = sum[1] = sum[2] = 0;
            reset = 1;
        }
        for (int j = 0; j < P.nb; ++j) sum[(P.comp_pack >> (4 * j)) & 15] += P.coef[((int64_t)m * P.nb + j) * 64];
    }
    int32_t* o = P.dc_part + 4 * g;
    o[0] = sum[0];
    o[1] = sum[1];
    o[2] = sum[2];
    o[3] = reset;
}
JD_FN int32_t dc_span(const Params& P) { return (dc_ngroups(P) + SCAN_T - 1) / SCAN_T; }
// phase a: thread t folds its span of groups into one (sum since the last reset, has reset) tuple in part[4 t ..]
JD_FN void dc_scan_phase_a(const Params& P, int t, int32_t* part) {
    const int span = dc_span(P), ng = dc_ngroups(P);
    int32_t sum[3] = {0, 0, 0}, reset = 0;
    for (int g = t * span; g < (t + 1) * span && g < ng; ++g) {
        const int32_t* p = P.dc_part + 4 * g;
        if (p[3]) {
            sum[0] = p[0]; sum[1] = p[1]; sum[2] = p[2];
            reset = 1;
        } else {
            sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2];
        }
    }
    part[4 * t] = sum[0]; part[4 * t + 1] = sum[1]; part[4 * t + 2] = sum[2]; part[4 * t + 3] = reset;
}
// (sum, reset) tuples compose: folding a tuple into a running prediction gives  reset ? sum : prediction + sum.  Same three
// phases as above: b1 replaces every tuple of a group by the fold of the tuples BEFORE it in the group and leaves the group's
// own fold in grp; b2 turns the group folds into the prediction each group starts from; b3 applies it.
JD_FN void dc_fold(int32_t* acc, const int32_t* t) {        // acc := acc followed by t
    if (t[3]) { acc[0] = t[0]; acc[1] = t[1]; acc[2] = t[2]; acc[3] = 1; }
    else { acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2]; }
}
JD_FN void dc_scan_phase_b1(int32_t* part, int32_t* grp, int g) {
    int32_t run[4] = {0, 0, 0, 0};
    for (int t = g * (SCAN_T / SCAN_G); t < (g + 1) * (SCAN_T / SCAN_G); ++t) {
        int32_t* p = part + 4 * t;
        const int32_t cur[4] = {p[0], p[1], p[2], p[3]};
        p[0] = run[0]; p[1] = run[1]; p[2] = run[2]; p[3] = run[3];
        dc_fold(run, cur);
    }
    grp[4 * g] = run[0]; grp[4 * g + 1] = run[1]; grp[4 * g + 2] = run[2]; grp[4 * g + 3] = run[3];
}
JD_FN void dc_scan_phase_b2(int32_t* grp) {                 // thread 0: grp[g] := prediction at the start of group g
    int32_t run[4] = {0, 0, 0, 0};
    for (int g = 0; g < SCAN_G; ++g) {
        int32_t* p = grp + 4 * g;
        const int32_t cur[4] = {p[0], p[1], p[2], p[3]};
        p[0] = run[0]; p[1] = run[1]; p[2] = run[2];
        dc_fold(run, cur);
    }
}
JD_FN void dc_scan_phase_b3(int32_t* part, const int32_t* grp, int t) {   // part[t] := prediction at the start of span t
    int32_t* p = part + 4 * t;
    if (!p[3]) {
        const int32_t* c = grp + 4 * (t / (SCAN_T / SCAN_G));
        p[0] += c[0]; p[1] += c[1]; p[2] += c[2];
    }
}
JD_FN void dc_scan_phase_c(const Params& P, int t, const int32_t* part) {
    const int span = dc_span(P), ng = dc_ngroups(P);
    int32_t run[3] = {part[4 * t], part[4 * t + 1], part[4 * t + 2]};
    for (int g = t * span; g < (t + 1) * span && g < ng; ++g) {
        const int32_t* p = P.dc_part + 4 * g;
        int32_t* o = P.dc_base + 4 * g;
        o[0] = run[0]; o[1] = run[1]; o[2] = run[2];
        if (p[3]) { run[0] = p[0]; run[1] = p[1]; run[2] = p[2]; }
        else { run[0] += p[0]; run[1] += p[1]; run[2] += p[2]; }
    }
}
JD_FN void dc_apply_thread(const Params& P, int g) {
    int32_t pred[3] = {P.dc_base[4 * g], P.dc_base[4 * g + 1], P.dc_base[4 * g + 2]};
    for (int m = g * DC_GROUP; m < (g + 1) * DC_GROUP && m < P.nmcu; ++m) {
        if (mcu_resets(P, m)) pred[0] = pred[1] = pred[2] = 0;
        for (int j = 0; j < P.nb; ++j) {
            int16_t* d = P.coef + ((int64_t)m * P.nb + j) * 64;
            const int comp = (P.comp_pack >> (4 * j)) & 15;
            pred[comp] += *d;
            *d = (int16_t)pred[comp];
        }
    }
}

// ---- jidctint.c: jpeg_idct_islow -----------------------------------------------------------------------------------
// Loeffler-Ligtenberg-Moschytz with 13-bit constants: a column pass that keeps 2 extra bits, a row pass, the 1024-entry
// range-limit table indexed modulo (jdmaster.c prepare_range_limit_table) written out as arithmetic.  The library's
// zero-AC shortcuts give the same numbers as the full butterfly, so there are none here.
JD_FN void idct_1d(const int32_t in0, const int32_t in1, const int32_t in2, const int32_t in3, const int32_t in4,
                   const int32_t in5, const int32_t in6, const int32_t in7, int32_t* o) {
    int32_t z1 = (in2 + in6) * 4433;                        // FIX(0.541196100)
    const int32_t t2 = z1 + in6 * (-15137);                 // FIX(1.847759065)
    const int32_t t3 = z1 + in2 * 6270;                     // FIX(0.765366865)
    const int32_t t0 = (in0 + in4) * 8192, t1 = (in0 - in4) * 8192;
    const int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    int32_t a0 = in7, a1 = in5, a2 = in3, a3 = in1;
    z1 = a0 + a3;
    int32_t z2 = a1 + a2, z3 = a0 + a2, z4 = a1 + a3;
    const int32_t z5 = (z3 + z4) * 9633;                    // FIX(1.175875602)
    a0 *= 2446;                                             // FIX(0.298631336)
    a1 *= 16819;                                            // FIX(2.053119869)
    a2 *= 25172;                                            // FIX(3.072711026)
    a3 *= 12299;                                            // FIX(1.501321110)
    z1 *= -7373;                                            // FIX(0.899976223)
    z2 *= -20995;                                           // FIX(2.562915447)
    z3 = z3 * -16069 + z5;                                  // FIX(1.961570560)
    z4 = z4 * -3196 + z5;                                   // FIX(0.390180644)
    a0 += z1 + z3;
    a1 += z2 + z4;
    a2 += z2 + z3;
    a3 += z1 + z4;
    o[0] = t10 + a3; o[7] = t10 - a3;
    o[1] = t11 + a2; o[6] = t11 - a2;
    o[2] = t12 + a1; o[5] = t12 - a1;
    o[3] = t13 + a0; o[4] = t13 - a0;
}
JD_FN uint8_t range_limit(int32_t x) {
    const int32_t i = x & 1023;
    if (i < 512) return (uint8_t)(i + 128 > 255 ? 255 : i + 128);
    const int32_t v = i - 1024 + 128;
    return (uint8_t)(v < 0 ? 0 : v);
}
JD_FN void idct_thread(const Params& P, int blk) {
    const int m = blk / P.nb, j = blk - m * P.nb;
    const int comp = P.blk_comp[j];
    if (P.out_channels == 1 && comp != 0) return;
    const int row0 = ((m / P.mcux) * P.comp_v[comp] + P.blk_by[j]) * 8, col0 = ((m % P.mcux) * P.comp_h[comp] + P.blk_bx[j]) * 8;
    if (P.out_channels == 1 && (row0 >= P.height || col0 >= P.width)) return;
    const int16_t* cf = P.coef + (int64_t)blk * 64;
    const uint16_t* q = P.qt + comp * 64;
    int32_t ws[64], o[8];
    // coefficient (row r, column c) sits at zig-zag position ZZI[8 r + c] of the block: compile-time indices in the unrolled loop
    constexpr int ZZI[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
                             10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
    JD_UNROLL
    for (int c = 0; c < 8; ++c) {
        idct_1d((int32_t)cf[ZZI[c]] * q[c], (int32_t)cf[ZZI[8 + c]] * q[8 + c], (int32_t)cf[ZZI[16 + c]] * q[16 + c],
                (int32_t)cf[ZZI[24 + c]] * q[24 + c], (int32_t)cf[ZZI[32 + c]] * q[32 + c], (int32_t)cf[ZZI[40 + c]] * q[40 + c],
                (int32_t)cf[ZZI[48 + c]] * q[48 + c], (int32_t)cf[ZZI[56 + c]] * q[56 + c], o);
        JD_UNROLL
        for (int r = 0; r < 8; ++r) ws[r * 8 + c] = (o[r] + (1 << 10)) >> 11;          // DESCALE(CONST_BITS - PASS1_BITS)
    }
    uint8_t* dst;
    int64_t stride;
    int rows = 8, cols = 8;
    if (P.out_channels == 1) {
        dst = P.out + (int64_t)row0 * P.out_stride + col0;
        stride = P.out_stride;
        if (P.height - row0 < rows) rows = P.height - row0;
        if (P.width - col0 < cols) cols = P.width - col0;
    } else {
        dst = P.plane[comp] + (int64_t)row0 * P.plane_w[comp] + col0;
        stride = P.plane_w[comp];
    }
    JD_UNROLL
    for (int r = 0; r < 8; ++r) {
        idct_1d(ws[r * 8], ws[r * 8 + 1], ws[r * 8 + 2], ws[r * 8 + 3], ws[r * 8 + 4], ws[r * 8 + 5], ws[r * 8 + 6], ws[r * 8 + 7], o);
        uint8_t px[8];
        JD_UNROLL
        for (int c = 0; c < 8; ++c) px[c] = range_limit((o[c] + (1 << 17)) >> 18);      // CONST_BITS + PASS1_BITS + 3
        if (r < rows) {
            if (cols == 8) {                                    // blocks start on 8-byte columns of rows the caller aligned or not:
                JD_UNROLL                                       // byte stores keep any out_stride legal
                for (int c = 0; c < 8; ++c) dst[r * stride + c] = px[c];
            } else {
                JD_UNROLL
                for (int c = 0; c < 8; ++c)
                    if (c < cols) dst[r * stride + c] = px[c];
            }
        }
    }
}

// ---- jdsample.c fancy upsampling + jdcolor.c ycc_rgb_convert, one output pixel -----------------------------------------
JD_FN int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
JD_FN int chroma_at(const Params& P, int comp, int x, int y) {
    const uint8_t* pl = P.plane[comp];
    const int pw = P.plane_w[comp], rw = P.real_w[comp], rh = P.real_h[comp];
    const int hr = P.hmax / P.comp_h[comp], vr = P.vmax / P.comp_v[comp];
    if (hr == 1 && vr == 1) return pl[(int64_t)y * pw + x];
    // jinit_upsampler: 4:1:1 goes through int_upsample (replication); the 2h fancy filters only for components more than two samples
    // wide, else h2v1_upsample / h2v2_upsample (replication)
    if (hr == 4) return pl[(int64_t)y * pw + (x >> 2)];
    if (hr == 2 && rw <= 2) return pl[(int64_t)(vr == 2 ? y >> 1 : y) * pw + (x >> 1)];
    if (hr == 1) {                                          // h1v2_fancy_upsample (4:4:0): 3/4 nearer row + 1/4 further, bias 1 / 2
        const int cy = y >> 1;
        const int fy = clampi((y & 1) ? cy + 1 : cy - 1, 0, rh - 1);
        return (3 * pl[(int64_t)cy * pw + x] + pl[(int64_t)fy * pw + x] + ((y & 1) ? 2 : 1)) >> 2;
    }
    const int cx = x >> 1;
    if (vr == 1) {                                          // h2v1_fancy_upsample: 3/4 nearer + 1/4 further, edges replicated
        const uint8_t* row = pl + (int64_t)y * pw;
        const int cur = row[cx];
        return (x & 1) ? (3 * cur + row[clampi(cx + 1, 0, rw - 1)] + 2) >> 2 : (3 * cur + row[clampi(cx - 1, 0, rw - 1)] + 1) >> 2;
    }
    // h2v2_fancy_upsample: column sums 3 nearer row + further row (the row above the first / below the last real row is a copy
    // of it, jdmainct.c), then 3/4 : 1/4 across, rounding 8 / 7 alternating
    const int cy = y >> 1;
    const int fy = clampi((y & 1) ? cy + 1 : cy - 1, 0, rh - 1);
    const uint8_t* r0 = pl + (int64_t)cy * pw;
    const uint8_t* r1 = pl + (int64_t)fy * pw;
    const int cur = 3 * r0[cx] + r1[cx];
    const int ox = clampi((x & 1) ? cx + 1 : cx - 1, 0, rw - 1);
    return (3 * cur + 3 * r0[ox] + r1[ox] + ((x & 1) ? 7 : 8)) >> 4;
}
JD_FN void color_thread(const Params& P, int x, int y) {
    const int Y = P.plane[0][(int64_t)y * P.plane_w[0] + x];
    uint8_t* o = P.out + (int64_t)y * P.out_stride + 3 * x;
    if (P.ncomp == 1) {
        o[0] = o[1] = o[2] = (uint8_t)Y;
        return;
    }
    const int cb = chroma_at(P, 1, x, y) - 128, cr = chroma_at(P, 2, x, y) - 128;
    // build_ycc_rgb_table: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554
    o[0] = (uint8_t)clampi(Y + ((91881 * cr + 32768) >> 16), 0, 255);
    o[1] = (uint8_t)clampi(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16), 0, 255);
    o[2] = (uint8_t)clampi(Y + ((116130 * cb + 32768) >> 16), 0, 255);
}

}  // namespace jd
