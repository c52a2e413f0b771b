/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the JPEG decode the reference's image readers start with.
 *
 * The reference decodes every frame with OpenCV:
 *     cv2.imread(path, cv2.IMREAD_GRAYSCALE)      src/dataset/utils.py:127 (read_grayscale), :183 (read_grayscale_megadepth)
 *     cv2.imread(path, cv2.IMREAD_COLOR) + BGR2RGB  src/dataset/utils.py:86-92 (read_rgb)
 * OpenCV is a third-party dependency (requirements.txt: opencv-python, not vendored under /root/reference) whose JPEG
 * reader (modules/imgcodecs/src/grfmt_jpeg.cpp) is a thin wrapper of libjpeg-turbo with the library defaults
 * (dct_method = JDCT_ISLOW, do_fancy_upsampling = TRUE) and out_color_space = JCS_GRAYSCALE for IMREAD_GRAYSCALE -- the
 * luma plane itself, no colour conversion -- or JCS_RGB (BGR byte order) for IMREAD_COLOR.  This file restates the published
 * algorithm of that path for baseline (SOF0 / SOF1, 8-bit, Huffman) files:
 *     jdhuff.c   decode_mcu                (ITU T.81 Annex F.2.2: DC differences, AC run / size pairs, EOB, ZRL, restarts)
 *     jidctint.c jpeg_idct_islow           (Loeffler-Ligtenberg-Moschytz, 13-bit constants, two passes, range limit table)
 *     jdsample.c h2v1_fancy_upsample, h2v2_fancy_upsample (triangle filters, 3:1 and 9:3:3:1, edge samples replicated)
 *     jdcolor.c  ycc_rgb_convert           (16-bit fixed-point tables)
 * written sequentially (one bit reader, MCU after MCU) -- on purpose nothing like the device decoder it checks
 * (detectorfreesfm_amd/csrc/jpeg_decode.hip: chunk-parallel, self-synchronising).
 *
 * PINNED: tests/test_jpeg_cpu.py compares this file byte for byte with libjpeg-turbo itself through the installed Pillow
 * (whose decoder makes the same library calls: draft('L') selects JCS_GRAYSCALE, 'RGB' is the library's own output) on
 * synthetic files of every supported sampling / restart / table combination and, where /root/reference is mounted, on the
 * reference's eight example-scene JPEGs.  cv2 itself is not installed in the build image: the pin is to the library it wraps.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define E_FORMAT (-1)      /* not a JPEG / truncated / corrupt */
#define E_UNSUPPORTED (-2) /* progressive, arithmetic, 12-bit, CMYK, multi-scan, unsupported sampling */

typedef struct {
    int present;
    uint8_t bits[17];
    uint8_t vals[256];
    int mincode[17], maxcode[17], valptr[17];
} Huff;

typedef struct {
    int id, h, v, tq, td, ta;
    int bw, bh;            /* blocks per row / column of the padded plane */
    int pw, ph;            /* padded plane size in samples */
    int rw, rh;            /* real ("downsampled") size in samples: ceil(W h / hmax), ceil(H v / vmax) */
    uint8_t* plane;
    int pred;
} Comp;

typedef struct {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc;
    int n;
    int marker;            /* a marker was reached: feed zeros */
} Bits;

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static void huff_prepare(Huff* h) {
    /* ITU T.81 Annex C (code generation) + F.2.2.3 (decoder tables) */
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        code += h->bits[l];
        k += h->bits[l];
        h->maxcode[l] = h->bits[l] ? code - 1 : -1;
        code <<= 1;
    }
}

static void fill(Bits* b) {
    while (b->n <= 24) {
        int c = 0;
        if (!b->marker && b->p < b->end) {
            c = *b->p;
            if (c == 0xFF) {
                const int c2 = b->p + 1 < b->end ? b->p[1] : 0xD9;
                if (c2 == 0) b->p += 2;
                else { b->marker = 1; c = 0; }
            } else {
                b->p++;
            }
        }
        b->acc |= (uint32_t)c << (24 - b->n);
        b->n += 8;
    }
}
static int getbits(Bits* b, int s) {
    if (s == 0) return 0;
    fill(b);
    const int v = (int)(b->acc >> (32 - s));
    b->acc <<= s;
    b->n -= s;
    return v;
}
static int decode_sym(Bits* b, const Huff* h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | getbits(b, 1);
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) return h->vals[h->valptr[l] + code - h->mincode[l]];
    }
    return -1;
}
static int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

/* jidctint.c */
#define CONST_BITS 13
#define PASS1_BITS 2
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
static uint8_t range_limit(int x) {
    const int i = x & 1023;                   /* the 1024-entry post-IDCT table of jdmaster.c, indexed modulo */
    if (i < 512) return (uint8_t)(i + 128 > 255 ? 255 : i + 128);
    const int v = i - 1024 + 128;
    return (uint8_t)(v < 0 ? 0 : v);
}
static void idct_1d(const int32_t* in, int stride, int32_t* o) {
    /* one LL&M pass over in[0], in[stride], ..: returns the eight UNscaled outputs o[0..7] */
    int32_t z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
    z2 = in[2 * stride];
    z3 = in[6 * stride];
    z1 = (z2 + z3) * 4433;
    tmp2 = z1 + z3 * (-15137);
    tmp3 = z1 + z2 * 6270;
    z2 = in[0];
    z3 = in[4 * stride];
    tmp0 = (z2 + z3) * (1 << CONST_BITS);
    tmp1 = (z2 - z3) * (1 << CONST_BITS);
    tmp10 = tmp0 + tmp3;
    tmp13 = tmp0 - tmp3;
    tmp11 = tmp1 + tmp2;
    tmp12 = tmp1 - tmp2;
    tmp0 = in[7 * stride];
    tmp1 = in[5 * stride];
    tmp2 = in[3 * stride];
    tmp3 = in[1 * stride];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    z4 = tmp1 + tmp3;
    z5 = (z3 + z4) * 9633;
    tmp0 *= 2446;
    tmp1 *= 16819;
    tmp2 *= 25172;
    tmp3 *= 12299;
    z1 *= -7373;
    z2 *= -20995;
    z3 *= -16069;
    z4 *= -3196;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    o[0] = tmp10 + tmp3;
    o[7] = tmp10 - tmp3;
    o[1] = tmp11 + tmp2;
    o[6] = tmp11 - tmp2;
    o[2] = tmp12 + tmp1;
    o[5] = tmp12 - tmp1;
    o[3] = tmp13 + tmp0;
    o[4] = tmp13 - tmp0;
}
static void idct_islow(const int16_t* coef, const uint16_t* q, uint8_t* out, int stride) {
    int32_t deq[64], ws[64], o[8];
    for (int i = 0; i < 64; ++i) deq[i] = (int32_t)coef[i] * (int32_t)q[i];
    for (int c = 0; c < 8; ++c) {
        idct_1d(deq + c, 8, o);
        for (int r = 0; r < 8; ++r) ws[r * 8 + c] = DESCALE(o[r], CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; ++r) {
        idct_1d(ws + r * 8, 1, o);
        for (int c = 0; c < 8; ++c) out[r * stride + c] = range_limit(DESCALE(o[c], CONST_BITS + PASS1_BITS + 3));
    }
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* one chroma sample at full resolution (jdsample.c): fancy h2v1 / h2v2 / h1v2, replication for h4v1, plain copy for h1v1 */
static int upsampled(const Comp* c, int hmax, int vmax, int x, int y) {
    const int hr = hmax / c->h, vr = vmax / c->v;
    if (hr == 1 && vr == 1) return c->plane[y * c->pw + x];
    /* jinit_upsampler: 4:1:1 (and every other integral ratio without a special case) goes through int_upsample = replication;
       the 2h fancy filters are only chosen when the component is more than two samples wide, else h2v1_upsample / h2v2_upsample
       (replication again) */
    if (hr == 4 && vr == 1) return c->plane[y * c->pw + (x >> 2)];
    if (hr == 2 && c->rw <= 2) return c->plane[(vr == 2 ? y >> 1 : y) * c->pw + (x >> 1)];
    if (hr == 1 && vr == 2) {
        /* h1v2_fancy_upsample (4:4:0): 3/4 nearer row + 1/4 further row, bias 1 for the upper output row, 2 for the lower */
        const int cy = y >> 1;
        const int fy = clampi((y & 1) ? cy + 1 : cy - 1, 0, c->rh - 1);
        return (3 * c->plane[cy * c->pw + x] + c->plane[fy * c->pw + x] + ((y & 1) ? 2 : 1)) >> 2;
    }
    if (hr == 2 && vr == 1) {
        const int cx = x >> 1;
        const uint8_t* row = c->plane + y * c->pw;
        const int cur = row[cx];
        if (x & 1) return (3 * cur + row[clampi(cx + 1, 0, c->rw - 1)] + 2) >> 2;
        return (3 * cur + row[clampi(cx - 1, 0, c->rw - 1)] + 1) >> 2;
    }
    /* h2v2: vertical 3:1 first (the nearer row is y >> 1, the further one above for even y, below for odd y; the rows
       above the first / below the last REAL row are copies of it, jdmainct.c), then horizontal 3:1 on the column sums */
    const int cy = y >> 1, cx = x >> 1;
    const int fy = clampi((y & 1) ? cy + 1 : cy - 1, 0, c->rh - 1);
    const uint8_t* r0 = c->plane + cy * c->pw;
    const uint8_t* r1 = c->plane + fy * c->pw;
    const int cur = 3 * r0[cx] + r1[cx];
    if (x & 1) {
        const int nx = clampi(cx + 1, 0, c->rw - 1);
        return (3 * cur + (3 * r0[nx] + r1[nx]) + 7) >> 4;
    }
    const int lx = clampi(cx - 1, 0, c->rw - 1);
    return (3 * cur + (3 * r0[lx] + r1[lx]) + 8) >> 4;
}

/* info[0..]: width, height, ncomp, progressive, restart interval, h0, v0, h1, v1, h2, v2, exif orientation (0 = none) */
static int parse_and_decode(const uint8_t* buf, long n, int want_color, uint8_t* out, int* info) {
    if (n < 4 || buf[0] != 0xFF || buf[1] != 0xD8) return E_FORMAT;
    uint16_t qt[4][64];
    Huff dc[4], ac[4];
    Comp comp[3];
    memset(dc, 0, sizeof dc);
    memset(ac, 0, sizeof ac);
    memset(comp, 0, sizeof comp);
    int W = 0, H = 0, ncomp = 0, progressive = 0, restart = 0, have_sof = 0, orientation = 0, adobe_transform = -1, jfif = 0;
    int scan_comp = -1;            /* -1: the scan interleaves every component; c: a non-interleaved scan of component c */
    int geometry_done = 0, decoded[3] = {0, 0, 0}, hmax = 1, vmax = 1, rc = 0;
    long p = 2;
  next_scan:                       /* r06: a sequential file may carry its components in separate scans (T.81 A.2.2) */
    scan_comp = -1;
    {
    int saw_sos = 0;
    while (p + 4 <= n) {
        if (buf[p] != 0xFF) { rc = E_FORMAT; goto done; }
        while (p < n && buf[p] == 0xFF) ++p;                      /* fill bytes */
        if (p >= n) { rc = E_FORMAT; goto done; }
        const int m = buf[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) { rc = E_FORMAT; goto done; }
        if (p + 2 > n) { rc = E_FORMAT; goto done; }
        const long len = (buf[p] << 8) | buf[p + 1];
        if (len < 2 || p + len > n) { rc = E_FORMAT; goto done; }
        const uint8_t* s = buf + p + 2;
        const long sl = len - 2;
        if (m == 0xDB) {                                          /* DQT */
            long i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                if (tq > 3) { rc = E_FORMAT; goto done; }
                ++i;
                for (int k = 0; k < 64; ++k) {
                    int v;
                    if (pq) { v = (s[i] << 8) | s[i + 1]; i += 2; } else v = s[i++];
                    qt[tq][ZIGZAG[k]] = (uint16_t)v;
                }
            }
        } else if (m == 0xC4) {                                   /* DHT */
            long i = 0;
            while (i < sl) {
                const int tc = s[i] >> 4, th = s[i] & 15;
                if (th > 3 || tc > 1) { rc = E_FORMAT; goto done; }
                Huff* h = tc ? &ac[th] : &dc[th];
                int cnt = 0;
                h->bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { h->bits[l] = s[i + l]; cnt += h->bits[l]; }
                if (cnt > 256) { rc = E_FORMAT; goto done; }
                memcpy(h->vals, s + i + 17, cnt);
                h->present = 1;
                huff_prepare(h);
                i += 17 + cnt;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {         /* SOF0 / SOF1 / SOF2 */
            if (s[0] != 8) { rc = E_UNSUPPORTED; goto done; }
            H = (s[1] << 8) | s[2];
            W = (s[3] << 8) | s[4];
            ncomp = s[5];
            progressive = m == 0xC2;
            if (ncomp != 1 && ncomp != 3) { rc = E_UNSUPPORTED; goto done; }
            for (int c = 0; c < ncomp; ++c) {
                comp[c].id = s[6 + 3 * c];
                comp[c].h = s[7 + 3 * c] >> 4;
                comp[c].v = s[7 + 3 * c] & 15;
                comp[c].tq = s[8 + 3 * c];
            }
            have_sof = 1;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            { rc = E_UNSUPPORTED; goto done; }                                 /* lossless, differential, arithmetic */
        } else if (m == 0xDD) {
            restart = (s[0] << 8) | s[1];
        } else if (m == 0xE0 && sl >= 5 && !memcmp(s, "JFIF", 5)) {
            jfif = 1;
        } else if (m == 0xEE && sl >= 12 && !memcmp(s, "Adobe", 5)) {
            adobe_transform = s[11];
        } else if (m == 0xE1 && sl >= 14 && !memcmp(s, "Exif\0\0", 6)) {     /* orientation tag 0x0112 of IFD0 */
            const uint8_t* t = s + 6;
            const long tl = sl - 6;
            const int le = t[0] == 'I';
#define RD16(o) (le ? (t[o] | (t[(o) + 1] << 8)) : ((t[o] << 8) | t[(o) + 1]))
#define RD32(o) (le ? ((uint32_t)t[o] | ((uint32_t)t[(o) + 1] << 8) | ((uint32_t)t[(o) + 2] << 16) | ((uint32_t)t[(o) + 3] << 24)) \
                    : (((uint32_t)t[o] << 24) | ((uint32_t)t[(o) + 1] << 16) | ((uint32_t)t[(o) + 2] << 8) | (uint32_t)t[(o) + 3]))
            if (tl >= 8) {
                const long ifd = (long)RD32(4);
                if (ifd + 2 <= tl) {
                    const int ne = RD16(ifd);
                    for (int e = 0; e < ne && ifd + 2 + 12 * (e + 1) <= tl; ++e) {
                        const long o = ifd + 2 + 12 * e;
                        if (RD16(o) == 0x0112) orientation = RD16(o + 8);
                    }
                }
            }
        } else if (m == 0xDA) {                                   /* SOS */
            if (!have_sof) { rc = E_FORMAT; goto done; }
            if (info) {
                info[0] = W; info[1] = H; info[2] = ncomp; info[3] = progressive; info[4] = restart;
                for (int c = 0; c < 3; ++c) { info[5 + 2 * c] = comp[c].h; info[6 + 2 * c] = comp[c].v; }
                info[11] = orientation;
            }
            if (progressive) { rc = E_UNSUPPORTED; goto done; }
            if (s[0] == ncomp) {                                  /* one scan that interleaves every component */
                for (int c = 0; c < ncomp; ++c) {
                    if (s[1 + 2 * c] != comp[c].id) { rc = E_UNSUPPORTED; goto done; }
                    comp[c].td = s[2 + 2 * c] >> 4;
                    comp[c].ta = s[2 + 2 * c] & 15;
                }
            } else if (s[0] == 1 && ncomp == 3) {                 /* a non-interleaved scan of one component */
                for (int c = 0; c < 3; ++c)
                    if (s[1] == comp[c].id) scan_comp = c;
                if (scan_comp < 0) { rc = E_FORMAT; goto done; }
                comp[scan_comp].td = s[2] >> 4;
                comp[scan_comp].ta = s[2] & 15;
            } else {
                { rc = E_UNSUPPORTED; goto done; }                             /* partial interleaves (Y, then Cb + Cr together) */
            }
            if (s[1 + 2 * s[0]] != 0 || s[2 + 2 * s[0]] != 63) { rc = E_UNSUPPORTED; goto done; }     /* spectral selection = a progressive scan */
            if (ncomp == 3) {
                /* colour space as jdapimin.c default_decompress_parms decides it: YCbCr only here */
                int ycc = 1;
                if (!jfif && adobe_transform == 0) ycc = 0;
                if (!jfif && adobe_transform < 0 && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') ycc = 0;
                if (!ycc) { rc = E_UNSUPPORTED; goto done; }
            }
            if (!out) { rc = 0; goto done; }
            p += len;
            saw_sos = 1;
            break;
        }
        p += len;
    }
    if (!have_sof || !saw_sos || p >= n) { rc = E_FORMAT; goto done; }
    }

    if (!geometry_done) {
    for (int c = 0; c < ncomp; ++c) { if (comp[c].h > hmax) hmax = comp[c].h; if (comp[c].v > vmax) vmax = comp[c].v; }
    if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }      /* a single-component scan is never interleaved */
    else {
        if (comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1) { rc = E_UNSUPPORTED; goto done; }
        if (!((comp[0].h == 1 && comp[0].v == 1) || (comp[0].h == 2 && comp[0].v == 1) || (comp[0].h == 2 && comp[0].v == 2) ||
              (comp[0].h == 1 && comp[0].v == 2) || (comp[0].h == 4 && comp[0].v == 1)))     /* 4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1 */
            { rc = E_UNSUPPORTED; goto done; }
    }
    const int fmx = (W + 8 * hmax - 1) / (8 * hmax), fmy = (H + 8 * vmax - 1) / (8 * vmax);
    for (int c = 0; c < ncomp; ++c) {
        comp[c].bw = fmx * comp[c].h;
        comp[c].bh = fmy * comp[c].v;
        comp[c].pw = comp[c].bw * 8;
        comp[c].ph = comp[c].bh * 8;
        comp[c].rw = (W * comp[c].h + hmax - 1) / hmax;
        comp[c].rh = (H * comp[c].v + vmax - 1) / vmax;
    }
    for (int c = 0; c < ncomp; ++c) comp[c].plane = (uint8_t*)calloc((size_t)comp[c].pw, comp[c].ph);
    geometry_done = 1;
    }
    for (int c = 0; c < ncomp; ++c)
        if ((scan_comp < 0 || scan_comp == c) && (!dc[comp[c].td].present || !ac[comp[c].ta].present)) { rc = E_FORMAT; goto done; }
    if (scan_comp >= 0 ? decoded[scan_comp] : (decoded[0] || decoded[1] || decoded[2])) { rc = E_FORMAT; goto done; }   /* a component twice */

    /* MCU grid of this scan: the frame's for an interleaved scan; the component's own blocks, ceil(samples / 8) each way and ONE
       block per MCU, for a non-interleaved one (its restart interval counts those) */
    const int mw = 8 * hmax, mh = 8 * vmax;
    const int mx = scan_comp < 0 ? (W + mw - 1) / mw : (comp[scan_comp].rw + 7) / 8;
    const int my = scan_comp < 0 ? (H + mh - 1) / mh : (comp[scan_comp].rh + 7) / 8;
    const int c_lo = scan_comp < 0 ? 0 : scan_comp, c_hi = scan_comp < 0 ? ncomp : scan_comp + 1;
    for (int c = c_lo; c < c_hi; ++c) comp[c].pred = 0;
    Bits b = {buf + p, buf + n, 0, 0, 0};
    int todo = restart;
    for (int m = 0; m < mx * my && rc == 0; ++m) {
        if (restart && todo == 0) {                               /* RSTn: byte align, skip the marker, reset predictions */
            const uint8_t* q = b.p;
            while (q + 1 < b.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
            if (q + 1 >= b.end) { rc = E_FORMAT; break; }
            b.p = q + 2;
            b.acc = 0; b.n = 0; b.marker = 0;
            for (int c = c_lo; c < c_hi; ++c) comp[c].pred = 0;
            todo = restart;
        }
        const int mcx = m % mx, mcy = m / mx;
        for (int c = c_lo; c < c_hi && rc == 0; ++c) {
            const int bh_ = scan_comp < 0 ? comp[c].h : 1, bv_ = scan_comp < 0 ? comp[c].v : 1;
            for (int by = 0; by < bv_ && rc == 0; ++by)
                for (int bx = 0; bx < bh_; ++bx) {
                    int16_t coef[64];
                    memset(coef, 0, sizeof coef);
                    int s = decode_sym(&b, &dc[comp[c].td]);
                    if (s < 0 || s > 15) { rc = E_FORMAT; break; }
                    comp[c].pred += extend(getbits(&b, s), s);
                    coef[0] = (int16_t)comp[c].pred;
                    for (int k = 1; k < 64;) {
                        const int rs = decode_sym(&b, &ac[comp[c].ta]);
                        if (rs < 0) { rc = E_FORMAT; break; }
                        const int r = rs >> 4;
                        s = rs & 15;
                        if (s == 0) {
                            if (r != 15) break;                   /* EOB */
                            k += 16;                              /* ZRL */
                            continue;
                        }
                        k += r;
                        if (k > 63) { rc = E_FORMAT; break; }
                        coef[ZIGZAG[k]] = (int16_t)extend(getbits(&b, s), s);
                        ++k;
                    }
                    if (rc) break;
                    const int row = (mcy * bv_ + by) * 8, col = (mcx * bh_ + bx) * 8;
                    idct_islow(coef, qt[comp[c].tq], comp[c].plane + (size_t)row * comp[c].pw + col, comp[c].pw);
                }
        }
        if (restart) --todo;
    }
    if (rc == 0) {
        for (int c = c_lo; c < c_hi; ++c) decoded[c] = 1;
        int all = 1;
        for (int c = 0; c < ncomp; ++c) all = all && decoded[c];
        if (!all) {
            /* the next marker that is not RSTn ends this scan's entropy-coded data: more tables / the next scan follow */
            long q = p;
            while (q + 1 < n && !(buf[q] == 0xFF && buf[q + 1] != 0 && buf[q + 1] != 0xFF && !(buf[q + 1] >= 0xD0 && buf[q + 1] <= 0xD7))) ++q;
            if (q + 1 >= n || buf[q + 1] == 0xD9) { rc = E_FORMAT; goto done; }          /* EOI before every component came */
            p = q;
            goto next_scan;
        }
    }
    if (rc == 0) {
        if (!want_color) {
            for (int y = 0; y < H; ++y) memcpy(out + (size_t)y * W, comp[0].plane + (size_t)y * comp[0].pw, W);
        } else if (ncomp == 1) {
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const uint8_t v = comp[0].plane[(size_t)y * comp[0].pw + x];
                    uint8_t* o = out + ((size_t)y * W + x) * 3;
                    o[0] = o[1] = o[2] = v;
                }
        } else {
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const int Y = comp[0].plane[(size_t)y * comp[0].pw + x];
                    const int cb = upsampled(&comp[1], hmax, vmax, x, y) - 128, cr = upsampled(&comp[2], hmax, vmax, x, y) - 128;
                    /* jdcolor.c build_ycc_rgb_table: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802,
                       FIX(0.34414) = 22554, ONE_HALF = 32768; arithmetic right shifts */
                    const int r = Y + ((91881 * cr + 32768) >> 16);
                    const int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
                    const int bl = Y + ((116130 * cb + 32768) >> 16);
                    uint8_t* o = out + ((size_t)y * W + x) * 3;
                    o[0] = (uint8_t)clampi(r, 0, 255);
                    o[1] = (uint8_t)clampi(g, 0, 255);
                    o[2] = (uint8_t)clampi(bl, 0, 255);
                }
        }
    }
  done:
    for (int c = 0; c < ncomp; ++c) free(comp[c].plane);
    return rc;
}

int jpegref_info(const uint8_t* buf, long n, int* info12) { return parse_and_decode(buf, n, 0, 0, info12); }
/* out: H*W bytes (want_color = 0: the luma plane, what IMREAD_GRAYSCALE returns) or H*W*3 RGB bytes (want_color = 1) */
int jpegref_decode(const uint8_t* buf, long n, int want_color, uint8_t* out) { return parse_and_decode(buf, n, want_color, out, 0); }
