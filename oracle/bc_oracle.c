/*
 * bc_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED BY THE REFERENCE: /root/reference contains no RGBA->DXT
 * encoder (hap.h:82-104 takes already-compressed textures; the encoders live
 * in client codecs).  This file therefore DEFINES, in scalar integer C, the
 * block-compression algorithm the HIP kernels in hap_amd/csrc/bc_encode.hip
 * must reproduce bit-for-bit.  Block bit layouts follow the public S3TC
 * (DXT1/DXT5), RGTC1 and scaled YCoCg-DXT5 descriptions the reference cites
 * at documentation/HapVideoDRAFT.md:22-27 (restated in SURVEY.md App. C);
 * tests check the layouts independently by decoding with Pillow's DDS reader.
 *
 * Algorithm family: per-block bounding box with inset, covariance-sign
 * diagonal selection, endpoints rounded to 5:6:5, indices by projection onto
 * the endpoint segment (the rule of the fast real-time encoders this stage
 * stands in for: one dot product and one multiply per pixel instead of four
 * distance evaluations; within 0.05 dB of exhaustive nearest-entry search on
 * the test pictures, build with -DOBC_EXACT_NEAREST to compare).  Every
 * operation is integer and the fixed-point constants are part of the definition.
 *
 *   colour block (DXT1 / DXT5 colour half)
 *     lo,hi    = per-channel min,max of the 16 pixels
 *     cov_xg   = sum (2x-lo_x-hi_x)(2g-lo_g-hi_g)          x in {r,b}
 *     inset    = (hi-lo)>>4 ;  lo+=inset ; hi-=inset
 *     A        = (cov_rg<0 ? lo_r:hi_r, hi_g, cov_bg<0 ? lo_b:hi_b), B = the rest
 *     c0,c1    = max,min of pack565(A),pack565(B)  (so c0>c1 -> 4-colour mode)
 *     palette  = expand(c0), expand(c1), (2p0+p1)/3, (p0+2p1)/3   (floor)
 *     dir = p0-p1, len2 = |dir|^2, t_i = (pixel_i-p1).dir + len2/6 clamped to 0..len2+len2/6
 *     pos_i    = (t_i * floor(3*2^24/len2)) >> 24   (0..3: thirds of the segment from p1)
 *     index_i  = {1,3,2,0}[pos_i]
 *     c0==c1  -> all indices 0
 *
 *   alpha block (DXT5 alpha half, RGTC1, Y of YCoCg)
 *     a0 = hi ; a1 = lo   (the exact range, as stb_dxt / DirectXTex do for ramps: every pixel lies on the ramp)
 *     a0==a1 -> indices 0, else d = a0-a1, u = a0-a,
 *     ramp position r = (14u + max(d-6,0)) / 2d (floor), code = r==0?0 : r==7?1 : r+1
 *
 *   YCoCg: Y=(R+2G+B+2)>>2, Co=((R-B+1)>>1)+128, Cg=((-R+2G-B+2)>>2)+128   (1..256: no clamp per pixel -- 256 only
 *     for saturated primaries, and the 5:6:5 endpoints quantise 256 to their top code like 255)
 *     scale s = 4 if max|C-128|<=31, 2 if <=63, else 1 ; C' = (C-128)s+128
 *     2-D (Co',Cg') bounding box with the same diagonal/inset/565 rules,
 *     blue 5-bit field = s-1 in both endpoints (decodes to 0/8/24).
 */
#include "oracle.h"
#include <string.h>

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int clamp255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

static int quant5(int v) { int t = v * 31 + 128; return (t + (t >> 8)) >> 8; }
static int quant6(int v) { int t = v * 63 + 128; return (t + (t >> 8)) >> 8; }
static int expand5(int q) { return (q << 3) | (q >> 2); }
static int expand6(int q) { return (q << 2) | (q >> 4); }

static void store16(uint8_t *p, unsigned v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }
static void store32(uint8_t *p, uint32_t v) { store16(p, v & 0xFFFF); store16(p + 2, v >> 16); }

/* alpha-style block: 2 endpoint bytes + 48 bits of 3-bit codes */
static void alpha_block(const int a[16], uint8_t out[8])
{
    int lo = 255, hi = 0, i, a0, a1;
    uint64_t bits = 0;
    for (i = 0; i < 16; i++) {
        lo = imin(lo, a[i]);
        hi = imax(hi, a[i]);
    }
    a0 = hi;
    a1 = lo;
    if (a0 != a1) {
#ifdef OBC_EXACT_NEAREST
        int q[8], j;
        for (j = 0; j < 8; j++)
            q[j] = ((7 - j) * a0 + j * a1) / 7;
        for (i = 0; i < 16; i++) {
            int r = 0;
            for (j = 0; j < 7; j++)
                r += (2 * a[i] < q[j] + q[j + 1]);
            bits |= (uint64_t)(r == 0 ? 0 : r == 7 ? 1 : r + 1) << (3 * i);
        }
#else
        /* ramp position = the pixel's place on the ideal ramp a0 .. a1, rounded to the nearest of its 8 steps; the
           decoder's steps ((7 - j) a0 + j a1) / 7 are rounded DOWN, 3/7 on average, which moves the best thresholds by
           6/14 of a level: (14 u + max(d - 6, 0)) / 2d, in the fixed point below (one case in 33 000 rounds up) */
        const int d = a0 - a1, bias = d > 6 ? d - 6 : 0;
        const uint32_t m = (1u << 19) / (uint32_t)d + 1u;      /* (x * m) >> 20 = x / 2d, the kernel's fixed point */
        for (i = 0; i < 16; i++) {
            const int u = a0 - a[i];                             /* 0 .. d: r comes out in 0 .. 7 */
            const int r = (int)(((uint32_t)(14 * u + bias) * m) >> 20);
            bits |= (uint64_t)(r == 0 ? 0 : r == 7 ? 1 : r + 1) << (3 * i);
        }
#endif
    }
    out[0] = (uint8_t)a0;
    out[1] = (uint8_t)a1;
    for (i = 0; i < 6; i++)
        out[2 + i] = (uint8_t)(bits >> (8 * i));
}

/* 2-bit indices of 16 pixels for the 4-entry palette pal[0] (c0), pal[1] (c1), (2 pal0 + pal1) / 3, (pal0 + 2 pal1) / 3 */
static uint32_t pick_indices(const int px[16][3], const int pal[4][3], int channels)
{
    uint32_t idx = 0;
    int i, k, c;
#ifdef OBC_EXACT_NEAREST
    for (i = 0; i < 16; i++) {
        int best = 0, bestd = 0x7fffffff;
        for (k = 0; k < 4; k++) {
            int d = 0;
            for (c = 0; c < channels; c++) {
                int e = px[i][c] - pal[k][c];
                d += e * e;
            }
            if (d < bestd) {
                bestd = d;
                best = k;
            }
        }
        idx |= (uint32_t)best << (2 * i);
    }
#else
    /* The four entries lie on the segment pal1 .. pal0 at 0, 1/3, 2/3, 1: project the pixel onto it and round to the
       nearest third.  With dir = pal0 - pal1, len2 = |dir|^2, t = (pixel - pal1) . dir:
           pos = ((t + len2 / 6, clamped to 0 .. len2 + len2 / 6) * floor(3 * 2^24 / len2)) >> 24        (0 .. 3)
       -- fixed point on purpose: this IS the definition, the kernel evaluates the same integers -- and
       pos 3 -> index 0 (pal0), 0 -> 1 (pal1), 2 -> 2, 1 -> 3. */
    static const uint32_t code[4] = {1, 3, 2, 0};
    int dir[3] = {0, 0, 0}, len2 = 0, base = 0;
    uint32_t m24;
    (void)k;
    for (c = 0; c < channels; c++) {
        dir[c] = pal[0][c] - pal[1][c];
        len2 += dir[c] * dir[c];
        base += pal[1][c] * dir[c];
    }
    m24 = (uint32_t)(50331648u / (uint32_t)len2);
    for (i = 0; i < 16; i++) {
        int t = len2 / 6 - base;
        uint32_t pos;
        for (c = 0; c < channels; c++)
            t += px[i][c] * dir[c];
        t = t < 0 ? 0 : t > len2 + len2 / 6 ? len2 + len2 / 6 : t;
        pos = ((uint32_t)t * m24) >> 24;
        idx |= code[pos > 3 ? 3 : pos] << (2 * i);
    }
#endif
    return idx;
}

static void colour_block(const int px[16][3], uint8_t out[8])
{
    int lo[3] = {255, 255, 255}, hi[3] = {0, 0, 0}, i, c;
    int cov_rg = 0, cov_bg = 0, ea[3], eb[3];
    unsigned qa, qb, c0, c1;
    uint32_t idx = 0;
    for (i = 0; i < 16; i++)
        for (c = 0; c < 3; c++) {
            lo[c] = imin(lo[c], px[i][c]);
            hi[c] = imax(hi[c], px[i][c]);
        }
    for (i = 0; i < 16; i++) {
        int dr = 2 * px[i][0] - lo[0] - hi[0];
        int dg = 2 * px[i][1] - lo[1] - hi[1];
        int db = 2 * px[i][2] - lo[2] - hi[2];
        cov_rg += dr * dg;
        cov_bg += db * dg;
    }
    for (c = 0; c < 3; c++) {
        int inset = (hi[c] - lo[c]) >> 4;
        lo[c] += inset;
        hi[c] -= inset;
    }
    ea[0] = cov_rg < 0 ? lo[0] : hi[0];
    eb[0] = cov_rg < 0 ? hi[0] : lo[0];
    ea[1] = hi[1];
    eb[1] = lo[1];
    ea[2] = cov_bg < 0 ? lo[2] : hi[2];
    eb[2] = cov_bg < 0 ? hi[2] : lo[2];
    qa = (unsigned)(quant5(ea[0]) << 11 | quant6(ea[1]) << 5 | quant5(ea[2]));
    qb = (unsigned)(quant5(eb[0]) << 11 | quant6(eb[1]) << 5 | quant5(eb[2]));
    c0 = qa > qb ? qa : qb;
    c1 = qa > qb ? qb : qa;
    if (c0 != c1) {
        int pal[4][3];
        pal[0][0] = expand5(c0 >> 11); pal[0][1] = expand6((c0 >> 5) & 63); pal[0][2] = expand5(c0 & 31);
        pal[1][0] = expand5(c1 >> 11); pal[1][1] = expand6((c1 >> 5) & 63); pal[1][2] = expand5(c1 & 31);
        for (c = 0; c < 3; c++) {
            pal[2][c] = (2 * pal[0][c] + pal[1][c]) / 3;
            pal[3][c] = (pal[0][c] + 2 * pal[1][c]) / 3;
        }
        idx = pick_indices(px, pal, 3);
    }
    store16(out, c0);
    store16(out + 2, c1);
    store32(out + 4, idx);
}

static void ycocg_colour_block(const int co[16], const int cg[16], uint8_t out[8])
{
    int lo_o = 256, hi_o = 0, lo_g = 256, hi_g = 0, i, m, s, cov = 0, ins;
    int px[16][3], ao, ag, bo, bg;
    unsigned qa, qb, c0, c1;
    uint32_t idx = 0;
    for (i = 0; i < 16; i++) {
        lo_o = imin(lo_o, co[i]); hi_o = imax(hi_o, co[i]);
        lo_g = imin(lo_g, cg[i]); hi_g = imax(hi_g, cg[i]);
    }
    m = imax(imax(128 - lo_o, hi_o - 128), imax(128 - lo_g, hi_g - 128));
    s = m <= 31 ? 4 : m <= 63 ? 2 : 1;
    for (i = 0; i < 16; i++)
        cov += (2 * co[i] - lo_o - hi_o) * (2 * cg[i] - lo_g - hi_g);
    lo_o = (lo_o - 128) * s + 128; hi_o = (hi_o - 128) * s + 128;
    lo_g = (lo_g - 128) * s + 128; hi_g = (hi_g - 128) * s + 128;
    ins = (hi_o - lo_o) >> 4; lo_o += ins; hi_o -= ins;
    ins = (hi_g - lo_g) >> 4; lo_g += ins; hi_g -= ins;
    ao = hi_o; bo = lo_o;
    ag = cov < 0 ? lo_g : hi_g;
    bg = cov < 0 ? hi_g : lo_g;
    qa = (unsigned)(quant5(ao) << 11 | quant6(ag) << 5 | (s - 1));
    qb = (unsigned)(quant5(bo) << 11 | quant6(bg) << 5 | (s - 1));
    c0 = qa > qb ? qa : qb;
    c1 = qa > qb ? qb : qa;
    if (c0 != c1) {
        int pal[4][3], c;
        pal[0][0] = expand5(c0 >> 11); pal[0][1] = expand6((c0 >> 5) & 63); pal[0][2] = 0;
        pal[1][0] = expand5(c1 >> 11); pal[1][1] = expand6((c1 >> 5) & 63); pal[1][2] = 0;
        for (c = 0; c < 3; c++) {
            pal[2][c] = (2 * pal[0][c] + pal[1][c]) / 3;
            pal[3][c] = (pal[0][c] + 2 * pal[1][c]) / 3;
        }
        for (i = 0; i < 16; i++) {
            px[i][0] = (co[i] - 128) * s + 128;
            px[i][1] = (cg[i] - 128) * s + 128;
            px[i][2] = 0;
        }
        idx = pick_indices(px, pal, 2);
    }
    store16(out, c0);
    store16(out + 2, c1);
    store32(out + 4, idx);
}

static const uint8_t *pixel(const uint8_t *rgba, size_t row_bytes, unsigned x, unsigned y)
{
    return rgba + (size_t)y * row_bytes + 4u * x;
}

void obc_encode_dxt1(const uint8_t *rgba, unsigned w, unsigned h, size_t row_bytes, uint8_t *out)
{
    unsigned bx, by, i;
    for (by = 0; by < h / 4; by++)
        for (bx = 0; bx < w / 4; bx++) {
            int px[16][3];
            for (i = 0; i < 16; i++) {
                const uint8_t *p = pixel(rgba, row_bytes, bx * 4 + (i & 3), by * 4 + (i >> 2));
                px[i][0] = p[0]; px[i][1] = p[1]; px[i][2] = p[2];
            }
            colour_block(px, out);
            out += 8;
        }
}

void obc_encode_dxt5(const uint8_t *rgba, unsigned w, unsigned h, size_t row_bytes, uint8_t *out)
{
    unsigned bx, by, i;
    for (by = 0; by < h / 4; by++)
        for (bx = 0; bx < w / 4; bx++) {
            int px[16][3], a[16];
            for (i = 0; i < 16; i++) {
                const uint8_t *p = pixel(rgba, row_bytes, bx * 4 + (i & 3), by * 4 + (i >> 2));
                px[i][0] = p[0]; px[i][1] = p[1]; px[i][2] = p[2]; a[i] = p[3];
            }
            alpha_block(a, out);
            colour_block(px, out + 8);
            out += 16;
        }
}

void obc_encode_rgtc1_alpha(const uint8_t *rgba, unsigned w, unsigned h, size_t row_bytes, uint8_t *out)
{
    unsigned bx, by, i;
    for (by = 0; by < h / 4; by++)
        for (bx = 0; bx < w / 4; bx++) {
            int a[16];
            for (i = 0; i < 16; i++)
                a[i] = pixel(rgba, row_bytes, bx * 4 + (i & 3), by * 4 + (i >> 2))[3];
            alpha_block(a, out);
            out += 8;
        }
}

void obc_encode_ycocg_dxt5(const uint8_t *rgba, unsigned w, unsigned h, size_t row_bytes, uint8_t *out)
{
    unsigned bx, by, i;
    for (by = 0; by < h / 4; by++)
        for (bx = 0; bx < w / 4; bx++) {
            int y[16], co[16], cg[16];
            for (i = 0; i < 16; i++) {
                const uint8_t *p = pixel(rgba, row_bytes, bx * 4 + (i & 3), by * 4 + (i >> 2));
                int r = p[0], g = p[1], b = p[2];
                y[i] = (r + 2 * g + b + 2) >> 2;
                co[i] = ((r - b + 1) >> 1) + 128;                  /* 1 .. 256 (256: R = 255, B = 0) */
                cg[i] = ((-r + 2 * g - b + 2) >> 2) + 128;         /* 1 .. 256 (256: G = 255, R = B = 0) */
            }
            alpha_block(y, out);
            ycocg_colour_block(co, cg, out + 8);
            out += 16;
        }
}

/* ------------------------------------------------------------------ */
/* decoders (sanity only)                                               */
/* ------------------------------------------------------------------ */

static void decode_colour(const uint8_t *b, int pal[4][3], uint32_t *idx, int dxt1_modes)
{
    unsigned c0 = b[0] | b[1] << 8, c1 = b[2] | b[3] << 8;
    int c;
    pal[0][0] = expand5(c0 >> 11); pal[0][1] = expand6((c0 >> 5) & 63); pal[0][2] = expand5(c0 & 31);
    pal[1][0] = expand5(c1 >> 11); pal[1][1] = expand6((c1 >> 5) & 63); pal[1][2] = expand5(c1 & 31);
    for (c = 0; c < 3; c++) {
        if (!dxt1_modes || c0 > c1) {
            pal[2][c] = (2 * pal[0][c] + pal[1][c]) / 3;
            pal[3][c] = (pal[0][c] + 2 * pal[1][c]) / 3;
        } else {
            pal[2][c] = (pal[0][c] + pal[1][c]) / 2;
            pal[3][c] = 0;
        }
    }
    *idx = b[4] | b[5] << 8 | b[6] << 16 | (uint32_t)b[7] << 24;
}

static void decode_alpha(const uint8_t *b, int out[16])
{
    int a0 = b[0], a1 = b[1], v[8], i;
    uint64_t bits = 0;
    for (i = 0; i < 6; i++)
        bits |= (uint64_t)b[2 + i] << (8 * i);
    v[0] = a0; v[1] = a1;
    if (a0 > a1) {
        for (i = 1; i < 7; i++)
            v[i + 1] = ((7 - i) * a0 + i * a1) / 7;
    } else {
        for (i = 1; i < 5; i++)
            v[i + 1] = ((5 - i) * a0 + i * a1) / 5;
        v[6] = 0; v[7] = 255;
    }
    for (i = 0; i < 16; i++)
        out[i] = v[(bits >> (3 * i)) & 7];
}

void obc_decode_dxt1(const uint8_t *blocks, unsigned w, unsigned h, uint8_t *rgba)
{
    unsigned bx, by, i;
    for (by = 0; by < h / 4; by++)
        for (bx = 0; bx < w / 4; bx++, blocks += 8) {
            int pal[4][3];
            uint32_t idx;
            decode_colour(blocks, pal, &idx, 1);
            for (i = 0; i < 16; i++) {
                uint8_t *p = rgba + ((size_t)(by * 4 + (i >> 2)) * w + bx * 4 + (i & 3)) * 4;
                int k = (idx >> (2 * i)) & 3;
                p[0] = (uint8_t)pal[k][0]; p[1] = (uint8_t)pal[k][1]; p[2] = (uint8_t)pal[k][2]; p[3] = 255;
            }
        }
}

void obc_decode_dxt5(const uint8_t *blocks, unsigned w, unsigned h, uint8_t *rgba)
{
    unsigned bx, by, i;
    for (by = 0; by < h / 4; by++)
        for (bx = 0; bx < w / 4; bx++, blocks += 16) {
            int pal[4][3], a[16];
            uint32_t idx;
            decode_alpha(blocks, a);
            decode_colour(blocks + 8, pal, &idx, 0);
            for (i = 0; i < 16; i++) {
                uint8_t *p = rgba + ((size_t)(by * 4 + (i >> 2)) * w + bx * 4 + (i & 3)) * 4;
                int k = (idx >> (2 * i)) & 3;
                p[0] = (uint8_t)pal[k][0]; p[1] = (uint8_t)pal[k][1]; p[2] = (uint8_t)pal[k][2];
                p[3] = (uint8_t)a[i];
            }
        }
}

void obc_decode_ycocg_dxt5(const uint8_t *blocks, unsigned w, unsigned h, uint8_t *rgba)
{
    unsigned bx, by, i;
    for (by = 0; by < h / 4; by++)
        for (bx = 0; bx < w / 4; bx++, blocks += 16) {
            int pal[4][3], y[16];
            uint32_t idx;
            decode_alpha(blocks, y);
            decode_colour(blocks + 8, pal, &idx, 0);
            for (i = 0; i < 16; i++) {
                uint8_t *p = rgba + ((size_t)(by * 4 + (i >> 2)) * w + bx * 4 + (i & 3)) * 4;
                int k = (idx >> (2 * i)) & 3;
                int s = (pal[k][2] >> 3) + 1;
                /* floor division toward -inf keeps the inverse symmetric */
                int co = pal[k][0] - 128, cg = pal[k][1] - 128;
                co = co >= 0 ? co / s : -((-co) / s);
                cg = cg >= 0 ? cg / s : -((-cg) / s);
                p[0] = (uint8_t)clamp255(y[i] + co - cg);
                p[1] = (uint8_t)clamp255(y[i] + cg);
                p[2] = (uint8_t)clamp255(y[i] - co - cg);
                p[3] = 255;
            }
        }
}

void obc_decode_rgtc1(const uint8_t *blocks, unsigned w, unsigned h, uint8_t *plane)
{
    unsigned bx, by, i;
    for (by = 0; by < h / 4; by++)
        for (bx = 0; bx < w / 4; bx++, blocks += 8) {
            int a[16];
            decode_alpha(blocks, a);
            for (i = 0; i < 16; i++)
                plane[(size_t)(by * 4 + (i >> 2)) * w + bx * 4 + (i & 3)] = (uint8_t)a[i];
        }
}
