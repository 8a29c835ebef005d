/*
 * snappy_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar C restatement of the Snappy block format as used by the Hap
 * reference through snappy-c.h (call sites /root/reference/source/hap.c:313,
 * 453, 612, 813, 890, 899).  google/snappy is a third-party dependency that
 * the reference does not vendor or pin; the behaviour restated here is that
 * of libsnappy 1.1.8 (the version available to build oracle/_ref), from the
 * published format description (format_description.txt) and the published
 * structure of its compressor:
 *
 *   stream  = varint32(uncompressed length) , element*
 *   element = literal | copy-1 | copy-2 | copy-4          (tag & 3 = 0,1,2,3)
 *
 *   compressor: input is cut into independent 64 KiB fragments; per fragment a
 *   zeroed u16 hash table of min(2^14, next pow2 >= fragment size, >= 256)
 *   entries, hash = (load32 * 0x1e35a7bd) >> (32 - log2(table)), greedy
 *   first-match with the "skip" heuristic (probe stride grows by 1 every 32
 *   misses), 15-byte input margin, copies emitted as copy-1 when len < 12 and
 *   offset < 2048, otherwise copy-2 pieces of <= 64.
 *
 * Pinning: tests/test_oracle_pinning.py requires osnappy_compress to be
 * BYTE-IDENTICAL to libsnappy 1.1.8's snappy_compress, and osnappy_uncompress
 * to agree with snappy_uncompress on output bytes and status for valid,
 * truncated and corrupted streams.
 */
#include "oracle.h"
#include <string.h>
#include <stdlib.h>

#define FRAGMENT_BYTES 65536u
#define MAX_TABLE_ENTRIES 16384u
#define INPUT_MARGIN 15u

static uint32_t ld32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

size_t osnappy_max_compressed_length(size_t n)
{
    return 32 + n + n / 6;
}

/* ------------------------------------------------------------------ */
/* decoder                                                              */
/* ------------------------------------------------------------------ */

/* varint32 exactly as a 32-bit length prefix: at most 5 bytes, the fifth may
 * only carry the top 4 bits. Returns header size or 0 when malformed. */
static unsigned read_varint32(const uint8_t *in, size_t n, uint32_t *value)
{
    uint32_t v = 0;
    unsigned i;
    for (i = 0; i < 5; i++) {
        uint8_t b;
        if (i >= n)
            return 0;
        b = in[i];
        if (i == 4 && b >= 16)
            return 0;
        v |= (uint32_t)(b & 0x7f) << (7 * i);
        if (!(b & 0x80)) {
            *value = v;
            return i + 1;
        }
    }
    return 0;
}

int osnappy_uncompressed_length(const uint8_t *in, size_t n, size_t *result)
{
    uint32_t v;
    if (!read_varint32(in, n, &v))
        return OSNAPPY_INVALID_INPUT;
    *result = v;
    return OSNAPPY_OK;
}

int osnappy_uncompress(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len)
{
    uint32_t expect;
    size_t ip, op = 0;
    unsigned hdr = read_varint32(in, n, &expect);
    if (!hdr)
        return OSNAPPY_INVALID_INPUT;
    if (*out_len < expect)
        return OSNAPPY_BUFFER_TOO_SMALL;
    ip = hdr;
    while (ip < n) {
        uint8_t tag = in[ip];
        unsigned kind = tag & 3u;
        if (kind == 0) {
            size_t len = (size_t)(tag >> 2) + 1;
            ip += 1;
            if (len > 60) {
                unsigned extra = (unsigned)len - 60, k;
                uint32_t v = 0;
                if (n - ip < extra)
                    return OSNAPPY_INVALID_INPUT;
                for (k = 0; k < extra; k++)
                    v |= (uint32_t)in[ip + k] << (8 * k);
                ip += extra;
                len = (size_t)v + 1;
            }
            if (len > n - ip || len > expect - op)
                return OSNAPPY_INVALID_INPUT;
            memcpy(out + op, in + ip, len);
            ip += len;
            op += len;
        } else {
            size_t len, off, k;
            if (kind == 1) {
                if (n - ip < 2)
                    return OSNAPPY_INVALID_INPUT;
                len = 4 + ((tag >> 2) & 7u);
                off = ((size_t)(tag >> 5) << 8) | in[ip + 1];
                ip += 2;
            } else if (kind == 2) {
                if (n - ip < 3)
                    return OSNAPPY_INVALID_INPUT;
                len = (size_t)(tag >> 2) + 1;
                off = (size_t)in[ip + 1] | ((size_t)in[ip + 2] << 8);
                ip += 3;
            } else {
                if (n - ip < 5)
                    return OSNAPPY_INVALID_INPUT;
                len = (size_t)(tag >> 2) + 1;
                off = ld32(in + ip + 1);
                ip += 5;
            }
            if (off == 0 || off > op || len > expect - op)
                return OSNAPPY_INVALID_INPUT;
            /* byte-serial so that overlapping copies replicate the pattern */
            for (k = 0; k < len; k++)
                out[op + k] = out[op + k - off];
            op += len;
        }
    }
    if (op != expect)
        return OSNAPPY_INVALID_INPUT;
    *out_len = expect;
    return OSNAPPY_OK;
}

/* ------------------------------------------------------------------ */
/* compressor                                                           */
/* ------------------------------------------------------------------ */

static uint8_t *put_literal(uint8_t *op, const uint8_t *src, size_t len)
{
    size_t n = len - 1;
    if (n < 60) {
        *op++ = (uint8_t)(n << 2);
    } else {
        uint8_t *tagp = op++;
        unsigned count = 0;
        while (n > 0) {
            *op++ = (uint8_t)(n & 0xff);
            n >>= 8;
            count++;
        }
        *tagp = (uint8_t)((59 + count) << 2);
    }
    memcpy(op, src, len);
    return op + len;
}

static uint8_t *put_copy_upto64(uint8_t *op, size_t off, size_t len)
{
    if (len < 12 && off < 2048) {
        *op++ = (uint8_t)(1 | ((len - 4) << 2) | ((off >> 8) << 5));
        *op++ = (uint8_t)(off & 0xff);
    } else {
        *op++ = (uint8_t)(2 | ((len - 1) << 2));
        *op++ = (uint8_t)(off & 0xff);
        *op++ = (uint8_t)(off >> 8);
    }
    return op;
}

static uint8_t *put_copy(uint8_t *op, size_t off, size_t len)
{
    /* long matches: 64-byte pieces, leaving a tail that is never < 4 */
    while (len >= 68) {
        op = put_copy_upto64(op, off, 64);
        len -= 64;
    }
    if (len > 64) {
        op = put_copy_upto64(op, off, 60);
        len -= 60;
    }
    return put_copy_upto64(op, off, len);
}

static uint32_t hash32(uint32_t v, int shift)
{
    return (v * 0x1e35a7bdu) >> shift;
}

static uint8_t *compress_fragment(const uint8_t *base, size_t size, uint8_t *op,
                                  uint16_t *table, unsigned table_entries)
{
    const uint8_t *ip = base, *end = base + size, *pending = base;
    int shift = 32;
    unsigned t = table_entries;
    while (t > 1) {
        t >>= 1;
        shift--;
    }
    if (size >= INPUT_MARGIN) {
        const uint8_t *limit = end - INPUT_MARGIN;
        uint32_t next_h;
        ip++;
        next_h = hash32(ld32(ip), shift);
        for (;;) {
            uint32_t skip = 32;
            const uint8_t *probe = ip, *cand;
            /* scan forward for a 4-byte match, probing ever more sparsely */
            do {
                uint32_t h = next_h, step = skip >> 5;
                ip = probe;
                skip += step;
                probe = ip + step;
                if (probe > limit)
                    goto tail;
                next_h = hash32(ld32(probe), shift);
                cand = base + table[h];
                table[h] = (uint16_t)(ip - base);
            } while (ld32(ip) != ld32(cand));

            op = put_literal(op, pending, (size_t)(ip - pending));

            /* emit copies back-to-back while the position right after a match
             * immediately matches again */
            for (;;) {
                const uint8_t *s1 = cand + 4, *s2 = ip + 4;
                size_t matched = 4, off = (size_t)(ip - cand);
                uint32_t hprev, hcur;
                while (s2 < end && *s1 == *s2) {
                    s1++;
                    s2++;
                    matched++;
                }
                ip += matched;
                op = put_copy(op, off, matched);
                pending = ip;
                if (ip >= limit)
                    goto tail;
                hprev = hash32(ld32(ip - 1), shift);
                table[hprev] = (uint16_t)(ip - base - 1);
                hcur = hash32(ld32(ip), shift);
                cand = base + table[hcur];
                table[hcur] = (uint16_t)(ip - base);
                if (ld32(ip) != ld32(cand))
                    break;
            }
            ip++;
            next_h = hash32(ld32(ip), shift);
        }
    }
tail:
    if (pending < end)
        op = put_literal(op, pending, (size_t)(end - pending));
    return op;
}

int osnappy_compress(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len)
{
    uint8_t *op = out;
    uint16_t *table;
    size_t done = 0;
    uint32_t v = (uint32_t)n;
    if (*out_len < osnappy_max_compressed_length(n))
        return OSNAPPY_BUFFER_TOO_SMALL;
    while (v >= 0x80) {
        *op++ = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    *op++ = (uint8_t)v;
    table = (uint16_t *)malloc(MAX_TABLE_ENTRIES * sizeof(uint16_t));
    if (!table)
        return OSNAPPY_INVALID_INPUT;
    while (done < n) {
        size_t frag = n - done < FRAGMENT_BYTES ? n - done : FRAGMENT_BYTES;
        unsigned entries = 256;
        while (entries < MAX_TABLE_ENTRIES && entries < frag)
            entries <<= 1;
        memset(table, 0, entries * sizeof(uint16_t));
        op = compress_fragment(in + done, frag, op, table, entries);
        done += frag;
    }
    free(table);
    *out_len = (size_t)(op - out);
    return OSNAPPY_OK;
}
