/*
 * field_stream_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU definition of the block-per-lane "field stream" compressor (hap_amd/csrc/snappy_compress_blocks.hip),
 * the GPU stage that stands where the reference calls snappy_compress (/root/reference/source/hap.c:453)
 * for block textures.  The reference's own compressor is libsnappy (not vendored; oracle/snappy_oracle.c
 * restates it); Snappy encodings are not unique, so parity for an ENCODER is defined two ways:
 *   1. the reference decoder reproduces the input from what the kernel wrote (tests/test_gpu_parity.py), and
 *   2. the kernel's bytes are exactly the bytes of this scalar definition (same tests), which makes every
 *      matching rule below a checked statement instead of a comment.
 * The stream this writes is ordinary Snappy (literal / copy-1 / copy-2 elements) obeying the extra promises of the
 * private fragment table version 4 (include/hap_gpu.h, snappy_decode_fields.hip): no element crosses a 128-byte
 * half-tile, every element starts and ends on a block-field boundary, copy offsets are whole blocks; the table lists
 * the compressed and the produced bytes of 64 groups of equally many elements, one group per decoder lane, and the
 * number of elements.
 *
 * Layouts (the "fields per block" nibble of the table): 4 = DXT5 / YCoCg-DXT5 [2, 6, 4, 4]; 2 = DXT1 [4, 4], two blocks
 * per unit; 6 = RGTC1 [2, 6], two blocks per unit; 8 = opaque 16-byte blocks (BC7, BC6H) taken as four dwords, copy
 * distances in whole blocks (the size-for-speed option HAPGPU_ENCODE_COARSE_MATCHES).
 *
 * Algorithm, per fragment of <= 8 KiB (units of 16 bytes = 4 fields; 32 fields = one half-tile):
 *   a. field i matches at distance d (1..4 blocks) when its bytes equal the same field d blocks back;
 *      a 2-byte field only counts next to a neighbouring field that matches at the same distance, and never makes a
 *      copy of its own;
 *   b. index fields (6-byte / 4-byte index words) look up the most recent earlier block with the same value in a
 *      direct-mapped table that is updated after every 64 units ("step") with the fields that differ from the same
 *      field one block back, most recent block wins;
 *   c. per half-tile: copies are chosen left to right -- at the first uncovered matching position the NEAREST
 *      matching distance starts a copy that runs to the end of its match ("sticky"), the next one starts where it
 *      ends; a copy longer than 64 bytes is cut at field 16 (byte 64); fields no distance covers are single-field copies from their
 *      table candidate if they have one, literals otherwise; a copy of one 4-byte field between two literal fields becomes
 *      literal bytes too (one byte more, two elements fewer);
 *   d. literal runs take a 1-byte header up to 60 bytes and (0xF0, len - 1) above; copies are copy-1 when shorter
 *      than 12 bytes and nearer than 2048 bytes, else copy-2.
 */
#include <stdint.h>
#include <string.h>

typedef struct {
    unsigned fo[4], fs[4];      /* field offsets / sizes inside a 16-byte unit */
    unsigned block;             /* bytes per block: copy distances are multiples of it */
    unsigned cls[4];            /* table class of a field (0: not looked up) */
} ofs_layout;

static const ofs_layout k_layout4 = {{0, 2, 8, 12}, {2, 6, 4, 4}, 16, {0, 1, 0, 3}};   /* DXT5 / YCoCg-DXT5 */
static const ofs_layout k_layout2 = {{0, 4, 8, 12}, {4, 4, 4, 4}, 8, {0, 1, 0, 1}};    /* DXT1, two blocks per unit */
static const ofs_layout k_layout6 = {{0, 2, 8, 10}, {2, 6, 2, 6}, 8, {0, 1, 0, 1}};    /* RGTC1, two blocks per unit */
static const ofs_layout k_layout8 = {{0, 4, 8, 12}, {4, 4, 4, 4}, 16, {0, 1, 0, 3}};   /* opaque 16-byte blocks (BC7, BC6H): four dwords */

#define OFS_TABLE_BITS 9u
#define OFS_STEP_UNITS 64u
#define OFS_DISTANCES 4u
#define OFS_GROUP_TABLE_BYTES 196u

static uint64_t field_value(const uint8_t *p, unsigned size)
{
    uint64_t v = 0;
    memcpy(&v, p, size);          /* little endian hosts only (test infrastructure) */
    return v;
}

static unsigned table_slot(uint64_t value, unsigned cls)
{
    const uint32_t lo = (uint32_t)value, hi = (uint32_t)(value >> 32);
    const uint32_t z = lo ^ ((hi << 13) | (hi >> 19)) ^ (cls << 29);
    return (uint32_t)(z * 0x9E3779B1u) >> (32u - OFS_TABLE_BITS);
}

/* byte position of field i of a half-tile (i = 0..32) */
static unsigned fpos(const ofs_layout *L, unsigned i) { return (i >> 2) * 16u + (i < 32u ? L->fo[i & 3u] : 0u); }

unsigned ofs_compress_fragment(const uint8_t *src, unsigned n, unsigned layout, unsigned window_bytes, uint8_t *out,
                               uint8_t *group_table)
{
    const ofs_layout *L = layout == 4u ? &k_layout4 : layout == 2u ? &k_layout2 : layout == 8u ? &k_layout8 : &k_layout6;
    static uint8_t eq[OFS_DISTANCES][2048];
    static uint16_t hd[2048];                      /* table candidate of a field: distance in blocks, 0 = none */
    uint64_t table[1u << OFS_TABLE_BITS];
    const unsigned units = (n + 15u) / 16u, fields = units * 4u;
    unsigned produced = 0;
    static uint16_t element_at[2048 + 1];          /* stream offset of every element, in order */
    static uint16_t element_out[2048 + 1];         /* ... and the position of its first output byte */
    unsigned elements = 0;

    memset(group_table, 0, OFS_GROUP_TABLE_BYTES);
    if (n == 0 || n > 8192u || (n % L->block) != 0u)
        return 0;
    /* a. fixed distances */
    for (unsigned i = 0; i < fields; i++) {
        const unsigned k = i & 3u, pos = (i >> 2) * 16u + L->fo[k];
        const int valid = pos + L->fs[k] <= n;
        for (unsigned d = 1; d <= OFS_DISTANCES; d++)
            eq[d - 1][i] = valid && pos >= d * L->block && memcmp(src + pos, src + pos - d * L->block, L->fs[k]) == 0;
    }
    /* b. table candidates, step by step */
    memset(table, 0, sizeof table);
    memset(hd, 0, sizeof hd);
    for (unsigned u0 = 0; u0 < units; u0 += OFS_STEP_UNITS) {
        const unsigned u1 = u0 + OFS_STEP_UNITS < units ? u0 + OFS_STEP_UNITS : units;
        for (int pass = 0; pass < 2; pass++)
            for (unsigned u = u0; u < u1; u++)
                for (unsigned k = 0; k < 4; k++) {
                    const unsigned pos = u * 16u + L->fo[k];
                    if (!L->cls[k] || pos + L->fs[k] > n)
                        continue;
                    const uint64_t key = ((uint64_t)L->cls[k] << 48) | field_value(src + pos, L->fs[k]);
                    const uint64_t blk = pos / L->block;
                    const unsigned slot = table_slot(key & 0xFFFFFFFFFFFFull, L->cls[k]);
                    if (pass == 0) {
                        const uint64_t e = table[slot];
                        if ((e & 0x3FFFFFFFFFFFFull) == key) {
                            const unsigned dist = (unsigned)(blk - (e >> 50));
                            if (!window_bytes || dist * L->block <= window_bytes)
                                hd[4u * u + k] = (uint16_t)dist;
                        }
                    } else if (!eq[0][4u * u + k]) {
                        /* (a field that repeats the one a block earlier is not entered again: the run's first block
                           stays the candidate, and flat areas do not hammer one table entry) */
                        const uint64_t e = (blk << 50) | key;
                        if (e > table[slot])
                            table[slot] = e;
                    }
                }
    }
    /* c. + d. per half-tile */
    uint32_t small32 = 0, cls32 = 0, four32 = 0;
    for (unsigned i = 0; i < 32; i++) {
        small32 |= (uint32_t)(L->fs[i & 3u] == 2u) << i;
        cls32 |= (uint32_t)(L->cls[i & 3u] != 0u) << i;
        four32 |= (uint32_t)(L->fs[i & 3u] == 4u) << i;
    }
    for (unsigned h = 0; h * 32u < fields; h++) {
        const unsigned f0 = h * 32u;
        unsigned nv = 0;                                  /* valid fields of this half-tile (a prefix) */
        for (unsigned i = 0; i < 32u && f0 + i < fields; i++) {
            const unsigned k = i & 3u, pos = ((f0 + i) >> 2) * 16u + L->fo[k];
            if (pos + L->fs[k] <= n)
                nv = i + 1u;
        }
        const uint32_t valid = nv >= 32u ? 0xFFFFFFFFu : ((1u << nv) - 1u);
        uint32_t E[OFS_DISTANCES], A[OFS_DISTANCES], U = 0, H = 0;
        for (unsigned d = 0; d < OFS_DISTANCES; d++) {
            uint32_t e = 0;
            for (unsigned i = 0; i < nv; i++)
                e |= (uint32_t)eq[d][f0 + i] << i;
            e &= ~(small32 & ~(e >> 1) & ~(e << 1));      /* a 2-byte field only next to a neighbour that matches too */
            E[d] = e;
            A[d] = 0;
            U |= e;
        }
        for (unsigned i = 0; i < nv; i++)
            if (hd[f0 + i])
                H |= 1u << i;
        uint32_t front = U & ~(U << 1);
        while (front) {
            uint32_t taken = 0, ends = 0;
            for (unsigned d = 0; d < OFS_DISTANCES; d++) {
                const uint32_t seeds = front & E[d] & ~taken;
                const uint32_t sum = E[d] + seeds;
                A[d] |= (E[d] & ~sum) | seeds;
                ends |= sum & ~E[d];
                taken |= seeds;
            }
            front = ends & U;
        }
        uint32_t cov = 0;
        for (unsigned d = 0; d < OFS_DISTANCES; d++)
            cov |= A[d];
        H &= ~cov & cls32;
        uint32_t lit = valid & ~(cov | H), S = 0;
        {
            /* a copy of ONE 4-byte field with literal fields on both sides is not worth its two elements: as a copy it
               costs 2 bytes and a second literal header, as literal bytes 4 -- one byte more for two elements fewer
               (the decoder's lanes walk ceil(elements / 64) of them each) */
            const uint32_t cpy = cov | H;
            const uint32_t lone4 = cpy & ~(cpy << 1) & ~(cpy >> 1) & four32 & (lit << 1) & (lit >> 1);
            lit |= lone4;
            H &= ~lone4;
            for (unsigned d = 0; d < OFS_DISTANCES; d++)
                A[d] &= ~lone4;
        }
        for (int pass = 0; pass < 2; pass++) {
            S = H | (lit & ~(lit << 1));
            for (unsigned d = 0; d < OFS_DISTANCES; d++)
                S |= A[d] & ~(A[d] << 1);
            S &= valid;
            /* a copy is at most 64 bytes: the one that runs across field 16 (byte 64 of the half-tile) is cut there if
               it is longer -- both pieces then fit */
            if (((valid & ~S & ~lit) >> 16) & 1u) {
                unsigned a = 15, b = 17;
                while (!((S >> a) & 1u))
                    a--;                                   /* (field 0 always starts an element) */
                while (b < nv && !((S >> b) & 1u))
                    b++;
                if ((b < 32u ? fpos(L, b) : 128u) - fpos(L, a) > 64u)
                    S |= 1u << 16;
            }
            if (pass == 0) {
                /* a copy never consists of a 2-byte field alone (no copy element is that short): such fields -- the tail
                   of a run that was cut, a match that nothing continues -- become literals, and the starts are found again */
                const uint32_t next_starts = ((S | ~valid) >> 1) | 0x80000000u;
                const uint32_t lone = small32 & S & next_starts & ~lit & ~H & valid;
                lit |= lone;
                for (unsigned d = 0; d < OFS_DISTANCES; d++)
                    A[d] &= ~lone;
            }
        }
        /* elements in order */
        uint8_t *const start = out + produced;
        uint8_t *o = start;
        for (unsigned p = 0; p < nv;) {
            unsigned q = p + 1u;
            while (q < nv && !((S >> q) & 1u))
                q++;
            const unsigned at = (f0 >> 2) * 16u + fpos(L, p);
            element_at[elements] = (uint16_t)(o - out);
            element_out[elements++] = (uint16_t)at;
            const unsigned len = (q < 32u ? fpos(L, q) : 128u) - fpos(L, p);
            if ((lit >> p) & 1u) {
                if (len <= 60u) {
                    *o++ = (uint8_t)((len - 1u) << 2);
                } else {
                    *o++ = 0xF0u;
                    *o++ = (uint8_t)(len - 1u);
                }
                memcpy(o, src + at, len);
                o += len;
            } else {
                unsigned dist = hd[f0 + p];
                for (int d = (int)OFS_DISTANCES - 1; d >= 0; d--)
                    if ((A[d] >> p) & 1u)
                        dist = (unsigned)d + 1u;
                const unsigned off = dist * L->block;
                if (len < 12u && off < 2048u) {
                    *o++ = (uint8_t)(1u | ((len - 4u) << 2) | ((off >> 8) << 5));
                    *o++ = (uint8_t)off;
                } else {
                    *o++ = (uint8_t)(2u | ((len - 1u) << 2));
                    *o++ = (uint8_t)off;
                    *o++ = (uint8_t)(off >> 8);
                }
            }
            p = q;
        }
        produced += (unsigned)(o - start);
    }
    /* e. the group table the decoder's lanes start from (fragment table version 4): the elements in order, in 64 groups
          of G = ceil(elements / 64) (the last ones shorter or empty); entry g = 24 bits, little endian: the compressed
          bytes of group g | the bytes it produces << 12; then the element count (LE16) and two zero bytes */
    {
        const unsigned G = (elements + 63u) / 64u;
        element_at[elements] = (uint16_t)produced;
        element_out[elements] = (uint16_t)n;
        for (unsigned g = 0; g < 64u; g++) {
            const unsigned a = g * G < elements ? g * G : elements, b = (g + 1u) * G < elements ? (g + 1u) * G : elements;
            const unsigned size = (unsigned)element_at[b] - element_at[a];
            const unsigned made = (unsigned)element_out[b] - element_out[a];
            const unsigned entry = size | (made << 12);
            group_table[3u * g] = (uint8_t)entry;
            group_table[3u * g + 1u] = (uint8_t)(entry >> 8);
            group_table[3u * g + 2u] = (uint8_t)(entry >> 16);
        }
        group_table[192] = (uint8_t)elements;
        group_table[193] = (uint8_t)(elements >> 8);
    }
    return produced;
}

/* whole texture, chunk by chunk and fragment by fragment: total element bytes (for ratio studies) */
unsigned long ofs_texture_bytes(const uint8_t *tex, unsigned long bytes, unsigned chunks, unsigned layout)
{
    static uint8_t out[8192 + 512], hs[OFS_GROUP_TABLE_BYTES];
    unsigned long total = 0;
    const unsigned long cb = bytes / chunks;
    for (unsigned c = 0; c < chunks; c++)
        for (unsigned long o = 0; o < cb; o += 8192u)
            total += ofs_compress_fragment(tex + c * cb + o, (unsigned)(cb - o < 8192u ? cb - o : 8192u), layout, 0, out, hs);
    return total;
}
