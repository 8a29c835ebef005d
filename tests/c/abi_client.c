/* A plain C99 client of the library: proves that the headers under include/ are valid C (no C++ or HIP types), that the
 * library links like the reference's object file would, and exercises the host-only entry points.
 * Built and run by tests/test_c_client.py; prints "ok" and exits 0 on success. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hap.h"
#include "hap_gpu.h"
#include "hap_sequence.h"

static void serial(HapDecodeWorkFunction function, void *p, unsigned int count, void *info)
{
    unsigned int i;
    (void)info;
    for (i = 0; i < count; i++)
        function(p, i);
}

/* a client that runs the work items last to first, counts them, and -- as a threaded player might -- decodes another
 * frame through hap.h from inside its callback (the reference is re-entrant: hap.c has no globals) */
struct nested {
    const unsigned char *frame;
    unsigned long frame_bytes;
    unsigned int calls, items, nested_result;
    unsigned char out[64];
};

static void backwards_and_reentrant(HapDecodeWorkFunction function, void *p, unsigned int count, void *info)
{
    struct nested *n = (struct nested *)info;
    unsigned int i, format = 0;
    unsigned long used = 0;
    n->calls += 1;
    n->nested_result = HapDecode(n->frame, n->frame_bytes, 0, serial, NULL, n->out, sizeof(n->out), &used, &format);
    for (i = count; i > 0; i--) {
        function(p, i - 1);
        n->items += 1;
    }
}

/* HapEncode -> HapDecode of an 8-chunk Snappy frame through hap.h alone (GPU box only) */
static int chunked_round_trip(const unsigned char *small_frame, unsigned long small_bytes)
{
    enum { BYTES = 16 * 2048 };
    static unsigned char texture[BYTES], decoded[BYTES];
    unsigned long lengths[1] = {BYTES}, used = 0, frame_bytes = 0, cap;
    unsigned int formats[1] = {HapTextureFormat_RGBA_DXT5}, chunks[1] = {8}, comps[1] = {HapCompressorSnappy}, format = 0;
    const void *inputs[1] = {texture};
    unsigned char *frame;
    struct nested info;
    int chunk_count = 0;
    unsigned int i, rc;
    for (i = 0; i < BYTES; i++)            /* blocks that repeat with small changes: compressible, not trivial */
        texture[i] = (unsigned char)((i & 15u) < 8u ? (i >> 6) : (i * 2654435761u >> 13));
    cap = HapMaxEncodedLength(1, lengths, formats, chunks);
    frame = (unsigned char *)malloc(cap);
    if (!frame || cap == 0)
        return 20;
    if (HapEncode(1, inputs, lengths, formats, comps, chunks, frame, cap, &frame_bytes) != HapResult_No_Error)
        return 21;
    if (frame[3] != 0xCE || frame_bytes >= BYTES)       /* chunked (complex) DXT5 section that did shrink */
        return 22;
    if (HapGetFrameTextureChunkCount(frame, frame_bytes, 0, &chunk_count) != HapResult_No_Error || chunk_count != 8)
        return 23;
    memset(&info, 0, sizeof(info));
    info.frame = small_frame;
    info.frame_bytes = small_bytes;
    memset(decoded, 0xEE, sizeof(decoded));
    rc = HapDecode(frame, frame_bytes, 0, backwards_and_reentrant, &info, decoded, sizeof(decoded), &used, &format);
    if (rc != HapResult_No_Error || used != BYTES || format != HapTextureFormat_RGBA_DXT5)
        return 24;
    if (info.calls != 1 || info.items != 8)             /* hap.h:113-130: invoked once, count = chunk count */
        return 25;
    if (info.nested_result != HapResult_No_Error || memcmp(info.out, small_frame + 4, 64) != 0)
        return 26;
    if (memcmp(decoded, texture, BYTES) != 0)
        return 27;
    /* a too small destination is refused before any chunk runs (hap.c:840-843) */
    if (HapDecode(frame, frame_bytes, 0, serial, NULL, decoded, BYTES - 1, &used, &format) != HapResult_Buffer_Too_Small)
        return 28;
    free(frame);
    return 0;
}

int main(int argc, char **argv)
{
    /* 64 bytes DXT1 stored as-is by hand: [len 64][type 0xAB][payload] */
    unsigned char frame[68], out[64];
    unsigned long lengths[1] = {64}, used = 0;
    unsigned int formats[1] = {HapTextureFormat_RGB_DXT1}, chunks[1] = {2}, count = 0, format = 0, n = 0;
    int chunk_count = -1;
    const char *path = argc > 1 ? argv[1] : "/tmp/abi_client.hapseq";
    HapSequenceWriter *w = NULL;
    HapSequenceReader *r = NULL;
    unsigned int i;

    if (HapMaxEncodedLength(1, lengths, formats, chunks) != 176ul)
        return 1;
    memset(frame, 0, sizeof(frame));
    frame[0] = 64; frame[3] = 0xAB;
    for (i = 0; i < 64; i++)
        frame[4 + i] = (unsigned char)(i * 7u);
    if (HapGetFrameTextureCount(frame, sizeof(frame), &count) != HapResult_No_Error || count != 1)
        return 2;
    if (HapGetFrameTextureFormat(frame, sizeof(frame), 0, &format) != HapResult_No_Error ||
        format != HapTextureFormat_RGB_DXT1)
        return 3;
    if (HapGetFrameTextureChunkCount(frame, sizeof(frame), 0, &chunk_count) != HapResult_No_Error || chunk_count != 1)
        return 4;
    {
        unsigned long offsets[2] = {99, 99};
        if (HapGpuGetFrameTextureChunkLayout(frame, sizeof(frame), 0, 2, offsets, &n) != HapResult_No_Error ||
            n != 1 || offsets[0] != 0 || offsets[1] != 64)
            return 5;
    }
    /* decode needs a GPU: without one the call must fail loudly, never fall back */
    {
        unsigned int rc = HapDecode(frame, sizeof(frame), 0, serial, NULL, out, sizeof(out), &used, &format);
        if (HapGpuDefaultContext() == NULL) {
            if (rc != HapResult_Internal_Error)
                return 6;
        } else if (rc != HapResult_No_Error || used != 64 || memcmp(out, frame + 4, 64) != 0) {
            return 7;
        } else {
            int rt = chunked_round_trip(frame, sizeof(frame));
            if (rt != 0)
                return rt;
        }
    }
    /* sequence file round trip (host only) */
    if (HapSequenceWriterOpen(path, 16, 16, 60, 1, &w) != HapResult_No_Error)
        return 8;
    for (i = 0; i < 3; i++)
        if (HapSequenceWriterAppend(w, frame, sizeof(frame)) != HapResult_No_Error)
            return 9;
    if (HapSequenceWriterClose(w) != HapResult_No_Error || HapSequenceReaderOpen(path, &r) != HapResult_No_Error)
        return 10;
    if (HapSequenceReaderInfo(r, NULL, NULL, NULL, NULL, &n) != HapResult_No_Error || n != 3 ||
        HapSequenceReaderFrameBytes(r, 2) != sizeof(frame))
        return 11;
    {
        unsigned char two[2 * sizeof(frame)];
        unsigned long offs[3];
        if (HapSequenceReaderRead(r, 1, 2, two, sizeof(two), offs) != HapResult_No_Error || offs[1] != sizeof(frame) ||
            memcmp(two + offs[1], frame, sizeof(frame)) != 0)
            return 12;
    }
    HapSequenceReaderClose(r);
    /* joining one group gives a frame with the same content */
    {
        const void *groups[1] = {frame};
        unsigned long sizes[1] = {sizeof(frame)}, joined_bytes = 0;
        unsigned char joined[128];
        if (HapGpuJoinChunkGroups(1, groups, sizes, joined, sizeof(joined), &joined_bytes) != HapResult_No_Error ||
            joined_bytes != sizeof(frame) || memcmp(joined, frame, sizeof(frame)) != 0)
            return 13;
    }
    remove(path);
    puts("ok");
    return 0;
}
