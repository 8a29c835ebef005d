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
