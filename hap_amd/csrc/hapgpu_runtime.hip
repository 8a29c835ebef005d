// hapgpu_runtime.hip -- the HIP side of the C ABI in hapgpu_abi.h: device/stream ownership,
// grow-only scratch arenas, host<->device staging, kernel launchers and the HIP-event
// instrumentation bench.py reads.  Everything here is plumbing; the kernels live in
// bc_encode.hip, snappy_compress.hip, frame_pack.hip and snappy_decode.hip.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "hapgpu_abi.h"

extern "C" {
int hapgpu_launch_block_encode(const void *rgba, unsigned width, unsigned height, size_t row_bytes,
                               unsigned format, void *out, hipStream_t stream);
int hapgpu_launch_block_encode_batch(const uint64_t *sources, const uint64_t *outputs, unsigned pictures,
                                     unsigned width, unsigned height, size_t row_bytes, unsigned format,
                                     int wide, hipStream_t stream);
int hapgpu_launch_block_encode_batch_ycocg_alpha(const uint64_t *sources, const uint64_t *colour_outputs,
                                                 const uint64_t *alpha_outputs, unsigned pictures, unsigned width,
                                                 unsigned height, size_t row_bytes, int wide, hipStream_t stream);
int hapgpu_launch_block_decode(const void *blocks, const void *alpha, unsigned width, unsigned height,
                               unsigned format, void *rgba, size_t row_bytes, hipStream_t stream);
int hapgpu_launch_snappy_compress(const HapGpuFrameEnc *frames, unsigned frame_count, unsigned max_frags_per_texture,
                                  unsigned frag_log2, void *slots, unsigned slot_stride, uint32_t *frag_sizes,
                                  uint8_t *group_tables, unsigned granularity_mask, hipStream_t stream);
int hapgpu_launch_frame_pack(HapGpuFrameEnc *frames, unsigned frame_count, unsigned frag_log2, const void *slots,
                             unsigned slot_stride, const uint32_t *frag_sizes, const uint8_t *group_tables,
                             HapGpuCopyEntry *copies, unsigned extra_first, unsigned chunks_per_frame,
                             unsigned max_chunks_per_texture, unsigned textures, void *pack_scratch, hipStream_t stream);
int hapgpu_launch_frame_gather(const HapGpuCopyEntry *copies, unsigned count, hipStream_t stream);
int hapgpu_launch_decode_plan(HapGpuDecodeJob *jobs, unsigned job_count, unsigned max_chunks, hipStream_t stream);
int hapgpu_launch_scan_blocks(HapGpuDecodeUnit *units, const HapGpuDecodeJob *jobs, HapGpuScanChunk *chunks, unsigned chunk_count,
                              HapGpuScanSegment *segs, void *recs, void *joins, unsigned seg_total, uint32_t *fine_work,
                              unsigned fine_first, unsigned fine_pool, hipStream_t stream);
int hapgpu_launch_snappy_decode(const HapGpuDecodeUnit *units, unsigned unit_count, HapGpuDecodeJob *jobs,
                                unsigned frag_log2, unsigned fragment_kinds, int any_stream_or_copy_units,
                                const uint32_t *fine_work, unsigned fine_slots, const void *scan_recs, const void *scan_joins,
                                const HapGpuScanChunk *scan_chunks, unsigned scan_chunk_count, unsigned scan_blocks_hint,
                                uint32_t *resolved, hipStream_t stream);
}

namespace {
constexpr int kSlots = 16;

struct timed_launch {
    int cls;
    hipEvent_t start, stop;
};
struct recorded_graph {
    uint64_t key;
    hipGraphExec_t exec;
};
constexpr size_t kMaxGraphs = 24;
}

struct hapgpu_rt {
    int device;
    hipStream_t stream;
    void *dev[kSlots];
    size_t dev_cap[kSlots];
    void *pin[kSlots];
    size_t pin_cap[kSlots];
    pthread_mutex_t lock;
    int profiling;
    std::vector<timed_launch> pending;
    std::vector<hipEvent_t> free_events;
    hipEvent_t t0, t1;
    // launch sequences recorded as HIP graphs, by the caller's key (geometry of the call) mixed with `generation`,
    // which moves whenever a scratch arena is reallocated (a recorded graph holds the arenas' addresses)
    std::vector<recorded_graph> graphs;
    uint64_t generation;
    int graphs_off;      // unless HAP_AMD_GRAPHS=1
    int recording;
    const void *scan_recs, *scan_joins;      // the block scan's records of the call in progress (for its decode launch)
    unsigned scan_blocks_hint;
    const HapGpuScanChunk *scan_chunks;
    unsigned scan_chunk_count;
    uint32_t *resolved_blocks;               // device counter: 64 KiB blocks a workgroup decoded
};

#define HIP_OK(expr) ((expr) == hipSuccess)

static void complain(const char *what, hipError_t e)
{
    fprintf(stderr, "hap_amd: %s failed: %s\n", what, hipGetErrorString(e));
}

extern "C" int hapgpu_rt_create(int device, hapgpu_rt **out)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        fprintf(stderr, "hap_amd: no HIP device available (%s); the Hap hot path has no CPU fallback\n",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return 4;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess)
            device = 0;
    }
    if (device >= count)
        return 1;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        complain("hipSetDevice", e);
        return 4;
    }
    hapgpu_rt *rt = new hapgpu_rt();
    rt->device = device;
    memset(rt->dev, 0, sizeof(rt->dev));
    memset(rt->dev_cap, 0, sizeof(rt->dev_cap));
    memset(rt->pin, 0, sizeof(rt->pin));
    memset(rt->pin_cap, 0, sizeof(rt->pin_cap));
    rt->profiling = 0;
    rt->generation = 1;
    // Recording the batched encode's launch sequence as a HIP graph is OPT-IN (HAP_AMD_GRAPHS=1): measured on an
    // MI355X it buys nothing (60 8K frames 3.721 against 3.726 ms per step, 8 frames 0.689 / 0.694, one 1080p frame
    // 0.060 against 0.055 -- slower), and a library that records behind its client's back is a bad neighbour: while a
    // stream of the process is being captured, hipDeviceSynchronize() in ANY thread fails and invalidates the
    // recording, and memset nodes of a replayed graph ran late (see hapgpu_rt_zero).
    {
        const char *g = getenv("HAP_AMD_GRAPHS");
        rt->graphs_off = !(g && g[0] != '0') || HAP_AB_ENV("HAP_AMD_NO_GRAPHS") != NULL;
    }
    rt->recording = 0;
    pthread_mutex_init(&rt->lock, NULL);
    if ((e = hipStreamCreateWithFlags(&rt->stream, hipStreamNonBlocking)) != hipSuccess) {
        complain("hipStreamCreate", e);
        delete rt;
        return 4;
    }
    if (hipEventCreate(&rt->t0) != hipSuccess || hipEventCreate(&rt->t1) != hipSuccess) {
        delete rt;
        return 4;
    }
    if (hipMalloc((void **)&rt->resolved_blocks, 64u * sizeof(uint32_t)) != hipSuccess || hipMemset(rt->resolved_blocks, 0, 64u * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        rt->resolved_blocks = nullptr;       // (a statistic: the decoder runs without it)
    }
    *out = rt;
    return 0;
}

extern "C" void hapgpu_rt_destroy(hapgpu_rt *rt)
{
    if (!rt)
        return;
    (void)hipSetDevice(rt->device);
    (void)hipStreamSynchronize(rt->stream);
    for (int i = 0; i < kSlots; i++) {
        if (rt->dev[i]) (void)hipFree(rt->dev[i]);
        if (rt->pin[i]) (void)hipHostFree(rt->pin[i]);
    }
    for (auto &p : rt->pending) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    for (auto &ev : rt->free_events)
        (void)hipEventDestroy(ev);
    for (auto &g : rt->graphs)
        (void)hipGraphExecDestroy(g.exec);
    if (rt->resolved_blocks)
        (void)hipFree(rt->resolved_blocks);
    (void)hipEventDestroy(rt->t0);
    (void)hipEventDestroy(rt->t1);
    (void)hipStreamDestroy(rt->stream);
    pthread_mutex_destroy(&rt->lock);
    delete rt;
}

extern "C" void hapgpu_rt_lock(hapgpu_rt *rt)
{
    pthread_mutex_lock(&rt->lock);
    (void)hipSetDevice(rt->device);
}

extern "C" void hapgpu_rt_unlock(hapgpu_rt *rt) { pthread_mutex_unlock(&rt->lock); }

extern "C" int hapgpu_rt_trylock(hapgpu_rt *rt)
{
    if (pthread_mutex_trylock(&rt->lock) != 0)
        return 1;
    (void)hipSetDevice(rt->device);
    return 0;
}

extern "C" int hapgpu_rt_device(hapgpu_rt *rt) { return rt->device; }

extern "C" int hapgpu_rt_is_device_ptr(hapgpu_rt *rt, const void *p)
{
    (void)rt;
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();      // unregistered host memory: clear the sticky error
        return 0;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeArray;
}

extern "C" void *hapgpu_rt_device_scratch(hapgpu_rt *rt, int slot, size_t bytes)
{
    if (slot < 0 || slot >= kSlots)
        return NULL;
    if (bytes == 0)
        bytes = 256;
    if (rt->dev_cap[slot] < bytes) {
        // contents are not preserved; wait for work that may still use the old block
        (void)hipStreamSynchronize(rt->stream);
        rt->generation++;
        if (rt->dev[slot])
            (void)hipFree(rt->dev[slot]);
        rt->dev[slot] = NULL;
        rt->dev_cap[slot] = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipMalloc(&rt->dev[slot], want);
        if (e != hipSuccess) {
            complain("hipMalloc(scratch)", e);
            return NULL;
        }
        rt->dev_cap[slot] = want;
    }
    return rt->dev[slot];
}

extern "C" void *hapgpu_rt_pinned_scratch(hapgpu_rt *rt, int slot, size_t bytes)
{
    if (slot < 0 || slot >= kSlots)
        return NULL;
    if (bytes == 0)
        bytes = 256;
    if (rt->pin_cap[slot] < bytes) {
        (void)hipStreamSynchronize(rt->stream);
        rt->generation++;
        if (rt->pin[slot])
            (void)hipHostFree(rt->pin[slot]);
        rt->pin[slot] = NULL;
        rt->pin_cap[slot] = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&rt->pin[slot], want, hipHostMallocDefault);
        if (e != hipSuccess) {
            complain("hipHostMalloc(scratch)", e);
            return NULL;
        }
        rt->pin_cap[slot] = want;
    }
    return rt->pin[slot];
}

// Small transfers between the pinned scratch and device memory as a kernel on the stream's own queue: descriptors going
// up and result words coming back are a few KiB each, and as copies they travel through the DMA engines -- every one of
// them a hand-over between engines in front of and behind the kernels (HAP_AMD_COPY_KERNELS=0 keeps hipMemcpyAsync).
__global__ __launch_bounds__(256) void small_copy_kernel(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, size_t bytes)
{
    const size_t i = ((size_t)blockIdx.x * 256u + threadIdx.x) * 16u;
    if (i + 16u <= bytes && (((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
        *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(src + i);
    } else if (i < bytes) {
        const size_t n = bytes - i < 16u ? bytes - i : 16u;
        for (size_t k = 0; k < n; k++)
            dst[i + k] = src[i + k];
    }
}

static bool in_pinned_scratch(const hapgpu_rt *rt, const void *p, size_t bytes)
{
    for (int i = 0; i < kSlots; i++)
        if (rt->pin[i] && (const uint8_t *)p >= (const uint8_t *)rt->pin[i] &&
            (const uint8_t *)p + bytes <= (const uint8_t *)rt->pin[i] + rt->pin_cap[i])
            return true;
    return false;
}

static const size_t kSmallCopyBytes = (size_t)1 << 20;
static bool copy_kernels_enabled()
{
    static int on = -1;
    if (on < 0) {
        const char *v = HAP_AB_ENV("HAP_AMD_COPY_KERNELS");
        on = (v && v[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

// 1: kernels may read and write the pinned scratch in place (it is mapped into the device's address space)
extern "C" int hapgpu_rt_pinned_is_mapped(hapgpu_rt *rt)
{
    (void)rt;
    return copy_kernels_enabled() ? 1 : 0;
}

static int small_copy(hapgpu_rt *rt, void *dst, const void *src, size_t bytes)
{
    hipLaunchKernelGGL(small_copy_kernel, dim3((unsigned)((bytes + 4095u) / 4096u)), dim3(256), 0, rt->stream, (uint8_t *)dst,
                       (const uint8_t *)src, bytes);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

extern "C" int hapgpu_rt_h2d(hapgpu_rt *rt, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return 0;
    if (bytes <= kSmallCopyBytes && copy_kernels_enabled() && in_pinned_scratch(rt, src, bytes))
        return small_copy(rt, dst, src, bytes);
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, rt->stream);
    if (e != hipSuccess) { complain("hipMemcpyAsync(H2D)", e); return 4; }
    return 0;
}

extern "C" int hapgpu_rt_d2h(hapgpu_rt *rt, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return 0;
    if (bytes <= kSmallCopyBytes && copy_kernels_enabled() && in_pinned_scratch(rt, dst, bytes))
        return small_copy(rt, dst, src, bytes);
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, rt->stream);
    if (e != hipSuccess) { complain("hipMemcpyAsync(D2H)", e); return 4; }
    return 0;
}

// rows of `row` bytes from a device picture (pitch spitch) to a host one (pitch dpitch): the bytes between the rows of the
// destination stay as they are
extern "C" int hapgpu_rt_d2h_rows(hapgpu_rt *rt, void *dst, size_t dpitch, const void *src, size_t spitch, size_t row, size_t rows)
{
    if (!row || !rows) return 0;
    hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, row, rows, hipMemcpyDeviceToHost, rt->stream);
    if (e != hipSuccess) { complain("hipMemcpy2DAsync(D2H)", e); return 4; }
    return 0;
}

extern "C" int hapgpu_rt_d2d(hapgpu_rt *rt, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return 0;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rt->stream);
    if (e != hipSuccess) { complain("hipMemcpyAsync(D2D)", e); return 4; }
    return 0;
}

__global__ __launch_bounds__(256) void small_zero_kernel(uint8_t *__restrict__ dst, size_t bytes)
{
    const size_t i = ((size_t)blockIdx.x * 256u + threadIdx.x) * 16u;
    if (i + 16u <= bytes && ((uintptr_t)dst & 15u) == 0) {
        *reinterpret_cast<uint4 *>(dst + i) = make_uint4(0, 0, 0, 0);
    } else if (i < bytes) {
        const size_t n = bytes - i < 16u ? bytes - i : 16u;
        for (size_t k = 0; k < n; k++)
            dst[i + k] = 0;
    }
}

extern "C" int hapgpu_rt_zero(hapgpu_rt *rt, void *dst, size_t bytes)
{
    if (!bytes) return 0;
    // (a kernel of this library's own, whatever the size: hipMemsetAsync recorded into a HIP graph did not do its work in
    // order when the graph was launched a second time -- small buffers, ROCm 7.2: the compressor's published sizes were
    // wiped under its waiting wavefronts, every frame of the call was encoded twice.  HAP_AMD_MEMSET_NODES=1: the old way)
    if (bytes <= ((size_t)1 << 40) && !HAP_AB_ENV("HAP_AMD_MEMSET_NODES")) {
        for (size_t done = 0; done < bytes; done += (size_t)1 << 30) {      // (2^18 workgroups of 4 KiB a launch)
            const size_t part = bytes - done < ((size_t)1 << 30) ? bytes - done : (size_t)1 << 30;
            hipLaunchKernelGGL(small_zero_kernel, dim3((unsigned)((part + 4095u) / 4096u)), dim3(256), 0, rt->stream, (uint8_t *)dst + done, part);
        }
        return hipGetLastError() == hipSuccess ? 0 : 4;
    }
    hipError_t e = hipMemsetAsync(dst, 0, bytes, rt->stream);
    if (e != hipSuccess) { complain("hipMemsetAsync", e); return 4; }
    return 0;
}

extern "C" int hapgpu_rt_sync(hapgpu_rt *rt)
{
    hipError_t e = hipStreamSynchronize(rt->stream);
    if (e != hipSuccess) { complain("hipStreamSynchronize", e); return 4; }
    return 0;
}

// ---- recorded launch sequences ---------------------------------------------------------------
// A batched call issues the same copies and kernels with the same arguments whenever its geometry is the same: what
// changes from call to call (buffer addresses, sizes, per-frame results) travels through descriptor tables in pinned
// memory, not through kernel arguments.  Such a sequence is recorded once as a HIP graph (stream capture) and
// replayed with one launch: the fixed host cost of a call is one graph launch instead of 5-9 kernel launches and
// 2-3 copies.  Not used while per-kernel timing is on (bench.py's HIP events bracket individual launches).
static uint64_t mix_key(uint64_t key, uint64_t generation)
{
    uint64_t h = key ^ (generation * 0x9E3779B97F4A7C15ull);
    h ^= h >> 31;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
    return h ? h : 1;
}

// 1: a recorded sequence was launched, the caller skips its launches; 0: recording started, the caller issues its
// launches and then calls hapgpu_rt_graph_end; 2: no graph for this call, launch as usual (and do not call _end)
extern "C" int hapgpu_rt_graph_begin(hapgpu_rt *rt, uint64_t key)
{
    if (rt->graphs_off || rt->profiling || rt->recording)
        return 2;
    const uint64_t k = mix_key(key, rt->generation);
    for (auto &g : rt->graphs)
        if (g.key == k)
            return hipGraphLaunch(g.exec, rt->stream) == hipSuccess ? 1 : 2;
    // (relaxed: this thread makes no call between here and the end of the recording that a capture could object to, and
    // in thread-local mode ROCm 7.2 still refused ANOTHER thread's hipStreamSynchronize of another context's stream
    // while this one was recording -- two contexts, two threads: "operation not permitted when stream is capturing")
    if (hipStreamBeginCapture(rt->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        return 2;
    }
    rt->recording = 1;
    return 0;
}

// ends the recording; failed != 0: something in the sequence did not launch (the recording is dropped and nothing
// runs).  0: the sequence was instantiated, remembered and launched
extern "C" int hapgpu_rt_graph_end(hapgpu_rt *rt, uint64_t key, int failed)
{
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (!rt->recording)
        return 4;
    rt->recording = 0;
    if (hipStreamEndCapture(rt->stream, &graph) != hipSuccess || !graph) {
        (void)hipGetLastError();
        return 4;
    }
    if (failed) {
        (void)hipGraphDestroy(graph);
        return 4;
    }
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess || !exec) {
        complain("hipGraphInstantiate", e);
        return 4;
    }
    if (rt->graphs.size() >= kMaxGraphs) {
        (void)hipGraphExecDestroy(rt->graphs.front().exec);
        rt->graphs.erase(rt->graphs.begin());
    }
    rt->graphs.push_back({mix_key(key, rt->generation), exec});
    if (hipGraphLaunch(exec, rt->stream) != hipSuccess) {
        complain("hipGraphLaunch", hipGetLastError());
        return 4;
    }
    return 0;
}

extern "C" void hapgpu_rt_graphs_disable(hapgpu_rt *rt)
{
    if (!rt->graphs_off)
        fprintf(stderr, "hap_amd: HIP graph of a launch sequence failed; this context launches plainly from now on\n");
    rt->graphs_off = 1;
}

// ---- instrumentation -----------------------------------------------------------------------

static hipEvent_t take_event(hapgpu_rt *rt)
{
    if (!rt->free_events.empty()) {
        hipEvent_t ev = rt->free_events.back();
        rt->free_events.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}

struct scoped_timing {
    hapgpu_rt *rt;
    timed_launch t;
    bool on;
    scoped_timing(hapgpu_rt *r, int cls) : rt(r), on(r->profiling != 0)
    {
        if (on) {
            t.cls = cls;
            t.start = take_event(rt);
            t.stop = take_event(rt);
            (void)hipEventRecord(t.start, rt->stream);
        }
    }
    ~scoped_timing()
    {
        if (on) {
            (void)hipEventRecord(t.stop, rt->stream);
            rt->pending.push_back(t);
        }
    }
};

extern "C" void hapgpu_rt_set_profiling(hapgpu_rt *rt, int enable) { rt->profiling = enable; }

extern "C" int hapgpu_rt_collect_profile(hapgpu_rt *rt, unsigned long *launches, double *ms, unsigned classes)
{
    if (hipStreamSynchronize(rt->stream) != hipSuccess)
        return 4;
    for (auto &p : rt->pending) {
        float dt = 0.f;
        if (hipEventElapsedTime(&dt, p.start, p.stop) == hipSuccess && (unsigned)p.cls < classes) {
            launches[p.cls] += 1;
            ms[p.cls] += (double)dt;
        }
        rt->free_events.push_back(p.start);
        rt->free_events.push_back(p.stop);
    }
    rt->pending.clear();
    return 0;
}

extern "C" int hapgpu_rt_timer_start(hapgpu_rt *rt) { return HIP_OK(hipEventRecord(rt->t0, rt->stream)) ? 0 : 4; }

extern "C" int hapgpu_rt_timer_stop(hapgpu_rt *rt, double *ms)
{
    float dt = 0.f;
    if (!HIP_OK(hipEventRecord(rt->t1, rt->stream)) || !HIP_OK(hipEventSynchronize(rt->t1)) ||
        !HIP_OK(hipEventElapsedTime(&dt, rt->t0, rt->t1)))
        return 4;
    *ms = (double)dt;
    return 0;
}

// ---- header prefixes of device-resident frames ------------------------------------------------
// The host parses section headers and tables (hap_frame.c); for frames that live in HBM their first
// `prefix` bytes are gathered into one contiguous block so that a single copy brings them back.
__device__ void copy_prefix(const uint8_t *src, uint64_t n, uint8_t *dst)
{
    if ((((uintptr_t)src) & 15u) == 0) {
        const unsigned wide = (unsigned)(n >> 4);
        for (unsigned i = threadIdx.x; i < wide; i += 256u)
            reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
        for (unsigned i = (wide << 4) + threadIdx.x; i < n; i += 256u)
            dst[i] = src[i];
    } else {
        for (unsigned i = threadIdx.x; i < n; i += 256u)
            dst[i] = src[i];
    }
}

// section header at p (3-byte length + type, or 0 + type + 4-byte length: the container of hap.c:160-181): header
// bytes, or 0 when it does not fit `available`
__device__ unsigned section_header(const uint8_t *p, uint64_t available, uint64_t *length, unsigned *type)
{
    if (available < 4u)
        return 0u;
    uint64_t len = (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16);
    unsigned header = 4u;
    if (len == 0u) {
        if (available < 8u)
            return 0u;
        len = (uint64_t)p[4] | ((uint64_t)p[5] << 8) | ((uint64_t)p[6] << 16) | ((uint64_t)p[7] << 24);
        header = 8u;
    }
    *length = len;
    *type = p[3];
    return header + len > available ? 0u : header;
}

// `far` (optional): entries with far[f] != 0 also want the bytes where the SECOND texture's section of a
// multi-texture frame begins, when that lies beyond the first prefix.  The two section headers in front of it are
// read here, so that both prefixes come back in one copy (the host parses the same headers again and takes the second
// prefix only when it finds the same offset).
__global__ __launch_bounds__(256) void gather_prefix_kernel(const uint64_t *__restrict__ frames,
                                                            const uint64_t *__restrict__ lengths, unsigned prefix,
                                                            uint8_t *__restrict__ out, const uint8_t *__restrict__ far,
                                                            uint8_t *__restrict__ out2, uint64_t *__restrict__ far_at)
{
    const uint8_t *src = (const uint8_t *)frames[blockIdx.x];
    if (!src)
        return;
    const uint64_t total = lengths[blockIdx.x];
    copy_prefix(src, total < prefix ? total : prefix, out + (size_t)blockIdx.x * prefix);
    if (!far || !far[blockIdx.x])
        return;
    uint64_t at = 0, top_len = 0, first_len = 0;
    unsigned top_type = 0, first_type = 0;
    const unsigned top_header = section_header(src, total, &top_len, &top_type);
    if (top_header && top_type == 0x0Du) {
        const unsigned first_header = section_header(src + top_header, top_len, &first_len, &first_type);
        if (first_header) {
            at = (uint64_t)top_header + first_header + first_len;
            if (at + 16u <= prefix || at >= total)
                at = 0;                                              // inside the first prefix, or nothing there
        }
    }
    if (threadIdx.x == 0)
        far_at[blockIdx.x] = at;
    if (at)
        copy_prefix(src + at, total - at < prefix ? total - at : prefix, out2 + (size_t)blockIdx.x * prefix);
}

extern "C" int hapgpu_k_gather_prefixes(hapgpu_rt *rt, const uint64_t *frames_dev, const uint64_t *lengths_dev,
                                        unsigned count, unsigned prefix, void *out_dev)
{
    if (count == 0)
        return 0;
    hipLaunchKernelGGL(gather_prefix_kernel, dim3(count), dim3(256), 0, rt->stream, frames_dev, lengths_dev, prefix,
                       (uint8_t *)out_dev, (const uint8_t *)nullptr, (uint8_t *)nullptr, (uint64_t *)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

extern "C" int hapgpu_k_gather_prefixes_far(hapgpu_rt *rt, const uint64_t *frames_dev, const uint64_t *lengths_dev,
                                            unsigned count, unsigned prefix, void *out_dev, const uint8_t *far_dev,
                                            void *out2_dev, uint64_t *far_at_dev)
{
    if (count == 0)
        return 0;
    hipLaunchKernelGGL(gather_prefix_kernel, dim3(count), dim3(256), 0, rt->stream, frames_dev, lengths_dev, prefix,
                       (uint8_t *)out_dev, far_dev, (uint8_t *)out2_dev, far_at_dev);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// ---- launchers -----------------------------------------------------------------------------

extern "C" int hapgpu_k_block_encode(hapgpu_rt *rt, const void *rgba, unsigned width, unsigned height,
                                     size_t row_bytes, unsigned hap_texture_format, void *out)
{
    scoped_timing st(rt, 0);
    return hapgpu_launch_block_encode(rgba, width, height, row_bytes, hap_texture_format, out, rt->stream);
}

extern "C" int hapgpu_k_block_encode_batch(hapgpu_rt *rt, const uint64_t *sources, const uint64_t *outputs,
                                           unsigned pictures, unsigned width, unsigned height, size_t row_bytes,
                                           unsigned hap_texture_format, int wide)
{
    scoped_timing st(rt, 0);
    return hapgpu_launch_block_encode_batch(sources, outputs, pictures, width, height, row_bytes, hap_texture_format,
                                            wide, rt->stream);
}

extern "C" int hapgpu_k_block_encode_batch_ycocg_alpha(hapgpu_rt *rt, const uint64_t *sources, const uint64_t *colour_outputs,
                                                       const uint64_t *alpha_outputs, unsigned pictures, unsigned width,
                                                       unsigned height, size_t row_bytes, int wide)
{
    scoped_timing st(rt, 0);
    return hapgpu_launch_block_encode_batch_ycocg_alpha(sources, colour_outputs, alpha_outputs, pictures, width, height,
                                                        row_bytes, wide, rt->stream);
}

extern "C" int hapgpu_launch_block_decode_batch(const uint64_t *table, unsigned pictures, int with_alpha, unsigned width,
                                                unsigned height, unsigned format, size_t row_bytes, hipStream_t stream);
extern "C" int hapgpu_k_block_decode_batch(hapgpu_rt *rt, const uint64_t *table, unsigned pictures, int with_alpha,
                                           unsigned width, unsigned height, unsigned hap_texture_format, size_t row_bytes)
{
    scoped_timing st(rt, 6);
    return hapgpu_launch_block_decode_batch(table, pictures, with_alpha, width, height, hap_texture_format, row_bytes, rt->stream);
}

extern "C" int hapgpu_k_block_decode(hapgpu_rt *rt, const void *blocks, const void *alpha, unsigned width,
                                     unsigned height, unsigned hap_texture_format, void *rgba, size_t row_bytes)
{
    scoped_timing st(rt, 6);
    return hapgpu_launch_block_decode(blocks, alpha, width, height, hap_texture_format, rgba, row_bytes, rt->stream);
}

extern "C" int hapgpu_k_snappy_compress(hapgpu_rt *rt, const HapGpuFrameEnc *frames, unsigned frame_count,
                                        unsigned max_frags_per_texture, unsigned frag_log2, void *slots,
                                        unsigned slot_stride, uint32_t *frag_sizes, uint8_t *group_tables,
                                        unsigned granularity_mask)
{
    scoped_timing st(rt, ((granularity_mask >> 16) & 0xFu) ? 8 : 1);      // (8: the block encoder runs inside, HapGpuKernel_EncodeFused)
    return hapgpu_launch_snappy_compress(frames, frame_count, max_frags_per_texture, frag_log2, slots, slot_stride,
                                         frag_sizes, group_tables, granularity_mask, rt->stream);
}

extern "C" int hapgpu_k_frame_pack(hapgpu_rt *rt, HapGpuFrameEnc *frames, unsigned frame_count, unsigned frag_log2,
                                   const void *slots, unsigned slot_stride, const uint32_t *frag_sizes,
                                   const uint8_t *group_tables, HapGpuCopyEntry *copies, unsigned extra_first,
                                   unsigned chunks_per_frame, unsigned max_chunks_per_texture, unsigned textures,
                                   void *pack_scratch)
{
    scoped_timing st(rt, 2);
    return hapgpu_launch_frame_pack(frames, frame_count, frag_log2, slots, slot_stride, frag_sizes, group_tables, copies,
                                    extra_first, chunks_per_frame, max_chunks_per_texture, textures, pack_scratch, rt->stream);
}

extern "C" int hapgpu_k_frame_gather(hapgpu_rt *rt, const HapGpuCopyEntry *copies, unsigned count)
{
    scoped_timing st(rt, 3);
    return hapgpu_launch_frame_gather(copies, count, rt->stream);
}

extern "C" int hapgpu_k_decode_plan(hapgpu_rt *rt, HapGpuDecodeJob *jobs, unsigned job_count,
                                    HapGpuDecodeUnit *units, unsigned unit_count, unsigned max_chunks)
{
    // unit slots the planner does not reach (it stops at the first malformed chunk) must read as SKIP
    if (unit_count && hipMemsetAsync(units, 0, (size_t)unit_count * sizeof(HapGpuDecodeUnit), rt->stream) != hipSuccess)
        return 4;
    scoped_timing st(rt, 4);
    return hapgpu_launch_decode_plan(jobs, job_count, max_chunks, rt->stream);
}

extern "C" int hapgpu_k_scan_blocks(hapgpu_rt *rt, HapGpuDecodeUnit *units, const HapGpuDecodeJob *jobs, HapGpuScanChunk *chunks,
                                    unsigned chunk_count, HapGpuScanSegment *segs, void *recs, void *joins, unsigned seg_total,
                                    uint32_t *fine_work, unsigned fine_first, unsigned fine_pool)
{
    scoped_timing st(rt, 7);
    // (the decode launch of the same call reads the scan's records again: the 64 KiB blocks as workgroups)
    rt->scan_recs = recs;
    rt->scan_joins = joins;
    rt->scan_chunks = chunks;
    rt->scan_chunk_count = chunk_count;
    // (about how many 64 KiB blocks the scanned streams hold: an eighth of the 8 KiB pieces the host made room for -- what
    // the frames' textures hold -- or, without those, what the compressed bytes would be at three to one)
    rt->scan_blocks_hint = fine_pool ? fine_pool / 8u + 1u : seg_total / 5u + 1u;
    return hapgpu_launch_scan_blocks(units, jobs, chunks, chunk_count, segs, recs, joins, seg_total, fine_work, fine_first, fine_pool,
                                     rt->stream);
}

extern "C" int hapgpu_launch_guess_group_tables(HapGpuDecodeUnit *units, unsigned unit_count, const HapGpuDecodeJob *jobs,
                                                const uint32_t *work, unsigned work_slots, hipStream_t stream);
#ifdef BRK_TIMING
extern "C" void hapgpu_debug_merge_counters(unsigned *out);
#endif
extern "C" int hapgpu_launch_group_tables_from_records(HapGpuDecodeUnit *units, unsigned unit_count, const HapGpuDecodeJob *jobs,
                                                       const uint32_t *work, unsigned work_slots, const void *recs, const void *joins,
                                                       hipStream_t stream);
extern "C" int hapgpu_k_guess_group_tables(hapgpu_rt *rt, HapGpuDecodeUnit *units, unsigned unit_count, const HapGpuDecodeJob *jobs,
                                           const uint32_t *work, unsigned work_slots)
{
    scoped_timing st(rt, 7);          // (with the block scan: finding where wavefronts may start in streams that do not say)
    // the pieces the scan of this call listed: from its records, a wavefront per piece (r06); fragments that are chunks of
    // their own (no scan): a lane per fragment walks it
    if (work && rt->scan_recs && rt->scan_joins)
        return hapgpu_launch_group_tables_from_records(units, unit_count, jobs, work, work_slots, rt->scan_recs, rt->scan_joins, rt->stream);
    return hapgpu_launch_guess_group_tables(units, unit_count, jobs, work, work_slots, rt->stream);
}

extern "C" int hapgpu_k_snappy_decode(hapgpu_rt *rt, const HapGpuDecodeUnit *units, unsigned unit_count,
                                      HapGpuDecodeJob *jobs, unsigned frag_log2, unsigned fragment_kinds,
                                      int any_stream_or_copy_units, const uint32_t *fine_work, unsigned fine_slots)
{
    scoped_timing st(rt, 5);
    const void *recs = rt->scan_recs, *joins = rt->scan_joins;
    rt->scan_recs = nullptr;                 // (one call's records: never another's)
    rt->scan_joins = nullptr;
    return hapgpu_launch_snappy_decode(units, unit_count, jobs, frag_log2, fragment_kinds, any_stream_or_copy_units,
                                       fine_work, fine_slots, any_stream_or_copy_units == 2 ? recs : nullptr,
                                       any_stream_or_copy_units == 2 ? joins : nullptr, rt->scan_chunks, rt->scan_chunk_count,
                                       rt->scan_blocks_hint, rt->resolved_blocks, rt->stream);
}

// 64 KiB blocks of other encoders' streams that a workgroup decoded (snappy_decode_block_resolve_kernel) since the runtime
// was made; waits for the stream
extern "C" unsigned hapgpu_rt_resolved_blocks(hapgpu_rt *rt)
{
    uint32_t v = 0;
    if (!rt->resolved_blocks)
        return 0;
    if (hipStreamSynchronize(rt->stream) != hipSuccess || hipMemcpy(&v, rt->resolved_blocks, sizeof v, hipMemcpyDeviceToHost) != hipSuccess)
        return 0;
#ifdef BRK_TIMING
    {
        uint32_t d[24];
        if (hipMemcpy(d, rt->resolved_blocks, sizeof d, hipMemcpyDeviceToHost) == hipSuccess && getenv("BRK_PRINT")) {
            {
                unsigned m[8] = {0};
                hapgpu_debug_merge_counters(m);
                fprintf(stderr, "merge: streams %u segments %u good %u windows parsed by the merge %u record joins %u ticks %u\n", m[0], m[1], m[2], m[3], m[4], m[5]);
            }
            fprintf(stderr, "brk: declined: too long %u, window %u, gap walk %u, links %u, ends %u, first %u\n", d[16], d[17], d[18], d[19], d[20], d[21]);
            fprintf(stderr, "brk: blocks %u  ticks records %u windows %u verify %u jump %u fetch %u  rounds %u windows %u\n", d[0], d[2], d[3], d[4], d[5], d[6], d[8], d[9]);
        }
    }
#endif
    return v;
}
