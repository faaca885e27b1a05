// mcull_group_*: several shards of one scene, one context per shard (normally one per GPU), driven by
// ONE host thread of ONE process - the form a C++ host like the reference (single-threaded, Engine.cpp:106-234)
// can adopt without a process launcher.  Every shard culls on a stream of its own; the per-instance commands and
// the visible counts are then exchanged by direct device-to-device copies (xGMI is point to point: every shard
// sends its block to every other device over that pair's own link, no ring), and each device rebases and packs
// the blocks it received into one command buffer for the whole scene.  The reference has no counterpart
// (it picks one physical device, Engine.cpp:658-666).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mcull_internal.h"

namespace {

constexpr uint32_t kMaxShards = MCULL_GROUP_MAX_SHARDS;

// What every device needs to turn the received blocks into the scene-wide buffers.
struct GroupLayout {
    uint32_t n_shards;
    uint32_t stride, offset, form;        // command record layout
    uint32_t first_instance[kMaxShards + 1];
    uint64_t block_offset[kMaxShards];    // byte offset of shard r's block in the staging buffer
    uint64_t first_task_offset[kMaxShards];  // inside a block: the shard's first_task array (EXT records only)
    uint64_t total_offset[kMaxShards];    // inside a block: the shard's visible count
};

// One thread per instance of the whole scene: copies its record out of the owning shard's block and adds the
// visible counts of the lower shards to firstTask (modulo 2^32, like the 32-bit field it is).
__global__ void group_pack_kernel(const uint8_t *__restrict__ staged, GroupLayout lay, uint8_t *__restrict__ commands,
                                  uint32_t *__restrict__ first_task, uint64_t *__restrict__ totals) {
    const uint32_t n_total = lay.first_instance[lay.n_shards];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < lay.n_shards) totals[i] = *(const uint64_t *)(staged + lay.block_offset[i] + lay.total_offset[i]);
    if (i == 0) {
        uint64_t sum = 0;
        for (uint32_t r = 0; r < lay.n_shards; r++) sum += *(const uint64_t *)(staged + lay.block_offset[r] + lay.total_offset[r]);
        totals[lay.n_shards] = sum;
    }
    if (i >= n_total) return;
    uint32_t shard = 0;
    uint64_t base = 0;
    while (shard + 1 < lay.n_shards && i >= lay.first_instance[shard + 1]) {
        base += *(const uint64_t *)(staged + lay.block_offset[shard] + lay.total_offset[shard]);
        shard++;
    }
    const uint32_t local = i - lay.first_instance[shard];
    const uint8_t *block = staged + lay.block_offset[shard];
    const uint32_t *src = (const uint32_t *)(block + (uint64_t)local * lay.stride);
    uint32_t *dst = (uint32_t *)(commands + (uint64_t)i * lay.stride);
    const uint32_t words = lay.stride / 4, first_task_word = lay.offset / 4 + 1;
    for (uint32_t w = 0; w < words; w++) {
        uint32_t x = src[w];
        if (lay.form != MCULL_CMD_EXT_GROUPS && w == first_task_word) x += (uint32_t)base;
        dst[w] = x;
    }
    if (lay.form == MCULL_CMD_EXT_GROUPS)
        first_task[i] = ((const uint32_t *)(block + lay.first_task_offset[shard]))[local] + (uint32_t)base;
}

// Destinations of one shard's block: the staging buffer of every device, at this shard's offset.
struct PushTargets {
    uint32_t n;
    uint4 *dst[kMaxShards];
};

// One launch sends a shard's block to every device: workgroup (x, y) stores chunk x of the block into
// destination y.  Remote destinations are written straight over the link of that pair (peer access is
// enabled), so the n copies of a block leave on n different links at once; the stores become visible to the
// destination's next kernel through the end-of-kernel release and the event the destination waits on.
__global__ void group_push_kernel(const uint4 *__restrict__ block, uint64_t n_vec, PushTargets t) {
    uint4 *__restrict__ dst = t.dst[blockIdx.y];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = block[i];
}

// The shards' depth images of one frame (each holds only its own instances) merged into the depth image of the
// whole scene: reverse-Z, larger = nearer, and a rasterised depth is a non-negative float, so the nearest
// surface is the maximum of the bit patterns - the same merge the rasteriser's atomicMax performs inside a shard.
struct DepthSources {
    uint32_t n;
    const uint32_t *src[kMaxShards];
};

__global__ void group_depth_max_kernel(uint32_t *__restrict__ dst, DepthSources s, uint64_t n_words) {
    const uint64_t n_vec = n_words / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = tid; i < n_vec; i += stride) {
        uint4 m = reinterpret_cast<const uint4 *>(s.src[0])[i];
        for (uint32_t r = 1; r < s.n; r++) {
            const uint4 v = reinterpret_cast<const uint4 *>(s.src[r])[i];
            m.x = max(m.x, v.x); m.y = max(m.y, v.y); m.z = max(m.z, v.z); m.w = max(m.w, v.w);
        }
        reinterpret_cast<uint4 *>(dst)[i] = m;
    }
    for (uint64_t i = n_vec * 4 + tid; i < n_words; i += stride) {
        uint32_t m = s.src[0][i];
        for (uint32_t r = 1; r < s.n; r++) m = max(m, s.src[r][i]);
        dst[i] = m;
    }
}

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct mcull_group {
    uint32_t n = 0;
    std::string err;
    struct Shard {
        int device = 0;
        hipStream_t stream = nullptr;
        mcull_ctx *ctx = nullptr;
        hipEvent_t pushed = nullptr, step = nullptr;  // step: this shard's rasteriser / merge of mcull_group_frame is done
        Buf merged, peer_depth;                       // the scene's depth image; copies of the other shards' images (no peer access)
        // blocks are received into staged[frame & 1]: a frame's blocks never land in the buffer the previous
        // frame's packing kernel may still be reading, and the one before that is done on every device by the
        // time anybody sends again (each sender has by then waited for all blocks of the frame in between)
        Buf send, staged[2], commands, first_task, totals;
        mcull_cull_result local{};
    } shard[kMaxShards];
    GroupLayout lay{};
    bool have_result = false;
    // the n-to-n "all blocks have arrived" dependency goes through one stream: it waits for every shard's
    // push, records `arrived`, and every shard waits for that one event (2n stream waits per frame, not n*n)
    hipStream_t join = nullptr;
    hipEvent_t arrived = nullptr, all_merged = nullptr;
    bool merged_recorded = false;
    int join_device = 0;
    bool direct = true;   // every pair of devices can store into each other's memory: one push kernel per shard
    uint64_t frame = 0;
};

namespace {

int gfail(mcull_group *g, int code, const char *fmt, ...) {
    if (g) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        g->err = buf;
    }
    return code;
}

#define GHIP(g, call)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return gfail((g), MCULL_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline bool too_small(const Buf &b, size_t bytes) { return !b.p || b.cap < (bytes ? bytes : 256); }

// grows (never shrinks) a device buffer of the current device; new memory is zeroed on `st`
int grow(mcull_group *g, Buf &b, size_t bytes, hipStream_t st) {
    if (bytes == 0) bytes = 256;
    if (b.p && b.cap >= bytes) return MCULL_OK;
    if (b.p) GHIP(g, hipFree(b.p));  // the caller has drained every stream of the group
    b.p = nullptr;
    b.cap = 0;
    const size_t want = (bytes + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return gfail(g, e == hipErrorOutOfMemory ? MCULL_E_NOMEM : MCULL_E_HIP, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    GHIP(g, hipMemsetAsync(b.p, 0, want, st));
    return MCULL_OK;
}

inline uint64_t pad16(uint64_t x) { return (x + 15) & ~uint64_t(15); }

}  // namespace

extern "C" {

int mcull_group_create(const int *device_ordinals, uint32_t n_shards, mcull_group **out) {
    if (!out) return MCULL_E_INVALID;
    *out = nullptr;
    if (!device_ordinals || n_shards == 0 || n_shards > kMaxShards) return MCULL_E_INVALID;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MCULL_E_HIP;
    for (uint32_t r = 0; r < n_shards; r++)
        if (device_ordinals[r] < 0 || device_ordinals[r] >= count) return MCULL_E_INVALID;
    mcull_group *g = new (std::nothrow) mcull_group();
    if (!g) return MCULL_E_NOMEM;
    g->n = n_shards;
    for (uint32_t r = 0; r < n_shards; r++) {
        mcull_group::Shard &s = g->shard[r];
        s.device = device_ordinals[r];
        bool ok = hipSetDevice(s.device) == hipSuccess &&
                  hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&s.pushed, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&s.step, hipEventDisableTiming) == hipSuccess &&
                  mcull_create(s.device, s.stream, &s.ctx) == MCULL_OK;
        if (!ok) {
            mcull_group_destroy(g);
            return MCULL_E_HIP;
        }
        // direct access to every other device of the group (a pair that is already enabled, or the
        // device itself, is not an error)
        for (uint32_t q = 0; q < n_shards; q++) {
            if (device_ordinals[q] == s.device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, s.device, device_ordinals[q]) != hipSuccess || !can) {
                g->direct = false;  // this pair goes through hipMemcpyPeerAsync
                continue;
            }
            hipError_t e = hipDeviceEnablePeerAccess(device_ordinals[q], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                mcull_group_destroy(g);
                return MCULL_E_HIP;
            }
            (void)hipGetLastError();
        }
    }
    g->join_device = device_ordinals[0];
    if (hipSetDevice(g->join_device) != hipSuccess || hipStreamCreateWithFlags(&g->join, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g->arrived, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->all_merged, hipEventDisableTiming) != hipSuccess) {
        mcull_group_destroy(g);
        return MCULL_E_HIP;
    }
    // MCULL_GROUP_COPIES=1: exchange with one hipMemcpyPeerAsync per pair even where direct stores work
    const char *copies = std::getenv("MCULL_GROUP_COPIES");
    if (copies && copies[0] == '1') g->direct = false;
    *out = g;
    return MCULL_OK;
}

int mcull_group_destroy(mcull_group *g) {
    if (!g) return MCULL_E_INVALID;
    for (uint32_t r = 0; r < g->n; r++) {
        mcull_group::Shard &s = g->shard[r];
        (void)hipSetDevice(s.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
    }
    for (uint32_t r = 0; r < g->n; r++) {
        mcull_group::Shard &s = g->shard[r];
        (void)hipSetDevice(s.device);
        if (s.ctx) (void)mcull_destroy(s.ctx);
        for (Buf *b : {&s.send, &s.staged[0], &s.staged[1], &s.commands, &s.first_task, &s.totals, &s.merged, &s.peer_depth})
            if (b->p) (void)hipFree(b->p);
        if (s.pushed) (void)hipEventDestroy(s.pushed);
        if (s.step) (void)hipEventDestroy(s.step);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    (void)hipSetDevice(g->join_device);
    if (g->join) (void)hipStreamSynchronize(g->join);
    if (g->arrived) (void)hipEventDestroy(g->arrived);
    if (g->all_merged) (void)hipEventDestroy(g->all_merged);
    if (g->join) (void)hipStreamDestroy(g->join);
    delete g;
    return MCULL_OK;
}

uint32_t mcull_group_size(const mcull_group *g) { return g ? g->n : 0; }

mcull_ctx *mcull_group_ctx(mcull_group *g, uint32_t shard) { return (g && shard < g->n) ? g->shard[shard].ctx : nullptr; }

const char *mcull_group_last_error(const mcull_group *g) { return g ? g->err.c_str() : "null group"; }

int mcull_group_cull(mcull_group *g, const mcull_globals *globals, const mcull_cull_params *params, mcull_group_result *results) {
    if (!g) return MCULL_E_INVALID;
    if (!globals) return gfail(g, MCULL_E_INVALID, "globals is null");
    mcull_cull_params p;
    std::memset(&p, 0, sizeof p);
    if (params) p = *params;
    else p.tests = MCULL_TEST_ALL;
    if (p.compaction != MCULL_COMPACT_ORDERED)
        return gfail(g, MCULL_E_INVALID, "a group culls with ordered compaction only (firstTask must be a scan to be rebased)");
    if (p.list || p.commands || p.total) return gfail(g, MCULL_E_INVALID, "a group owns its output buffers: list, commands and total must be null");
    if (p.cmd_form > MCULL_CMD_NV_TASK_GROUPS) return gfail(g, MCULL_E_INVALID, "unknown command form %u", p.cmd_form);
    const uint32_t record = p.cmd_form == MCULL_CMD_EXT_GROUPS ? 12u : 8u;
    if (p.cmd_stride == 0) p.cmd_stride = record;
    if ((p.cmd_stride & 3u) || (p.cmd_offset & 3u) || p.cmd_offset + record > p.cmd_stride)
        return gfail(g, MCULL_E_INVALID, "command stride/offset must be 4-byte multiples with offset+%u <= stride", record);

    // the layout of this frame: where every shard's instances and block go
    GroupLayout &lay = g->lay;
    lay.n_shards = g->n;
    lay.stride = p.cmd_stride;
    lay.offset = p.cmd_offset;
    lay.form = p.cmd_form;
    uint64_t block_bytes[kMaxShards], staged_bytes = 0, n_total = 0;
    for (uint32_t r = 0; r < g->n; r++) {
        const uint64_t n_r = mcull_ctx_instances(g->shard[r].ctx);
        lay.first_instance[r] = (uint32_t)n_total;
        n_total += n_r;
        if (n_total > 0xffffffffull) return gfail(g, MCULL_E_INVALID, "more than 2^32 instances in the group");
        lay.first_task_offset[r] = pad16(n_r * p.cmd_stride);
        lay.total_offset[r] = lay.first_task_offset[r] + pad16(n_r * 4);
        block_bytes[r] = lay.total_offset[r] + 16;
        lay.block_offset[r] = staged_bytes;
        staged_bytes += block_bytes[r];
    }
    lay.first_instance[g->n] = (uint32_t)n_total;

    // a buffer that has to grow may still be the target of another device's stores or the source of a packing
    // kernel: drain the whole group before any of them is replaced
    bool regrow = false;
    for (uint32_t r = 0; r < g->n; r++) {
        const mcull_group::Shard &s = g->shard[r];
        regrow |= too_small(s.send, block_bytes[r]) || too_small(s.staged[0], staged_bytes) || too_small(s.staged[1], staged_bytes) ||
                  too_small(s.commands, n_total * p.cmd_stride) || too_small(s.first_task, (n_total + 1) * 4) ||
                  too_small(s.totals, (kMaxShards + 1) * 8);
    }
    if (regrow)
        if (int rc = mcull_group_synchronize(g)) return rc;

    // every shard: cull into its send block
    for (uint32_t r = 0; r < g->n; r++) {
        mcull_group::Shard &s = g->shard[r];
        GHIP(g, hipSetDevice(s.device));
        if (int rc = grow(g, s.send, block_bytes[r], s.stream)) return rc;
        if (int rc = grow(g, s.staged[0], staged_bytes, s.stream)) return rc;
        if (int rc = grow(g, s.staged[1], staged_bytes, s.stream)) return rc;
        if (int rc = grow(g, s.commands, n_total * p.cmd_stride, s.stream)) return rc;
        if (int rc = grow(g, s.first_task, (n_total + 1) * 4, s.stream)) return rc;
        if (int rc = grow(g, s.totals, (kMaxShards + 1) * 8, s.stream)) return rc;
        mcull_cull_params pr = p;
        pr.commands = s.send.p;
        pr.total = (uint64_t *)((uint8_t *)s.send.p + lay.total_offset[r]);
        if (int rc = mcull_cull(s.ctx, globals, &pr, &s.local))
            return gfail(g, rc, "shard %u: %s", r, mcull_last_error(s.ctx));
        if (p.cmd_form == MCULL_CMD_EXT_GROUPS) {
            const uint64_t n_r = lay.first_instance[r + 1] - lay.first_instance[r];
            uint8_t *dst = (uint8_t *)s.send.p + lay.first_task_offset[r];
            if (s.local.first_task)
                GHIP(g, hipMemcpyAsync(dst, s.local.first_task, n_r * 4, hipMemcpyDeviceToDevice, s.stream));
            else
                GHIP(g, hipMemsetAsync(dst, 0, n_r * 4 ? n_r * 4 : 4, s.stream));
        }
    }
    // every shard sends its block to every device
    const uint32_t slot = (uint32_t)(g->frame & 1);
    for (uint32_t r = 0; r < g->n; r++) {
        mcull_group::Shard &s = g->shard[r];
        GHIP(g, hipSetDevice(s.device));
        if (g->direct) {
            PushTargets pt;
            pt.n = g->n;
            for (uint32_t d = 0; d < g->n; d++) pt.dst[d] = (uint4 *)((uint8_t *)g->shard[d].staged[slot].p + lay.block_offset[r]);
            const uint64_t n_vec = block_bytes[r] / 16;
            const uint32_t threads = 256;
            uint32_t chunks = (uint32_t)((n_vec + threads * 4 - 1) / (threads * 4));
            if (chunks == 0) chunks = 1;
            if (chunks > 256) chunks = 256;
            hipLaunchKernelGGL(group_push_kernel, dim3(chunks, g->n), dim3(threads), 0, s.stream, (const uint4 *)s.send.p, n_vec, pt);
            GHIP(g, hipGetLastError());
        } else {
            for (uint32_t d = 0; d < g->n; d++) {
                mcull_group::Shard &t = g->shard[d];
                uint8_t *dst = (uint8_t *)t.staged[slot].p + lay.block_offset[r];
                if (t.device == s.device)
                    GHIP(g, hipMemcpyAsync(dst, s.send.p, block_bytes[r], hipMemcpyDeviceToDevice, s.stream));
                else
                    GHIP(g, hipMemcpyPeerAsync(dst, t.device, s.send.p, s.device, block_bytes[r], s.stream));
            }
        }
        GHIP(g, hipEventRecord(s.pushed, s.stream));
    }
    // every device: wait for all blocks, then rebase and pack them
    if (g->n > 1) {
        GHIP(g, hipSetDevice(g->join_device));
        for (uint32_t r = 0; r < g->n; r++) GHIP(g, hipStreamWaitEvent(g->join, g->shard[r].pushed, 0));
        GHIP(g, hipEventRecord(g->arrived, g->join));
    }
    for (uint32_t d = 0; d < g->n; d++) {
        mcull_group::Shard &t = g->shard[d];
        GHIP(g, hipSetDevice(t.device));
        if (g->n > 1) GHIP(g, hipStreamWaitEvent(t.stream, g->arrived, 0));
        const uint32_t threads = 256;
        const uint64_t work = n_total > g->n ? n_total : g->n;
        const uint32_t blocks = (uint32_t)((work + threads - 1) / threads);
        hipLaunchKernelGGL(group_pack_kernel, dim3(blocks), dim3(threads), 0, t.stream, (const uint8_t *)t.staged[slot].p, lay,
                           (uint8_t *)t.commands.p, (uint32_t *)t.first_task.p, (uint64_t *)t.totals.p);
        GHIP(g, hipGetLastError());
        if (results) {
            mcull_group_result &o = results[d];
            o.local = t.local;
            o.commands = t.commands.p;
            o.first_task = p.cmd_form == MCULL_CMD_EXT_GROUPS ? (const uint32_t *)t.first_task.p : nullptr;
            o.totals = (const uint64_t *)t.totals.p;
            o.n_instances = n_total;
            o.first_instance = lay.first_instance[d];
        }
    }
    g->have_result = true;
    g->frame++;
    return MCULL_OK;
}


int mcull_group_frame(mcull_group *g, const mcull_globals *globals, uint32_t width, uint32_t height, uint32_t prepass_tests,
                      uint32_t view, const mcull_cull_params *params, mcull_group_result *results) {
    if (!g) return MCULL_E_INVALID;
    if (!globals) return gfail(g, MCULL_E_INVALID, "globals is null");
    if (width == 0 || height == 0 || width > 16384 || height > 16384) return gfail(g, MCULL_E_INVALID, "depth size %ux%u", width, height);
    if (view >= MCULL_MAX_VIEWS) return gfail(g, MCULL_E_INVALID, "view slot %u out of range", view);
    const uint64_t n_words = (uint64_t)width * height;
    // buffers first (growing one drains the whole group)
    bool regrow = false;
    for (uint32_t r = 0; r < g->n; r++) {
        regrow |= too_small(g->shard[r].merged, n_words * 4) || (!g->direct && too_small(g->shard[r].peer_depth, n_words * 4 * g->n));
        // a shard's own depth image is reallocated by its rasteriser when the size changes, while the other
        // devices may still be merging last frame's image over the links (their wait on all_merged orders
        // streams, not this thread's hipFree): drain the group before that, too
        const float *d = nullptr;
        uint32_t dw = 0, dh = 0;
        if (mcull_depth_buffer(g->shard[r].ctx, &d, &dw, &dh) == MCULL_OK) regrow |= dw != width || dh != height;
    }
    if (regrow)
        if (int rc = mcull_group_synchronize(g)) return rc;
    for (uint32_t r = 0; r < g->n; r++) {
        mcull_group::Shard &s = g->shard[r];
        GHIP(g, hipSetDevice(s.device));
        if (int rc = grow(g, s.merged, n_words * 4, s.stream)) return rc;
        if (!g->direct)
            if (int rc = grow(g, s.peer_depth, n_words * 4 * g->n, s.stream)) return rc;
    }
    // phase 1 on every shard: cull without history, rasterise the survivors into the shard's own depth image.
    // The image is cleared first, so the other devices must be done reading the last frame's.
    const uint32_t *own_depth[kMaxShards];
    mcull_cull_params pre;
    std::memset(&pre, 0, sizeof pre);
    pre.tests = prepass_tests;
    pre.hiz_view = view;
    for (uint32_t r = 0; r < g->n; r++) {
        mcull_group::Shard &s = g->shard[r];
        GHIP(g, hipSetDevice(s.device));
        if (g->n > 1 && g->merged_recorded) GHIP(g, hipStreamWaitEvent(s.stream, g->all_merged, 0));
        if (int rc = mcull_cull(s.ctx, globals, &pre, nullptr)) return gfail(g, rc, "shard %u: %s", r, mcull_last_error(s.ctx));
        if (int rc = mcull_depth_rasterize(s.ctx, globals, width, height, 1)) return gfail(g, rc, "shard %u: %s", r, mcull_last_error(s.ctx));
        const float *d = nullptr;
        if (int rc = mcull_depth_buffer(s.ctx, &d, nullptr, nullptr)) return gfail(g, rc, "shard %u: %s", r, mcull_last_error(s.ctx));
        own_depth[r] = reinterpret_cast<const uint32_t *>(d);
        GHIP(g, hipEventRecord(s.step, s.stream));
    }
    if (g->n > 1) {
        GHIP(g, hipSetDevice(g->join_device));
        for (uint32_t r = 0; r < g->n; r++) GHIP(g, hipStreamWaitEvent(g->join, g->shard[r].step, 0));
        GHIP(g, hipEventRecord(g->arrived, g->join));
    }
    // every device merges all shards' images (read in place over the links, or copied first) and builds the pyramid
    for (uint32_t d = 0; d < g->n; d++) {
        mcull_group::Shard &t = g->shard[d];
        GHIP(g, hipSetDevice(t.device));
        if (g->n > 1) GHIP(g, hipStreamWaitEvent(t.stream, g->arrived, 0));
        DepthSources ds;
        ds.n = g->n;
        for (uint32_t r = 0; r < g->n; r++) {
            ds.src[r] = own_depth[r];
            if (!g->direct && g->shard[r].device != t.device) {
                uint32_t *copy = (uint32_t *)t.peer_depth.p + (uint64_t)r * n_words;
                GHIP(g, hipMemcpyPeerAsync(copy, t.device, own_depth[r], g->shard[r].device, n_words * 4, t.stream));
                ds.src[r] = copy;
            }
        }
        const uint32_t threads = 256;
        uint64_t blocks = (n_words / 4 + threads - 1) / threads;
        if (blocks == 0) blocks = 1;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(group_depth_max_kernel, dim3((uint32_t)blocks), dim3(threads), 0, t.stream, (uint32_t *)t.merged.p, ds, n_words);
        GHIP(g, hipGetLastError());
        GHIP(g, hipEventRecord(t.step, t.stream));
        if (int rc = mcull_hiz_build(t.ctx, view, (const float *)t.merged.p, width, height, MCULL_MEM_DEVICE))
            return gfail(g, rc, "shard %u: %s", d, mcull_last_error(t.ctx));
    }
    if (g->n > 1) {
        GHIP(g, hipSetDevice(g->join_device));
        for (uint32_t r = 0; r < g->n; r++) GHIP(g, hipStreamWaitEvent(g->join, g->shard[r].step, 0));
        GHIP(g, hipEventRecord(g->all_merged, g->join));
        g->merged_recorded = true;
    }
    // phase 2: the cull with occlusion against the scene's pyramid, and the exchange
    mcull_cull_params p;
    std::memset(&p, 0, sizeof p);
    if (params) p = *params;
    else p.tests = MCULL_TEST_ALL;
    p.hiz_view = view;
    return mcull_group_cull(g, globals, &p, results);
}

int mcull_group_synchronize(mcull_group *g) {
    if (!g) return MCULL_E_INVALID;
    for (uint32_t r = 0; r < g->n; r++) {
        GHIP(g, hipSetDevice(g->shard[r].device));
        GHIP(g, hipStreamSynchronize(g->shard[r].stream));
    }
    return MCULL_OK;
}

int mcull_group_download(mcull_group *g, uint32_t shard, void *commands, uint64_t command_bytes, uint32_t *first_task,
                         uint64_t *totals) {
    if (!g) return MCULL_E_INVALID;
    if (shard >= g->n) return gfail(g, MCULL_E_INVALID, "shard %u out of range", shard);
    if (!g->have_result) return gfail(g, MCULL_E_STATE, "no group cull result");
    mcull_group::Shard &s = g->shard[shard];
    const uint64_t n_total = g->lay.first_instance[g->n];
    GHIP(g, hipSetDevice(s.device));
    if (commands) {
        const uint64_t have = n_total * g->lay.stride;
        if (command_bytes > have) command_bytes = have;
        if (command_bytes) GHIP(g, hipMemcpyAsync(commands, s.commands.p, command_bytes, hipMemcpyDeviceToHost, s.stream));
    }
    if (first_task) {
        if (g->lay.form != MCULL_CMD_EXT_GROUPS) return gfail(g, MCULL_E_STATE, "first_task is gathered for EXT records only");
        if (n_total) GHIP(g, hipMemcpyAsync(first_task, s.first_task.p, n_total * 4, hipMemcpyDeviceToHost, s.stream));
    }
    if (totals) GHIP(g, hipMemcpyAsync(totals, s.totals.p, (g->n + 1) * 8, hipMemcpyDeviceToHost, s.stream));
    GHIP(g, hipStreamSynchronize(s.stream));
    return MCULL_OK;
}

}  // extern "C"
