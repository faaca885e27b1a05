// C ABI of libmcull (include/mcull.h): context, buffers, host-side uniform math, launch sequencing.
// No CPU fallback exists anywhere in this library: every entry point that computes runs the
// gfx950 kernels, and fails with MCULL_E_HIP when no device is usable.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "mcull_internal.h"

#define MCULL_VERSION_NUMBER 110  // 110: mcull_scene_load(expected), mcull_cull_params.flags, MCULL_CMD_NV_TASK_GROUPS

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool owned = true;
};

struct HizSlot {
    DevBuf pyramid, table, strips;  // strips: largest depth per 64 x 8 strip of the depth image (large images only)
    DevBuf windows;                 // sliding-window minima (mcull::kHizWinBase), made by the first cull that wants them
    bool windows_current = false;   // ... from the pyramid as it was last built through this context
    uint32_t culls_since_build = 0; // culls with an instance level and the occlusion test that have used this build
    bool has_strips = false;
    uint32_t w = 0, h = 0, levels = 0;
    uint32_t lw[MCULL_MAX_HIZ_LEVELS], lh[MCULL_MAX_HIZ_LEVELS];
    uint64_t loff[MCULL_MAX_HIZ_LEVELS];
    uint64_t total_floats = 0;
    bool built = false;
};

}  // namespace

// Scratch of mcull_cull_static, the entry point that allocates nothing: the visibility ballots, the tile
// counters and their exclusive prefix, and the per-instance list offsets of one ordered cull of at most
// MCULL_STATIC_MAX_GROUPS groups and MCULL_STATIC_MAX_INSTANCES instances.  Every GPU the library runs on holds
// its own copy of this array until the process ends.  Host code never uses the array's name as an address:
// mcull_create asks the runtime for the copy's device address once per device (scratch_address) and every
// later use goes through that pointer.
namespace {
constexpr size_t kStaticMaskBytes = (size_t)MCULL_STATIC_MAX_GROUPS * 8;
constexpr size_t kStaticTileBytes = ((size_t)MCULL_STATIC_MAX_GROUPS / 256 + 1) * 4;      // one counter per 256 groups
constexpr size_t kStaticOffsetsBytes = ((size_t)MCULL_STATIC_MAX_GROUPS / 256 + 1) * 8;     // one 64-bit list offset per tile
constexpr size_t kStaticFirstBytes = ((size_t)MCULL_STATIC_MAX_INSTANCES + 1) * 4;
constexpr size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
constexpr size_t kStaticMaskOff = 0;
constexpr size_t kStaticTileOff = kStaticMaskOff + align256(kStaticMaskBytes);
constexpr size_t kStaticOffsetsOff = kStaticTileOff + align256(kStaticTileBytes);
constexpr size_t kStaticFirstOff = kStaticOffsetsOff + align256(kStaticOffsetsBytes);
constexpr size_t kStaticBytes = kStaticFirstOff + align256(kStaticFirstBytes);
static_assert(kStaticBytes == MCULL_STATIC_SCRATCH_BYTES, "mcull.h states the size of the static scratch");
}  // namespace

static __device__ __align__(256) unsigned char scratch[kStaticBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_lock;
unsigned char *g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each device, once resolved
// Who used a device's array last: an event recorded behind that call's launches (created with the address, in
// mcull_create, so that mcull_cull_static itself creates nothing) and the stream it was recorded on.
struct ScratchGuard {
    std::mutex lock;
    hipEvent_t done = nullptr;
    hipStream_t stream = nullptr;
    bool recorded = false;
};
ScratchGuard g_scratch_guard[kMaxDevices];

// Device address of this device's copy of `scratch`; the device must be current.
hipError_t scratch_address(int device, unsigned char **out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
        if (!g_scratch_guard[device].done) (void)hipEventCreateWithFlags(&g_scratch_guard[device].done, hipEventDisableTiming);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace

struct mcull_ctx {
    int device = 0;
    unsigned char *static_scratch = nullptr;  // this device's copy of `scratch` (never the array's name)
    int cu_count = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;       // a pair of timestamps has been recorded
    bool timing = false;      // record timestamps around kernels
    std::string err;

    // scene
    DevBuf aos, cone, sphere, vertices, mdata, transforms, xf_scale, ranges, group_base, group_rec, next_nonempty;
    uint64_t n_meshlets = 0, n_vertices = 0, n_words = 0, n_instances = 0;
    bool have_aos = false, have_bounds = false, have_geometry = false, have_ranges = false;
    bool geometry_checked = false;  // the uploaded meshlets index only inside the uploaded geometry (host check done)
    uint64_t ranged_groups = 0, ranged_pairs = 0, ranged_max_end = 0, ranged_instances = 0;
    uint32_t ranged_longest = 0;  // meshlets of the longest range
    // instance pre-cull: one enclosing sphere per instance range (ranged scenes) or one for [0, meshletCount);
    // computed on the first cull after the bounds or the ranges changed
    DevBuf range_spheres;
    bool spheres_valid = false, spheres_ranged = false;
    uint32_t spheres_uniform_count = 0;
    // group level of the pre-cull (scenes whose instances span many groups): sphere m encloses the meshlets [m, m + 64)
    DevBuf group_spheres;
    bool gspheres_valid = false;
    bool precull_groups = true;  // MCULL_PRECULL_GROUPS=0: no group level (measurements)
    long long precull_estimate = -1;  // MCULL_PRECULL_ESTIMATE=n: the work list is taken to hold n groups (0: no estimate) instead of what the last cull reported (tests)
    // A pre-cull that removes next to nothing costs its own pass and a slower mask pass (100 M pairs, all in
    // view: 0.77 against 0.69 ms).  Each single-view cull that runs it leaves the number of units it kept in
    // host-mapped memory (kept_host[slot], read later without waiting); when that was more than 90 % of the
    // units, the following kPrecullCooldown culls of the slot go without, then it is tried again.
    // The group level behind it is judged the same way on its own: it reports how many groups it was handed
    // (kept_host[2]), and when more than 90 % of them were still in the list the mask pass got, the following
    // kPrecullCooldown culls run the instance level alone.
    // MCULL_PRECULL_ADAPTIVE=0 switches the back-off off (tests that look at the pre-cull's statistics).
    volatile uint32_t *kept_host = nullptr;  // MCULL_MAX_VIEWS words, hipHostMalloc (mapped): [0] single-view culls, [1] multi-view, [2] group level
    uint32_t *kept_dev = nullptr;            // the same memory as the device sees it
    bool precull_adaptive = true;
    // the scatter pass's rounds of 32 groups for lists expected to be sparse: -1 = by the list estimate, MCULL_COMPACT_SHORT_ROUNDS=1 / 0 = always / never
    int compact_short = -1;

    // Hi-Z
    HizSlot hiz[MCULL_MAX_VIEWS];
    DevBuf depth_staging, hiz_counter;
    DevBuf depth;  // rasterised depth image (mcull_depth_rasterize)
    DevBuf raster_scratch;  // the rasteriser's queue of huge triangles
    DevBuf raster_mvp;      // the rasteriser's per-instance (proj*view)*model matrices
    DevBuf view_dev;  // MCULL_MAX_VIEWS + 1 device-resident ViewConst records (the *_device_globals calls; the last one is the rasteriser's)
    uint32_t depth_w = 0, depth_h = 0;
    int hiz_mode = 0;  // pyramid scheme (see launch_hiz): 0 chosen by image size; 1-3 force one of the schemes
    bool hiz_windows = true;  // MCULL_HIZ_WINDOWS=0: the instance and group levels prove occlusion from the pyramid's own levels only (measurements)
    bool hiz_strips = true;  // MCULL_HIZ_STRIPS=0: no strip maxima, every occlusion test reads its texel (measurements)
    bool precull_flags = true;  // MCULL_PRECULL_FLAGS=0: the instance level only drops instances, it narrows no tests (measurements)
    DevBuf view_flags;          // multi-view culls: one byte per instance and view slot (classify_views_kernel)
    bool views_probe_pending = false;  // a multi-view pre-pass whose yield has not been looked at yet (kept_host[1])
    uint64_t views_probe_units = 0;
    uint32_t views_cooldown = 0;
    // Scenes below this many 64-meshlet groups are launch-bound: the instance pre-cull's extra launch and memset
    // (8 - 15 us) cost more than the groups it removes (the mask pass does about 4000 groups per microsecond; the
    // reference's scene, 6000 groups: 76 us per two-phase frame without it, 100 us with it).
    // MCULL_PRECULL_MIN_GROUPS overrides it (tests).
    uint64_t precull_min_groups = 65536;
    bool force_streaming = false;  // MCULL_FORCE_STREAMING=1: the kernels for bounds larger than the caches, whatever the size (tests)

    // cull scratch + outputs, one set per view slot: mcull_cull uses slot 0, mcull_cull_views slot i
    // for its i-th view; `cur` is the slot the download / rebase / payload / rasterise calls refer to
    struct Last {
        bool valid = false;
        mcull::SceneDev scene{};
        uint32_t uniform_count = 0;
        uint32_t compaction = 0;
        mcull::CmdLayout lay{8, 0, 0};
        mcull_visible *list = nullptr;
        uint64_t capacity = 0;
        void *commands = nullptr;
        bool have_mask = false;
        uint64_t n_pairs = 0;
        // the instance pre-cull of this cull: its work list over `precull_units` groups (ranged scenes) or
        // instances; nullptr when it did not run
        const uint32_t *work = nullptr;
        uint64_t precull_units = 0;
        const uint32_t *group_work = nullptr;  // the group level's list (one entry per surviving group), when it ran
    };
    struct ViewOut {
        DevBuf mask, tile_counts, tile_offsets, inst_first, inst_count, total, list, commands, work, group_work;
        uint64_t commands_layout = 0;  // stride/offset the context-owned command buffer was last zeroed for
        bool counters_clean = false;   // tile counters are all zero (the mask pass adds into them, the command pass re-zeroes)
        bool work_clean = false;       // word 0 of the work list is zero (the pre-pass reserves from it, the command pass re-zeroes)
        uint64_t *total_ptr = nullptr;  // where the last cull wrote its count (context- or caller-owned)
        Last last;                      // description of the last cull (for rebase / downloads / payloads)
        bool counts_slow_pairs = false; // the last cull counted the pairs that left the fast predicate forms
        bool probe_pending = false;     // a pre-cull ran and its kept count has not been looked at yet
        uint64_t probe_units = 0;
        // per set of tests (index tests & 7), like the list estimates below: that the planes alone remove nothing says nothing
        // about a cull of the same scene with the occlusion test
        uint32_t precull_cooldown[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // culls still to go without the pre-cull
        uint32_t group_cooldown[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // culls still to go without the group level (it removed less than a tenth of what it was handed)
        // what the work list of the slot's last finished pre-culled cull held, in groups (0 = unknown): the streaming
        // mask pass sizes its grid for it (launch_cull_mask)
        bool probe_in_groups = false, probe_group_level = false;
        uint32_t probe_groups_per_instance = 0;
        // one estimate per set of tests: the two phases of a frame (frustum + cone, then all three) alternate on one slot, and
        // the short list of the second must not size the grid of the first (100 M pairs: 0.798 instead of 0.703 ms per pair of
        // culls while they shared one)
        uint64_t list_estimate[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t probe_tests = 0;          // the tests of the cull that owns, or last owned, the probe
        bool other_tests_waiting = false;  // a cull with other tests passed by while that probe was pending: its turn next
        uint32_t probe_deferred = 0;       // times in a row the last owner has left the probe to the others
    } out[MCULL_MAX_VIEWS];
    uint32_t cur = 0;
};

namespace {

int fail(mcull_ctx *c, int code, const char *fmt, ...) {
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

#define HIPCHK(c, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail((c), MCULL_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int ensure(mcull_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 256;
    if (b.owned && b.p && b.cap >= bytes) return MCULL_OK;
    if (b.p && b.owned) {
        hipError_t e = hipFree(b.p);
        if (e != hipSuccess) return fail(c, MCULL_E_HIP, "hipFree failed: %s", hipGetErrorString(e));
    }
    b.p = nullptr; b.cap = 0; b.owned = true;
    const size_t want = (bytes + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, e == hipErrorOutOfMemory ? MCULL_E_NOMEM : MCULL_E_HIP, "hipMalloc(%zu) failed: %s", want,
                    hipGetErrorString(e));
    }
    b.cap = want;
    return MCULL_OK;
}

void release(DevBuf &b) {
    if (b.p && b.owned) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0; b.owned = true;
}

// a changed scene makes every stored cull result meaningless
// keep_hints: the scene has the same shape as before (new matrices for the same instances - an animated frame): what the
// pre-cull learnt about the last frame is still the best guess for this one, and no result depends on it
void invalidate_results(mcull_ctx *c, bool keep_hints = false) {
    for (auto &o : c->out) o.last.valid = false;
    c->cur = 0;
    if (keep_hints) return;
    for (auto &o : c->out) {
        // what the pre-cull learnt about the old scene says nothing about the new one
        o.probe_pending = false;
        for (auto &e : o.list_estimate) e = 0;
        o.other_tests_waiting = false;
        o.probe_deferred = 0;
        for (int k = 0; k < 8; k++) o.precull_cooldown[k] = o.group_cooldown[k] = 0;
    }
    c->views_probe_pending = false;
    c->views_cooldown = 0;
    c->cur = 0;
}

inline bool mem_kind_ok(int mem) { return mem == MCULL_MEM_HOST || mem == MCULL_MEM_DEVICE; }

int copy_in(mcull_ctx *c, void *dst, const void *src, size_t bytes, int mem) {
    if (!mem_kind_ok(mem)) return fail(c, MCULL_E_INVALID, "mem must be MCULL_MEM_HOST or MCULL_MEM_DEVICE, not %d", mem);
    if (bytes == 0) return MCULL_OK;
    if (mem == MCULL_MEM_HOST) {
        // the caller may free / reuse a host pointer as soon as the call returns
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else {
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    return MCULL_OK;
}

// ---- uniform host math, IEEE binary32, fixed association order, no fused multiply-add ----
__host__ __device__ inline float Ec(const float *m, int r, int col) { return m[4 * col + r]; }

// proj * view (shader.task:60 evaluates (proj*view)*v): column c of the product is
// ((P[0]*V[c][0] + P[1]*V[c][1]) + P[2]*V[c][2]) + P[3]*V[c][3]
__host__ __device__ inline void mat4_mul(const float *a, const float *b, float *out) {
    float r[16];
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++)
            r[4 * col + row] = ((Ec(a, row, 0) * Ec(b, 0, col) + Ec(a, row, 1) * Ec(b, 1, col)) +
                                Ec(a, row, 2) * Ec(b, 2, col)) + Ec(a, row, 3) * Ec(b, 3, col);
    for (int i = 0; i < 16; i++) out[i] = r[i];
}

// inverse(view)[3].xyz (shader.task:88-89) by cofactor expansion: inv(r,c) = cof(c,r) * (1/det);
// 3x3 minors expand along their first row, the determinant along row 0, sums left to right.
__host__ __device__ inline void camera_position(const float *view, float cam[3]) {
    float cof[4][4];
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) {
            float e[9];
            int k = 0;
            for (int r = 0; r < 4; r++) {
                if (r == i) continue;
                for (int col = 0; col < 4; col++) {
                    if (col == j) continue;
                    e[k++] = Ec(view, r, col);
                }
            }
            const float minor = (e[0] * (e[4] * e[8] - e[5] * e[7]) - e[1] * (e[3] * e[8] - e[5] * e[6])) +
                                e[2] * (e[3] * e[7] - e[4] * e[6]);
            cof[i][j] = ((i + j) & 1) ? -minor : minor;
        }
    }
    const float det = ((Ec(view, 0, 0) * cof[0][0] + Ec(view, 0, 1) * cof[0][1]) + Ec(view, 0, 2) * cof[0][2]) +
                      Ec(view, 0, 3) * cof[0][3];
    const float inv_det = 1.0f / det;
    for (int r = 0; r < 3; r++) cam[r] = cof[3][r] * inv_det;  // inv(r, 3) = cof(3, r) / det
}

// The view constants of a cull whose Globals live in device memory: the same uniform expressions as on
// the host (same functions, same rounding), evaluated by one thread when the stream reaches this point,
// so that a captured graph picks up whatever Globals the buffer holds at replay time.  `base` carries
// what the host knows (tests, pyramid, meshlet count).
__global__ void derive_view_kernel(const mcull_globals *__restrict__ g, mcull::ViewConst base, mcull::ViewConst *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    mcull::ViewConst v = base;
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 4; j++) v.frustum[i][j] = g->frustum[i][j];
    mat4_mul(g->proj, g->view, v.vp);
    camera_position(g->view, v.cam);
    v.proj11 = g->proj[5];
    mcull::view_derive_bounds(v);
    *out = v;
}

// c->view_dev: MCULL_MAX_VIEWS + 1 ViewConst records (the last one is the rasteriser's), then the multi-view pass's
// paired form of the first MCULL_MAX_VIEWS
constexpr size_t kViewDevHotOff = ((MCULL_MAX_VIEWS + 1) * sizeof(mcull::ViewConst) + 255) & ~size_t(255);
constexpr size_t kViewDevBytes = kViewDevHotOff + (MCULL_MAX_VIEWS / 2) * sizeof(mcull::ViewHot2) + 256;  // + the pass's uniforms

uint32_t level_count(uint32_t w, uint32_t h) {
    uint32_t m = w > h ? w : h, l = 0;
    while (m > 1) { m >>= 1; l++; }
    return l + 1;
}

}  // namespace

hipStream_t mcull_ctx_stream(mcull_ctx *ctx) { return ctx->stream; }
int mcull_ctx_device(mcull_ctx *ctx) { return ctx->device; }
uint64_t mcull_ctx_instances(mcull_ctx *ctx) { return ctx->n_instances; }
int mcull_ctx_fail(mcull_ctx *ctx, int code, const char *message) { return fail(ctx, code, "%s", message); }

extern "C" {

int mcull_version(void) { return MCULL_VERSION_NUMBER; }

const char *mcull_status_string(int status) {
    switch (status) {
        case MCULL_OK: return "MCULL_OK";
        case MCULL_E_INVALID: return "MCULL_E_INVALID";
        case MCULL_E_HIP: return "MCULL_E_HIP";
        case MCULL_E_CAPACITY: return "MCULL_E_CAPACITY";
        case MCULL_E_STATE: return "MCULL_E_STATE";
        case MCULL_E_NOMEM: return "MCULL_E_NOMEM";
        case MCULL_E_IO: return "MCULL_E_IO";
        default: return "MCULL_E_UNKNOWN";
    }
}

const char *mcull_last_error(const mcull_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int mcull_create(int device_ordinal, void *stream, mcull_ctx **out) {
    if (!out) return MCULL_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MCULL_E_HIP;
    if (device_ordinal < 0 || device_ordinal >= count) return MCULL_E_INVALID;
    mcull_ctx *c = new (std::nothrow) mcull_ctx();
    if (!c) return MCULL_E_NOMEM;
    c->device = device_ordinal;
    c->stream = (hipStream_t)stream;
    hipDeviceProp_t prop;
    if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return MCULL_E_HIP;
    }
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // the device is current: resolve the address of its copy of the static scratch once
    if (scratch_address(device_ordinal, &c->static_scratch) != hipSuccess) {
        (void)hipEventDestroy(c->ev0);
        (void)hipEventDestroy(c->ev1);
        delete c;
        return MCULL_E_HIP;
    }
    // The pyramid scheme is chosen by image size (launch_hiz); these switches force one of the schemes
    // for evaluation and tests (results are identical)
    if (const char *hs = std::getenv("MCULL_HIZ_STRIPS")) c->hiz_strips = hs[0] != '0';
    if (const char *hw = std::getenv("MCULL_HIZ_WINDOWS")) c->hiz_windows = hw[0] != '0';
    if (const char *pf = std::getenv("MCULL_PRECULL_FLAGS")) c->precull_flags = pf[0] != '0';
    if (const char *pg = std::getenv("MCULL_PRECULL_GROUPS")) c->precull_groups = pg[0] != '0';
    if (const char *pe = std::getenv("MCULL_PRECULL_ESTIMATE")) c->precull_estimate = std::atoll(pe);
    const char *tiles = std::getenv("MCULL_HIZ_TILES"), *streamed = std::getenv("MCULL_HIZ_STREAM");
    const char *single = std::getenv("MCULL_HIZ_SINGLE_LAUNCH");
    const char *fs = std::getenv("MCULL_FORCE_STREAMING");
    c->force_streaming = fs && fs[0] == '1';
    if (const char *pm = std::getenv("MCULL_PRECULL_MIN_GROUPS")) c->precull_min_groups = std::strtoull(pm, nullptr, 10);
    if (const char *pa = std::getenv("MCULL_PRECULL_ADAPTIVE")) c->precull_adaptive = pa[0] != '0';
    if (const char *cs = std::getenv("MCULL_COMPACT_SHORT_ROUNDS")) c->compact_short = cs[0] != '0' ? 1 : 0;
    void *host_words = nullptr, *dev_words = nullptr;
    if (hipHostMalloc(&host_words, MCULL_MAX_VIEWS * 4, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(&dev_words, host_words, 0) != hipSuccess) {
        if (host_words) (void)hipHostFree(host_words);
        (void)hipEventDestroy(c->ev0);
        (void)hipEventDestroy(c->ev1);
        delete c;
        return MCULL_E_HIP;
    }
    c->kept_host = static_cast<volatile uint32_t *>(host_words);
    c->kept_dev = static_cast<uint32_t *>(dev_words);
    for (int i = 0; i < MCULL_MAX_VIEWS; i++) c->kept_host[i] = 0xffffffffu;
    c->hiz_mode = (single && single[0] == '1') ? 2 : (tiles && tiles[0] == '1') ? 1 : (streamed && streamed[0] == '1') ? 3 : 0;
    *out = c;
    return MCULL_OK;
}

int mcull_destroy(mcull_ctx *c) {
    if (!c) return MCULL_E_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *bufs[] = {&c->aos, &c->cone, &c->sphere, &c->vertices, &c->mdata, &c->transforms, &c->xf_scale, &c->ranges,
                      &c->group_base, &c->group_rec, &c->next_nonempty, &c->depth_staging, &c->hiz_counter, &c->view_flags, &c->depth, &c->view_dev, &c->raster_scratch, &c->raster_mvp,
                      &c->range_spheres, &c->group_spheres};
    for (DevBuf *b : bufs) release(*b);
    for (auto &o : c->out)
        for (DevBuf *b : {&o.mask, &o.tile_counts, &o.tile_offsets, &o.inst_first, &o.inst_count, &o.total, &o.list, &o.commands, &o.work,
                           &o.group_work})
            release(*b);
    for (auto &s : c->hiz) { release(s.pyramid); release(s.table); release(s.strips); release(s.windows); }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->kept_host) (void)hipHostFree(const_cast<uint32_t *>(c->kept_host));
    delete c;
    return MCULL_OK;
}

int mcull_synchronize(mcull_ctx *c) {
    if (!c) return MCULL_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCULL_OK;
}

// ------------------------------------------------------------------------------------ uploads

int mcull_upload_meshlets(mcull_ctx *c, const mcull_meshlet *meshlets, uint64_t count, int mem) {
    if (!c) return MCULL_E_INVALID;
    if (!meshlets && count) return fail(c, MCULL_E_INVALID, "meshlets is null");
    if (!mem_kind_ok(mem)) return fail(c, MCULL_E_INVALID, "mem must be MCULL_MEM_HOST or MCULL_MEM_DEVICE, not %d", mem);
    if (count > 0xffffffffull) return fail(c, MCULL_E_INVALID, "meshlet ids are 32-bit: at most 2^32 meshlets");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->aos, count * sizeof(mcull_meshlet)))) return rc;
    if (!c->cone.owned) c->cone = DevBuf();
    if (!c->sphere.owned) c->sphere = DevBuf();
    if ((rc = ensure(c, c->cone, count * 16))) return rc;
    if ((rc = ensure(c, c->sphere, count * 16))) return rc;
    if ((rc = copy_in(c, c->aos.p, meshlets, count * sizeof(mcull_meshlet), mem))) return rc;
    HIPCHK(c, mcull::launch_split_meshlets((const mcull_meshlet *)c->aos.p, count, (float4 *)c->cone.p,
                                           (float4 *)c->sphere.p, c->stream));
    c->n_meshlets = count;
    c->have_aos = true;
    c->geometry_checked = false;
    c->have_bounds = true;
    c->spheres_valid = c->gspheres_valid = false;  // the pre-cull's range spheres derive from the bounds
    invalidate_results(c);
    return MCULL_OK;
}

int mcull_upload_bounds_soa(mcull_ctx *c, const float *cone_axis_w, const float *centre_radius, uint64_t count,
                            int mem, int adopt) {
    if (!c) return MCULL_E_INVALID;
    if ((!cone_axis_w || !centre_radius) && count) return fail(c, MCULL_E_INVALID, "null bounds stream");
    if (!mem_kind_ok(mem)) return fail(c, MCULL_E_INVALID, "mem must be MCULL_MEM_HOST or MCULL_MEM_DEVICE, not %d", mem);
    if (count > 0xffffffffull) return fail(c, MCULL_E_INVALID, "meshlet ids are 32-bit: at most 2^32 meshlets");
    HIPCHK(c, hipSetDevice(c->device));
    if (adopt && mem == MCULL_MEM_DEVICE) {
        if ((reinterpret_cast<uintptr_t>(cone_axis_w) | reinterpret_cast<uintptr_t>(centre_radius)) & 15u)
            return fail(c, MCULL_E_INVALID, "adopted bounds streams must be 16-byte aligned");
        release(c->cone); release(c->sphere);
        c->cone.p = const_cast<float *>(cone_axis_w); c->cone.cap = count * 16; c->cone.owned = false;
        c->sphere.p = const_cast<float *>(centre_radius); c->sphere.cap = count * 16; c->sphere.owned = false;
    } else {
        int rc;
        if (!c->cone.owned) c->cone = DevBuf();
        if (!c->sphere.owned) c->sphere = DevBuf();
        if ((rc = ensure(c, c->cone, count * 16))) return rc;
        if ((rc = ensure(c, c->sphere, count * 16))) return rc;
        if ((rc = copy_in(c, c->cone.p, cone_axis_w, count * 16, mem))) return rc;
        if ((rc = copy_in(c, c->sphere.p, centre_radius, count * 16, mem))) return rc;
    }
    c->n_meshlets = count;
    c->have_aos = false;
    c->have_bounds = true;
    c->spheres_valid = c->gspheres_valid = false;
    invalidate_results(c);
    return MCULL_OK;
}

int mcull_upload_geometry(mcull_ctx *c, const mcull_vertex *vertices, uint64_t vertex_count,
                          const uint32_t *meshlet_data, uint64_t word_count, int mem) {
    if (!c) return MCULL_E_INVALID;
    if ((!vertices && vertex_count) || (!meshlet_data && word_count)) return fail(c, MCULL_E_INVALID, "null geometry");
    if (!mem_kind_ok(mem)) return fail(c, MCULL_E_INVALID, "mem must be MCULL_MEM_HOST or MCULL_MEM_DEVICE, not %d", mem);
    if (vertex_count > 0xffffffffull || word_count > 0xffffffffull)
        return fail(c, MCULL_E_INVALID, "vertex ids and data offsets are 32-bit: at most 2^32 vertices / words");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->vertices, vertex_count * sizeof(mcull_vertex)))) return rc;
    if ((rc = ensure(c, c->mdata, word_count * 4 + 4))) return rc;
    if ((rc = copy_in(c, c->vertices.p, vertices, vertex_count * sizeof(mcull_vertex), mem))) return rc;
    if ((rc = copy_in(c, c->mdata.p, meshlet_data, word_count * 4, mem))) return rc;
    c->n_vertices = vertex_count;
    c->n_words = word_count;
    c->have_geometry = true;
    c->geometry_checked = false;
    return MCULL_OK;
}

// Host-side shape check shared by the kernels that walk meshlet geometry: every meshlet's block
// must lie inside meshletData, every global vertex id inside the vertex array and every local index
// below the meshlet's vertex count; the kernels index without bounds checks.  Done once per upload
// (the kernels rewrite only cone/apex/radius, never the fields checked here).
static int check_geometry(mcull_ctx *c) {
    if (!c->have_aos) return fail(c, MCULL_E_STATE, "mcull_upload_meshlets has not been called");
    if (!c->have_geometry) return fail(c, MCULL_E_STATE, "mcull_upload_geometry has not been called");
    if (c->geometry_checked) return MCULL_OK;
    std::vector<mcull_meshlet> ml;
    std::vector<uint32_t> md;
    try {  // nothing may throw across the C boundary
        ml.resize(c->n_meshlets);
        md.resize(c->n_words);
    } catch (const std::bad_alloc &) {
        return fail(c, MCULL_E_NOMEM, "no host memory to check %llu meshlets", (unsigned long long)c->n_meshlets);
    }
    HIPCHK(c, hipMemcpyAsync(ml.data(), c->aos.p, c->n_meshlets * sizeof(mcull_meshlet), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(md.data(), c->mdata.p, c->n_words * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint64_t i = 0; i < c->n_meshlets; i++) {
        const mcull_meshlet &m = ml[i];
        const uint64_t words = (uint64_t)m.vertexCount + ((uint64_t)m.triangleCount * 3 + 3) / 4;
        if (m.vertexCount > 64 || m.triangleCount > 124 || (uint64_t)m.dataOffset + words > c->n_words)
            return fail(c, MCULL_E_INVALID, "meshlet %llu: counts/dataOffset outside meshletData", (unsigned long long)i);
        const uint8_t *loc = reinterpret_cast<const uint8_t *>(md.data() + m.dataOffset + m.vertexCount);
        for (uint32_t v = 0; v < m.vertexCount; v++)
            if (md[m.dataOffset + v] >= c->n_vertices)
                return fail(c, MCULL_E_INVALID, "meshlet %llu: vertex id out of range", (unsigned long long)i);
        for (uint32_t t = 0; t < (uint32_t)m.triangleCount * 3; t++)
            if (loc[t] >= m.vertexCount)
                return fail(c, MCULL_E_INVALID, "meshlet %llu: local index out of range", (unsigned long long)i);
    }
    c->geometry_checked = true;
    return MCULL_OK;
}

int mcull_compute_bounds(mcull_ctx *c) {
    if (!c) return MCULL_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_geometry(c)) return rc;
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    HIPCHK(c, mcull::launch_bounds((const mcull_vertex *)c->vertices.p, (const uint32_t *)c->mdata.p,
                                   (mcull_meshlet *)c->aos.p, c->n_meshlets, (float4 *)c->cone.p, (float4 *)c->sphere.p,
                                   c->stream));
    if (c->timing) {
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        c->timed = true;
    }
    c->have_bounds = true;
    c->spheres_valid = c->gspheres_valid = false;
    return MCULL_OK;
}

int mcull_download_meshlets(mcull_ctx *c, mcull_meshlet *out, uint64_t count) {
    if (!c || !out) return MCULL_E_INVALID;
    if (!c->have_aos) return fail(c, MCULL_E_STATE, "no AoS meshlets in this context");
    if (count > c->n_meshlets) return fail(c, MCULL_E_INVALID, "count exceeds uploaded meshlets");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->aos.p, count * sizeof(mcull_meshlet), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCULL_OK;
}

int mcull_upload_transforms(mcull_ctx *c, const mcull_transform *transforms, uint64_t count, int mem) {
    if (!c) return MCULL_E_INVALID;
    if (!transforms && count) return fail(c, MCULL_E_INVALID, "transforms is null");
    if (count >= 0xffffffffull) return fail(c, MCULL_E_INVALID, "too many instances");
    if (!mem_kind_ok(mem)) return fail(c, MCULL_E_INVALID, "mem must be MCULL_MEM_HOST or MCULL_MEM_DEVICE, not %d", mem);
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->transforms, count * sizeof(mcull_transform)))) return rc;
    if ((rc = ensure(c, c->xf_scale, count * 4))) return rc;
    if ((rc = copy_in(c, c->transforms.p, transforms, count * sizeof(mcull_transform), mem))) return rc;
    HIPCHK(c, mcull::launch_xform_scale((const float4 *)c->transforms.p, (uint32_t)count, (float *)c->xf_scale.p,
                                        c->stream));
    const bool same_shape = count == c->n_instances;
    c->n_instances = count;
    invalidate_results(c, same_shape);
    return MCULL_OK;
}

int mcull_set_instance_ranges(mcull_ctx *c, const uint32_t *meshlet_offset, const uint32_t *meshlet_count,
                              uint64_t n_instances) {
    if (!c) return MCULL_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    invalidate_results(c);
    c->spheres_valid = c->gspheres_valid = false;
    if (!meshlet_offset || !meshlet_count) {
        c->have_ranges = false;
        return MCULL_OK;
    }
    if (n_instances >= 0xffffffffull) return fail(c, MCULL_E_INVALID, "too many instances");
    std::vector<uint32_t> rg, base, next;
    try {  // nothing may throw across the C boundary
        rg.resize(2 * n_instances);
        base.resize(n_instances + 1);
        next.resize(n_instances + 1);
    } catch (const std::bad_alloc &) {
        return fail(c, MCULL_E_NOMEM, "no host memory for %llu instance ranges", (unsigned long long)n_instances);
    }
    uint64_t groups = 0, pairs = 0, max_end = 0;
    uint32_t longest = 0;
    for (uint64_t i = 0; i < n_instances; i++) {
        rg[2 * i] = meshlet_offset[i];
        rg[2 * i + 1] = meshlet_count[i];
        base[i] = (uint32_t)groups;
        groups += ((uint64_t)meshlet_count[i] + 63) / 64;
        pairs += meshlet_count[i];
        const uint64_t end = (uint64_t)meshlet_offset[i] + meshlet_count[i];
        if (end > max_end) max_end = end;
        if (meshlet_count[i] > longest) longest = meshlet_count[i];
        if (groups >= 0xffffffffull) return fail(c, MCULL_E_INVALID, "more than 2^32 meshlet groups");
    }
    base[n_instances] = (uint32_t)groups;
    uint32_t nxt = (uint32_t)n_instances;  // next[i]: the first instance after i that owns meshlets
    for (uint64_t i = n_instances; i-- > 0;) {
        next[i] = nxt;
        if (meshlet_count[i]) nxt = (uint32_t)i;
    }
    int rc;
    if ((rc = ensure(c, c->next_nonempty, next.size() * 4))) return rc;
    if ((rc = copy_in(c, c->next_nonempty.p, next.data(), next.size() * 4, MCULL_MEM_HOST))) return rc;
    if ((rc = ensure(c, c->ranges, rg.size() * 4))) return rc;
    if ((rc = ensure(c, c->group_base, base.size() * 4))) return rc;
    if ((rc = ensure(c, c->group_rec, groups * 16))) return rc;
    if ((rc = copy_in(c, c->ranges.p, rg.data(), rg.size() * 4, MCULL_MEM_HOST))) return rc;
    if ((rc = copy_in(c, c->group_base.p, base.data(), base.size() * 4, MCULL_MEM_HOST))) return rc;
    HIPCHK(c, mcull::launch_fill_group_rec((const uint2 *)c->ranges.p, (const uint32_t *)c->group_base.p,
                                           (uint32_t)n_instances, (uint4 *)c->group_rec.p, c->stream));
    c->have_ranges = true;
    c->ranged_groups = groups;
    c->ranged_pairs = pairs;
    c->ranged_max_end = max_end;
    c->ranged_longest = longest;
    c->ranged_instances = n_instances;
    return MCULL_OK;
}

// ------------------------------------------------------------------------------------ Hi-Z

uint32_t mcull_hiz_level_count(uint32_t width, uint32_t height) { return level_count(width, height); }

uint64_t mcull_hiz_layout(uint32_t width, uint32_t height, uint32_t *levels, uint32_t lw[MCULL_MAX_HIZ_LEVELS],
                          uint32_t lh[MCULL_MAX_HIZ_LEVELS], uint64_t loff[MCULL_MAX_HIZ_LEVELS]) {
    uint32_t n = level_count(width, height);
    if (n > MCULL_MAX_HIZ_LEVELS) n = MCULL_MAX_HIZ_LEVELS;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t w = (width >> i) ? (width >> i) : 1u, h = (height >> i) ? (height >> i) : 1u;
        if (lw) lw[i] = w;
        if (lh) lh[i] = h;
        if (loff) loff[i] = total;
        total += (uint64_t)w * h;
    }
    if (levels) *levels = n;
    return total;
}

namespace {

// buffers and level table of view slot `view` for a width x height image (everything of a build but its launches)
int hiz_prepare(mcull_ctx *c, uint32_t view, uint32_t width, uint32_t height) {
    HizSlot &s = c->hiz[view];
    uint32_t levels;
    const uint64_t total = mcull_hiz_layout(width, height, &levels, s.lw, s.lh, s.loff);
    if (total >= 0xffffffffull) return fail(c, MCULL_E_INVALID, "pyramid exceeds 2^32 texels");
    int rc;
    if ((rc = ensure(c, s.pyramid, total * 4))) return rc;
    if (!s.table.p) {  // level table + the slot's "a depth image held a NaN" word, which only the pyramid kernels ever set
        if ((rc = ensure(c, s.table, mcull::kHizTableWords * 4))) return rc;
        HIPCHK(c, hipMemsetAsync(s.table.p, 0, mcull::kHizTableWords * 4, c->stream));
    }
    if (!c->hiz_counter.p) {  // ticket words, one set per view of a batched build: zero once; the kernels' last workgroups re-zero them
        if ((rc = ensure(c, c->hiz_counter, MCULL_MAX_VIEWS * mcull::kHizCounterBytes))) return rc;
        HIPCHK(c, hipMemsetAsync(c->hiz_counter.p, 0, MCULL_MAX_VIEWS * mcull::kHizCounterBytes, c->stream));
    }
    if (!s.built || s.w != width || s.h != height) {
        uint32_t tbl[48];
        std::memset(tbl, 0, sizeof tbl);
        for (uint32_t i = 0; i < levels; i++) { tbl[i] = s.lw[i]; tbl[16 + i] = s.lh[i]; tbl[32 + i] = (uint32_t)s.loff[i]; }
        if ((rc = copy_in(c, s.table.p, tbl, sizeof tbl, MCULL_MEM_HOST))) return rc;
    }
    s.has_strips = c->hiz_strips && mcull::hiz_writes_strips(width, height, c->hiz_mode);
    if (s.has_strips && (rc = ensure(c, s.strips, mcull::hiz_strip_floats(width, height) * 4))) return rc;
    s.w = width; s.h = height; s.levels = levels; s.total_floats = total;
    return MCULL_OK;
}

void hiz_built(HizSlot &s) {
    s.built = true;
    s.windows_current = false;
    s.culls_since_build = 0;
}

}  // namespace

int mcull_hiz_build(mcull_ctx *c, uint32_t view, const float *depth, uint32_t width, uint32_t height, int mem) {
    if (!c) return MCULL_E_INVALID;
    if (view >= MCULL_MAX_VIEWS) return fail(c, MCULL_E_INVALID, "view slot %u out of range", view);
    if (!depth || width == 0 || height == 0) return fail(c, MCULL_E_INVALID, "empty depth image");
    if (!mem_kind_ok(mem)) return fail(c, MCULL_E_INVALID, "mem must be MCULL_MEM_HOST or MCULL_MEM_DEVICE, not %d", mem);
    if (level_count(width, height) > MCULL_MAX_HIZ_LEVELS) return fail(c, MCULL_E_INVALID, "depth image too large");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = hiz_prepare(c, view, width, height))) return rc;
    HizSlot &s = c->hiz[view];
    const float *src = depth;
    if (mem == MCULL_MEM_HOST) {
        if ((rc = ensure(c, c->depth_staging, (size_t)width * height * 4))) return rc;
        if ((rc = copy_in(c, c->depth_staging.p, depth, (size_t)width * height * 4, MCULL_MEM_HOST))) return rc;
        src = (const float *)c->depth_staging.p;
    }
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    HIPCHK(c, mcull::launch_hiz(src, width, height, (float *)s.pyramid.p, s.levels, (const uint32_t *)s.table.p,
                                (uint32_t *)c->hiz_counter.p, (uint32_t *)s.table.p + mcull::kHizNanWord, (float *)s.strips.p,
                                c->hiz_mode, c->stream));
    if (c->timing) {
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        c->timed = true;
    }
    hiz_built(s);
    return MCULL_OK;
}

int mcull_hiz_build_views(mcull_ctx *c, uint32_t n_views, const uint32_t *views, const float *const *depth, uint32_t width,
                          uint32_t height, int mem) {
    if (!c) return MCULL_E_INVALID;
    if (n_views == 0 || n_views > MCULL_MAX_VIEWS) return fail(c, MCULL_E_INVALID, "n_views must be 1..%d", MCULL_MAX_VIEWS);
    if (!views || !depth || width == 0 || height == 0) return fail(c, MCULL_E_INVALID, "empty depth images");
    if (!mem_kind_ok(mem)) return fail(c, MCULL_E_INVALID, "mem must be MCULL_MEM_HOST or MCULL_MEM_DEVICE, not %d", mem);
    if (level_count(width, height) > MCULL_MAX_HIZ_LEVELS) return fail(c, MCULL_E_INVALID, "depth image too large");
    uint32_t seen = 0;
    for (uint32_t i = 0; i < n_views; i++) {
        if (views[i] >= MCULL_MAX_VIEWS) return fail(c, MCULL_E_INVALID, "view slot %u out of range", views[i]);
        if (!depth[i]) return fail(c, MCULL_E_INVALID, "depth image %u is null", i);
        if (seen & (1u << views[i])) return fail(c, MCULL_E_INVALID, "view slot %u named twice", views[i]);
        seen |= 1u << views[i];
    }
    // images in host memory go through one staging buffer, and small images use schemes with launches of their own: one by one
    if (mem == MCULL_MEM_HOST || n_views == 1 || !mcull::hiz_views_batchable(width, height, level_count(width, height), c->hiz_mode)) {
        for (uint32_t i = 0; i < n_views; i++)
            if (int rc = mcull_hiz_build(c, views[i], depth[i], width, height, mem)) return rc;
        return MCULL_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    float *pyramid[MCULL_MAX_VIEWS], *strips[MCULL_MAX_VIEWS];
    uint32_t *table[MCULL_MAX_VIEWS], *counter[MCULL_MAX_VIEWS];
    for (uint32_t i = 0; i < n_views; i++) {
        if (int rc = hiz_prepare(c, views[i], width, height)) return rc;
        HizSlot &s = c->hiz[views[i]];
        pyramid[i] = (float *)s.pyramid.p;
        strips[i] = (float *)s.strips.p;
        table[i] = (uint32_t *)s.table.p;
        counter[i] = (uint32_t *)c->hiz_counter.p + (size_t)i * (mcull::kHizCounterBytes / 4);
    }
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    HIPCHK(c, mcull::launch_hiz_views(n_views, depth, width, height, pyramid, c->hiz[views[0]].levels, table, counter, strips, c->stream));
    if (c->timing) {
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        c->timed = true;
    }
    for (uint32_t i = 0; i < n_views; i++) hiz_built(c->hiz[views[i]]);
    return MCULL_OK;
}

int mcull_hiz_pyramid(mcull_ctx *c, uint32_t view, const float **device_ptr, uint64_t *float_count) {
    if (!c || view >= MCULL_MAX_VIEWS) return MCULL_E_INVALID;
    if (!c->hiz[view].built) return fail(c, MCULL_E_STATE, "no pyramid built for view %u", view);
    if (device_ptr) *device_ptr = (const float *)c->hiz[view].pyramid.p;
    if (float_count) *float_count = c->hiz[view].total_floats;
    return MCULL_OK;
}

int mcull_hiz_download_level(mcull_ctx *c, uint32_t view, uint32_t level, float *out) {
    if (!c || !out || view >= MCULL_MAX_VIEWS) return MCULL_E_INVALID;
    HizSlot &s = c->hiz[view];
    if (!s.built) return fail(c, MCULL_E_STATE, "no pyramid built for view %u", view);
    if (level >= s.levels) return fail(c, MCULL_E_INVALID, "level %u of %u", level, s.levels);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, (const float *)s.pyramid.p + s.loff[level], (size_t)s.lw[level] * s.lh[level] * 4,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCULL_OK;
}

// ------------------------------------------------------------------------------------ cull

namespace {

// Everything mcull_cull decides before it launches: validated parameters, the scene and view
// constants, and the output buffers of one view slot (allocated here).
constexpr uint32_t kPrecullCooldown = 15;  // culls without the pre-cull after one that kept more than 90 % of the units

struct CullPlan {
    mcull_cull_params p;
    mcull::SceneDev s;
    mcull::ViewConst v;
    uint32_t uniform_count;
    uint64_t n_pairs;
    uint64_t *total;
    mcull_visible *list;
    uint64_t capacity;
    void *commands;
    mcull::CmdLayout lay;
    bool want_mask, two_level, streaming;
    uint32_t n_tiles;
    uint32_t *work;             // work list of this cull's instance pre-cull, nullptr when it does not run
    uint64_t precull_units;     // groups (ranged scenes) or instances it decides about
    bool hand_on;               // the list's entries carry what the instance level has settled (cull_mask_kernel<..., FLAGS>)
    uint32_t *group_work;       // list of the group level behind it, nullptr when that does not run
    uint64_t list_estimate;     // groups the work list is expected to hold (0 = unknown), see launch_cull_mask
    bool reports;               // this cull's passes leave their counts in host-mapped memory (it armed the slot's probe)
};

// The half of the plan that allocates nothing: validated parameters, scene and view constants, sizes.
// g == nullptr: the Globals are in device memory; the view-dependent constants stay zero here (a kernel
// derives them) and meshlet_count stands in for globals.meshletCount.
int plan_inputs(mcull_ctx *c, const mcull_globals *g, uint32_t meshlet_count, const mcull_cull_params *params, CullPlan *plan) {
    if (!c->have_bounds) return fail(c, MCULL_E_STATE, "no meshlet bounds uploaded");
    mcull_cull_params &p = plan->p;
    std::memset(&p, 0, sizeof p);
    if (params) p = *params;
    else { p.tests = MCULL_TEST_ALL; p.compaction = MCULL_COMPACT_ORDERED; }
    const uint32_t record = p.cmd_form == MCULL_CMD_EXT_GROUPS ? 12u : 8u;
    if (p.cmd_stride == 0) p.cmd_stride = record;
    if (p.tests & ~MCULL_TEST_ALL) return fail(c, MCULL_E_INVALID, "unknown test bits 0x%x", p.tests);
    if (p.compaction > MCULL_COMPACT_BLOCK_ATOMIC) return fail(c, MCULL_E_INVALID, "unknown compaction %u", p.compaction);
    if (p.cmd_form > MCULL_CMD_NV_TASK_GROUPS) return fail(c, MCULL_E_INVALID, "unknown command form %u", p.cmd_form);
    if (p.flags & ~(uint32_t)(MCULL_CULL_NO_INSTANCE_PRECULL | MCULL_CULL_EXACT_ARITHMETIC | MCULL_CULL_COUNT_SLOW_PAIRS))
        return fail(c, MCULL_E_INVALID, "unknown flag bits 0x%x", p.flags);
    if ((p.cmd_stride & 3u) || (p.cmd_offset & 3u) || p.cmd_offset + record > p.cmd_stride)
        return fail(c, MCULL_E_INVALID, "command stride/offset must be 4-byte multiples with offset+%u <= stride", record);
    if ((p.tests & MCULL_TEST_HIZ) && (p.hiz_view >= MCULL_MAX_VIEWS || !c->hiz[p.hiz_view].built))
        return fail(c, MCULL_E_STATE, "Hi-Z test requested but no pyramid built for view %u", p.hiz_view);

    mcull::SceneDev &s = plan->s;
    s = mcull::SceneDev{};
    s.cone_axis_w = (const float4 *)c->cone.p;
    s.centre_radius = (const float4 *)c->sphere.p;
    s.transforms = (const float4 *)c->transforms.p;
    s.xf_scale = (const float *)c->xf_scale.p;
    s.n_instances = (uint32_t)c->n_instances;
    plan->uniform_count = 0;
    if (c->have_ranges) {
        if (c->ranged_instances != c->n_instances)
            return fail(c, MCULL_E_STATE, "instance ranges cover %llu instances, transforms %llu",
                        (unsigned long long)c->ranged_instances, (unsigned long long)c->n_instances);
        if (c->ranged_max_end > c->n_meshlets) return fail(c, MCULL_E_INVALID, "an instance range exceeds the meshlet array");
        s.ranges = (const uint2 *)c->ranges.p;
        s.next_nonempty = (const uint32_t *)c->next_nonempty.p;
        s.group_base = (const uint32_t *)c->group_base.p;
        s.group_rec = (const uint4 *)c->group_rec.p;
        s.groups_per_instance = 0;
        s.n_groups = c->ranged_groups;
        plan->n_pairs = c->ranged_pairs;
    } else {
        plan->uniform_count = g ? g->meshletCount : meshlet_count;  // shader.task:85
        if (plan->uniform_count > c->n_meshlets) return fail(c, MCULL_E_INVALID, "globals.meshletCount exceeds uploaded meshlets");
        s.groups_per_instance = (plan->uniform_count + 63u) / 64u;
        s.n_groups = (uint64_t)s.groups_per_instance * c->n_instances;
        plan->n_pairs = (uint64_t)plan->uniform_count * c->n_instances;
        if (s.n_groups >= 0xffffffffull) return fail(c, MCULL_E_INVALID, "more than 2^32 meshlet groups");
    }

    mcull::ViewConst &v = plan->v;
    v = mcull::ViewConst{};
    if (g) {
        std::memcpy(v.frustum, g->frustum, sizeof v.frustum);
        mat4_mul(g->proj, g->view, v.vp);
        camera_position(g->view, v.cam);
        v.proj11 = g->proj[5];
        mcull::view_derive_bounds(v);
    }
    v.tests = p.tests;
    v.uniform_count = plan->uniform_count;
    v.force_exact = (p.flags & MCULL_CULL_EXACT_ARITHMETIC) ? 1u : 0u;
    if (p.tests & MCULL_TEST_HIZ) {
        const HizSlot &hs = c->hiz[p.hiz_view];
        v.tex_max = (float)(hs.w > hs.h ? hs.w : hs.h);
        v.w0 = hs.w;
        v.h0 = hs.h;
        v.levels = hs.levels;
        v.level_table = (const uint32_t *)hs.table.p;
        v.pyramid = (const float *)hs.pyramid.p;
        v.strips = hs.has_strips ? (const float *)hs.strips.p : nullptr;
        v.strips_w = (hs.w + 63u) / 64u;
    }
    plan->want_mask = p.compaction == MCULL_COMPACT_ORDERED || p.write_mask;
    plan->work = nullptr;
    plan->precull_units = 0;
    plan->hand_on = false;
    plan->reports = false;
    plan->group_work = nullptr;
    plan->list_estimate = 0;
    plan->n_tiles = mcull::tile_count_for(s.n_groups);
    plan->lay = mcull::CmdLayout{p.cmd_stride, p.cmd_offset, p.cmd_form};
    plan->two_level = mcull::needs_tile_offsets(s.n_groups);
    // bounds that cannot stay cache-resident between culls are read with the streaming hint
    plan->streaming = c->force_streaming || c->n_meshlets * 32ull > (64ull << 20);
    // the strip maxima save memory traffic and cost arithmetic: for the passes that are bound by memory only
    // (1 M instances of one cache-resident mesh: 12.2 ms without, 12.9 ms with them)
    v.strips_per_pair = (plan->streaming && v.strips) ? 1u : 0u;
    return MCULL_OK;
}

// The enclosing spheres of the meshlet ranges (one per instance, or one for the uniform case), recomputed after the
// bounds, the ranges or the uniform meshlet count changed.
int ensure_range_spheres(mcull_ctx *c, const mcull::SceneDev &s, uint32_t uniform_count) {
    const bool ranged = c->have_ranges;
    if (c->spheres_valid && c->spheres_ranged == ranged && (ranged || c->spheres_uniform_count == uniform_count)) return MCULL_OK;
    const uint32_t n_ranges = ranged ? (uint32_t)c->n_instances : 1u;
    if (int rc = ensure(c, c->range_spheres, mcull::range_sphere_bytes(n_ranges))) return rc;
    HIPCHK(c, mcull::launch_range_spheres(s.centre_radius, s.ranges, n_ranges, uniform_count, ranged ? c->ranged_longest : uniform_count,
                                          (float4 *)c->range_spheres.p, c->stream));
    c->spheres_valid = true;
    c->spheres_ranged = ranged;
    c->spheres_uniform_count = uniform_count;
    return MCULL_OK;
}

// The other half: the output and scratch buffers of one view slot (allocated or grown here).
int plan_cull(mcull_ctx *c, const mcull_globals *g, uint32_t meshlet_count, const mcull_cull_params *params, mcull_ctx::ViewOut &o,
              CullPlan *plan, bool multi_view = false) {
    if (int rc0 = plan_inputs(c, g, meshlet_count, params, plan)) return rc0;
    mcull_cull_params &p = plan->p;
    mcull::SceneDev &s = plan->s;
    int rc;
    if (plan->want_mask && (rc = ensure(c, o.mask, s.n_groups * 8))) return rc;
    if ((rc = ensure(c, o.total, 256))) return rc;
    plan->total = p.total ? p.total : (uint64_t *)o.total.p;
    o.counts_slow_pairs = (p.flags & MCULL_CULL_COUNT_SLOW_PAIRS) != 0;
    if (o.counts_slow_pairs) {  // a spare word of the context-owned count buffer, zero at the start of the cull
        plan->v.slow_pairs = (unsigned long long *)o.total.p + 16;
        HIPCHK(c, mcull::launch_zero_words((uint32_t *)plan->v.slow_pairs, 2, c->stream));
    }
    plan->list = p.list;
    plan->capacity = p.list_capacity;
    if (!plan->list) {
        if (plan->capacity == 0) plan->capacity = plan->n_pairs;
        if ((rc = ensure(c, o.list, plan->capacity * sizeof(mcull_visible)))) return rc;
        plan->list = (mcull_visible *)o.list.p;
    }
    plan->commands = p.commands;
    if (!plan->commands) {
        const size_t bytes = (size_t)c->n_instances * p.cmd_stride;
        // bytes of a record outside the command itself are never written by the kernels: the
        // context-owned buffer is zeroed whenever it is (re)allocated or its record layout changes
        const uint64_t layout_key = ((uint64_t)p.cmd_stride << 32) | ((uint64_t)p.cmd_form << 24) | p.cmd_offset;
        const bool fresh = !o.commands.p || o.commands.cap < bytes || o.commands_layout != layout_key;
        if ((rc = ensure(c, o.commands, bytes))) return rc;
        if (fresh) HIPCHK(c, mcull::launch_zero_words((uint32_t *)o.commands.p, o.commands.cap / 4, c->stream));
        o.commands_layout = layout_key;
        plan->commands = o.commands.p;
    }
    if (s.n_groups == 0 || p.compaction != MCULL_COMPACT_ORDERED) {
        if ((rc = ensure(c, o.inst_count, (c->n_instances + 1) * 4))) return rc;
    } else {
        bool precull = !multi_view && (p.tests & (MCULL_TEST_FRUSTUM | MCULL_TEST_HIZ)) && !(p.flags & MCULL_CULL_NO_INSTANCE_PRECULL) &&
                       s.n_groups >= c->precull_min_groups;
        // A cull that is being captured into a graph will be replayed for cameras and scenes this call knows nothing about:
        // it runs both levels whatever the back-off says, owns no probe, and sizes its grid with a wide margin.
        bool capturing = false;
        {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(c->stream, &cs) == hipSuccess) capturing = cs != hipStreamCaptureStatusNone;
        }
        const bool adaptive = c->precull_adaptive && !capturing;
        if (precull && adaptive) {
            const size_t slot = (size_t)(&o - c->out);
            if (o.probe_pending) {
                const uint32_t kept = c->kept_host[slot];
                if (kept != 0xffffffffu) {  // the cull that ran it has finished
                    o.probe_pending = false;
                    if ((uint64_t)kept * 10 > o.probe_units * 9) o.precull_cooldown[o.probe_tests & 7u] = kPrecullCooldown;
                    o.list_estimate[o.probe_tests & 7u] =
                        o.probe_group_level ? 0 : o.probe_in_groups ? (uint64_t)kept : (uint64_t)kept * o.probe_groups_per_instance;
                    // the group level: what it kept (the mask pass's list) against what it was handed (kept_host[2])
                    const uint32_t handed = c->kept_host[2];
                    if (o.probe_group_level && handed != 0xffffffffu && (uint64_t)kept * 10 > (uint64_t)handed * 9)
                        o.group_cooldown[o.probe_tests & 7u] = kPrecullCooldown;
                }
            }
            if (o.precull_cooldown[p.tests & 7u]) {
                o.precull_cooldown[p.tests & 7u]--;
                precull = false;
            }
        }
        if (precull) {
            // instance pre-cull: its range spheres (recomputed after the bounds or ranges changed) and its output bytes
            const bool ranged = c->have_ranges;
            if ((rc = ensure_range_spheres(c, s, plan->uniform_count))) return rc;
            plan->precull_units = ranged ? s.n_groups : c->n_instances;
            {
                const void *before = o.work.p;
                if ((rc = ensure(c, o.work, mcull::work_list_bytes(plan->precull_units, ranged)))) return rc;
                // a new buffer, or a cull whose enqueue failed half way: the header starts at zero; otherwise the last
                // cull's command pass has left word 0 zero
                if (o.work.p != before) o.work_clean = false;
                if (!o.work_clean) HIPCHK(c, mcull::launch_zero_words((uint32_t *)o.work.p, 2, c->stream));
                o.work_clean = false;  // until the command pass that re-zeroes it has been enqueued
            }
            plan->work = (uint32_t *)o.work.p;
            // The pre-pass also tells the mask pass which instances need no frustum / no occlusion test (two bits in the
            // work list's entries) where that arithmetic is what the pass waits for: cache-resident bounds (1 M instances of
            // one mesh: mask pass 12.2 -> 11.3 ms).  A pass over streamed bounds is bound by its bytes and keeps its constant
            // tests (100 M pairs with the occlusion test: 0.278 ms without the flags, 0.285 ms with them).
            plan->hand_on = c->precull_flags && c->n_instances < (1ull << 30) && !plan->streaming;
            // instances of 32 groups and more: a group level thins the list once more
            const bool group_level = plan->hand_on && c->precull_groups && s.n_groups >= 32ull * c->n_instances && s.n_groups < (1ull << 30);
            if (group_level && adaptive && o.group_cooldown[p.tests & 7u]) {
                o.group_cooldown[p.tests & 7u]--;
            } else if (group_level) {
                if (!c->gspheres_valid) {
                    if ((rc = ensure(c, c->group_spheres, mcull::range_sphere_bytes((uint32_t)c->n_meshlets)))) return rc;
                    HIPCHK(c, mcull::launch_group_spheres(s.centre_radius, (uint32_t)c->n_meshlets, (float4 *)c->group_spheres.p, c->stream));
                    c->gspheres_valid = true;
                }
                if ((rc = ensure(c, o.group_work, mcull::work_list_bytes(s.n_groups, ranged)))) return rc;
                plan->group_work = (uint32_t *)o.group_work.p;
            }
            if (p.tests & MCULL_TEST_HIZ) {
            // The occlusion proofs of the two levels get the window tables of the pyramid, made here when the pyramid has
            // been rebuilt since (7 small launches, about 20 us at 3840 x 2160): always with a group level, whose scenes gain
            // milliseconds; otherwise from the second cull on that uses the same build - a frame loop that rebuilds the
            // pyramid for every cull never pays for tables it would use once (100 M pairs: 0.225 -> 0.210 ms per cull with
            // them, which one build per cull would eat).
            HizSlot &hs = c->hiz[p.hiz_view];
            const bool wanted = c->hiz_windows && (plan->group_work || hs.culls_since_build >= 1u);
            if (hs.culls_since_build < 0xffffffffu) hs.culls_since_build++;
            const uint32_t win_levels = wanted ? mcull::hiz_window_levels(hs.w, hs.h) : 0u;
            if (win_levels && hs.levels > mcull::kHizWinBase) {
                if ((rc = ensure(c, hs.windows, mcull::hiz_window_floats(hs.w, hs.h) * 4))) return rc;
                if (!hs.windows_current) {
                    HIPCHK(c, mcull::launch_hiz_windows((const float *)hs.pyramid.p, (uint32_t *)hs.table.p, hs.w, hs.h,
                                                        hs.loff[mcull::kHizWinBase], (float *)hs.windows.p, c->stream));
                    hs.windows_current = true;
                }
                plan->v.windows = (const float *)hs.windows.p;
                plan->v.win_levels = win_levels;
            }
            }
            // One probe in flight per slot.  Culls with different tests take turns at it: the one that owned the last probe
            // leaves the next to a cull with other tests that has passed by meanwhile (at most four times in a row, should
            // those not come back).
            const uint32_t my_tests = p.tests & 7u;
            if (o.probe_pending && my_tests != o.probe_tests) o.other_tests_waiting = true;
            bool my_turn = true;
            if (!o.probe_pending && my_tests == o.probe_tests && o.other_tests_waiting && o.probe_deferred < 4u) {
                o.probe_deferred++;
                my_turn = false;
            }
            if (adaptive && !o.probe_pending && my_turn) {
                o.probe_tests = my_tests;
                o.other_tests_waiting = false;
                o.probe_deferred = 0;
                c->kept_host[(size_t)(&o - c->out)] = 0xffffffffu;  // until this cull's mask pass reports
                if (plan->group_work) c->kept_host[2] = 0xffffffffu;       // and its group level (single-view culls are slot 0)
                o.probe_pending = true;
                plan->reports = true;  // only this cull writes the words: an earlier one still in flight cannot overwrite the sentinel
                o.probe_units = plan->group_work ? s.n_groups : plan->precull_units;
                o.probe_group_level = plan->group_work != nullptr;
                o.probe_in_groups = ranged;
                o.probe_groups_per_instance = s.groups_per_instance;
            }
            // the walking form that takes what lies beyond the estimate reads two flag bits off the top of an entry
            if (!plan->hand_on && (ranged || c->n_instances < (1ull << 30))) {
                // 3 % and 1 024 groups of margin on what was reported; not worth a second launch unless it spares 4 096 groups' workgroups
                const uint64_t known = o.list_estimate[p.tests & 7u];
                const uint64_t padded = capturing ? known + known / 2 + 4096 : known + known / 32 + 1024;
                plan->list_estimate = c->precull_estimate >= 0 ? (uint64_t)c->precull_estimate
                                      : (c->precull_adaptive && known && padded + 4096 < s.n_groups) ? padded : 0;
            }
        }
        if (!o.tile_counts.p || o.tile_counts.cap < (size_t)plan->n_tiles * 4) o.counters_clean = false;
        if ((rc = ensure(c, o.tile_counts, (size_t)plan->n_tiles * 4))) return rc;
        if (plan->two_level && (rc = ensure(c, o.tile_offsets, (size_t)plan->n_tiles * 8))) return rc;
        if ((rc = ensure(c, o.inst_first, (c->n_instances + 1) * 4))) return rc;
        if (!o.counters_clean) HIPCHK(c, mcull::launch_zero_words((uint32_t *)o.tile_counts.p, o.tile_counts.cap / 4, c->stream));
        o.counters_clean = false;  // until the command kernel that re-zeroes them has been enqueued
    }
    return MCULL_OK;
}

// scatter + commands of the ordered path (after the mask pass has filled mask and tile counters)
int launch_ordered_tail(mcull_ctx *c, mcull_ctx::ViewOut &o, const CullPlan &pl) {
    // a work list expected to hold less than a quarter of the scene's groups: the scatter pass's short rounds
    const bool short_rounds = c->compact_short >= 0 ? c->compact_short != 0
                                                    : (pl.work && pl.list_estimate && pl.list_estimate * 4ull < pl.s.n_groups);
    HIPCHK(c, mcull::launch_compact(pl.s, (const uint64_t *)o.mask.p, (const uint32_t *)o.tile_counts.p,
                                    pl.two_level ? (uint64_t *)o.tile_offsets.p : nullptr, pl.list, pl.capacity,
                                    (uint32_t *)o.inst_first.p, pl.total, pl.n_pairs * 8ull > (64ull << 20), c->stream, short_rounds));
    HIPCHK(c, mcull::launch_emit_commands(pl.s, pl.uniform_count, (const uint32_t *)o.inst_first.p, nullptr,
                                          (const uint64_t *)pl.total, nullptr, 0, pl.lay, pl.commands,
                                          (uint32_t *)o.tile_counts.p, c->stream,
                                          pl.two_level ? (const uint64_t *)o.tile_offsets.p : nullptr, (uint32_t *)o.inst_first.p, pl.work));
    o.counters_clean = true;
    if (pl.work) o.work_clean = true;
    return MCULL_OK;
}

void finish_cull(mcull_ctx::ViewOut &o, const CullPlan &pl, mcull_cull_result *result) {
    o.last.valid = true;
    o.total_ptr = pl.total;
    o.last.scene = pl.s;
    o.last.uniform_count = pl.uniform_count;
    o.last.compaction = pl.p.compaction;
    o.last.lay = pl.lay;
    o.last.list = pl.list;
    o.last.capacity = pl.capacity;
    o.last.commands = pl.commands;
    o.last.have_mask = pl.want_mask;  // a scene without any group has a mask of zero words
    o.last.n_pairs = pl.n_pairs;
    o.last.work = pl.work;
    o.last.precull_units = pl.precull_units;
    o.last.group_work = pl.group_work;
    if (result) {
        result->list = pl.list;
        result->commands = pl.commands;
        result->total = (const uint64_t *)pl.total;
        result->mask = (o.last.have_mask && pl.s.n_groups > 0) ? (const uint64_t *)o.mask.p : nullptr;
        result->n_groups = pl.s.n_groups;
        result->n_pairs = pl.n_pairs;
        result->first_task =
            (pl.p.compaction == MCULL_COMPACT_ORDERED && pl.s.n_groups > 0) ? (const uint32_t *)o.inst_first.p : nullptr;
    }
}

}  // namespace

namespace {

// g_dev != nullptr: the Globals are read on the device when the stream gets there (ordered compaction only)
int cull_impl(mcull_ctx *c, const mcull_globals *g, const mcull_globals *g_dev, uint32_t meshlet_count,
              const mcull_cull_params *params, mcull_cull_result *result) {
    if (!c) return MCULL_E_INVALID;
    if (!g && !g_dev) return fail(c, MCULL_E_INVALID, "globals is null");
    HIPCHK(c, hipSetDevice(c->device));
    mcull_ctx::ViewOut &o = c->out[0];
    CullPlan pl;
    if (int rc = plan_cull(c, g, meshlet_count, params, o, &pl)) return rc;
    const mcull::ViewConst *view_dev = nullptr;
    if (g_dev) {
        if (pl.p.compaction != MCULL_COMPACT_ORDERED)
            return fail(c, MCULL_E_INVALID, "device-resident globals support ordered compaction only");
        if (int rc = ensure(c, c->view_dev, kViewDevBytes)) return rc;
        view_dev = (const mcull::ViewConst *)c->view_dev.p;
    }
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (view_dev) {
        hipLaunchKernelGGL(derive_view_kernel, dim3(1), dim3(64), 0, c->stream, g_dev, pl.v, (mcull::ViewConst *)c->view_dev.p);
        HIPCHK(c, hipGetLastError());
    }
    if (pl.s.n_groups == 0) {
        HIPCHK(c, mcull::launch_zero_words((uint32_t *)pl.total, 2, c->stream));
        HIPCHK(c, mcull::launch_zero_words((uint32_t *)o.inst_count.p, c->n_instances + 1, c->stream));
        HIPCHK(c, mcull::launch_emit_commands(pl.s, pl.uniform_count, nullptr, (const uint32_t *)o.inst_count.p,
                                              (const uint64_t *)pl.total, nullptr, 0, pl.lay, pl.commands, nullptr, c->stream));
    } else if (pl.p.compaction == MCULL_COMPACT_ORDERED) {
        const bool hand_on = pl.work && pl.hand_on;
        if (pl.work) {
            // the mask pass will cover the work list's groups only: every other ballot is zero from here
            // (the list's length word is zero: the last cull's command pass, or the allocation, left it so)
            // few groups per instance: the pre-pass zeroes the dropped groups' ballots itself; many: one memset
            const bool inline_zero = pl.s.n_groups <= 8ull * pl.s.n_instances;
            if (!inline_zero) HIPCHK(c, mcull::launch_zero_words((uint32_t *)o.mask.p, pl.s.n_groups * 2, c->stream));
            HIPCHK(c, mcull::launch_instance_precull(pl.s, (const float4 *)c->range_spheres.p, c->have_ranges,
                                                     c->have_ranges ? (const uint32_t *)c->group_base.p : nullptr, pl.v, view_dev,
                                                     pl.work, inline_zero ? (uint64_t *)o.mask.p : nullptr, hand_on, c->stream,
                                                     pl.group_work));
            if (pl.group_work)
                HIPCHK(c, mcull::launch_group_precull(pl.s, (const float4 *)c->group_spheres.p, (uint32_t)c->n_meshlets, pl.v, view_dev,
                                                      pl.work, pl.group_work, c->stream, pl.reports ? c->kept_dev + 2 : nullptr));
        }
        HIPCHK(c, mcull::launch_cull_mask(pl.s, pl.v, (uint64_t *)o.mask.p, (uint32_t *)o.tile_counts.p, pl.streaming,
                                          c->stream, view_dev, pl.group_work ? pl.group_work : pl.work,
                                          pl.reports ? c->kept_dev : nullptr, hand_on, pl.group_work != nullptr,
                                          pl.work && !hand_on ? pl.list_estimate : 0));
        if (int rc = launch_ordered_tail(c, o, pl)) return rc;
    } else {
        HIPCHK(c, mcull::launch_zero_words((uint32_t *)pl.total, 2, c->stream));
        HIPCHK(c, mcull::launch_zero_words((uint32_t *)o.inst_count.p, c->n_instances + 1, c->stream));
        HIPCHK(c, mcull::launch_cull_atomic(pl.s, pl.v, pl.list, pl.capacity, (uint32_t *)o.inst_count.p, pl.total,
                                            pl.p.write_mask ? (uint64_t *)o.mask.p : nullptr,
                                            pl.p.compaction == MCULL_COMPACT_BLOCK_ATOMIC, c->stream));
        HIPCHK(c, mcull::launch_emit_commands(pl.s, pl.uniform_count, nullptr, (const uint32_t *)o.inst_count.p,
                                              (const uint64_t *)pl.total, nullptr, 0, pl.lay, pl.commands, nullptr, c->stream));
    }
    if (c->timing) {
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        c->timed = true;
    }
    finish_cull(o, pl, result);
    c->cur = 0;
    return MCULL_OK;
}

}  // namespace

int mcull_cull(mcull_ctx *c, const mcull_globals *g, const mcull_cull_params *params, mcull_cull_result *result) {
    if (c && !g) return fail(c, MCULL_E_INVALID, "globals is null");
    return cull_impl(c, g, nullptr, 0, params, result);
}

int mcull_cull_device_globals(mcull_ctx *c, const mcull_globals *device_globals, uint32_t meshlet_count,
                              const mcull_cull_params *params, mcull_cull_result *result) {
    if (c && !device_globals) return fail(c, MCULL_E_INVALID, "device_globals is null");
    return cull_impl(c, nullptr, device_globals, meshlet_count, params, result);
}

// mcull_cull without any allocation: outputs are the caller's, scratch is this device's static array.
int mcull_cull_static(mcull_ctx *c, const mcull_globals *g, const mcull_cull_params *params, mcull_cull_result *result) {
    if (!c) return MCULL_E_INVALID;
    if (!g) return fail(c, MCULL_E_INVALID, "globals is null");
    if (!params || !params->list || !params->commands || !params->total)
        return fail(c, MCULL_E_INVALID, "mcull_cull_static needs caller-owned list, commands and total");
    if (params->compaction != MCULL_COMPACT_ORDERED)
        return fail(c, MCULL_E_INVALID, "mcull_cull_static supports ordered compaction only");
    if (params->write_mask || (params->flags & MCULL_CULL_COUNT_SLOW_PAIRS))
        return fail(c, MCULL_E_INVALID, "mcull_cull_static supports neither write_mask nor MCULL_CULL_COUNT_SLOW_PAIRS");
    HIPCHK(c, hipSetDevice(c->device));
    CullPlan pl;
    if (int rc = plan_inputs(c, g, 0, params, &pl)) return rc;
    if (pl.s.n_groups > MCULL_STATIC_MAX_GROUPS)
        return fail(c, MCULL_E_CAPACITY, "mcull_cull_static: %llu meshlet groups, the static scratch holds %llu",
                    (unsigned long long)pl.s.n_groups, (unsigned long long)MCULL_STATIC_MAX_GROUPS);
    if (c->n_instances > MCULL_STATIC_MAX_INSTANCES)
        return fail(c, MCULL_E_CAPACITY, "mcull_cull_static: %llu instances, the static scratch holds %llu",
                    (unsigned long long)c->n_instances, (unsigned long long)MCULL_STATIC_MAX_INSTANCES);
    unsigned char *ptr = c->static_scratch;  // resolved by mcull_create with this device current
    if (!ptr) return fail(c, MCULL_E_STATE, "the static scratch of device %d was never resolved", c->device);
    // One user of the array at a time: refuse when the last call on this GPU came from another stream and has not finished.
    ScratchGuard &guard = g_scratch_guard[c->device];
    std::lock_guard<std::mutex> hold(guard.lock);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    HIPCHK(c, hipStreamIsCapturing(c->stream, &capturing));
    if (capturing == hipStreamCaptureStatusNone && guard.recorded && guard.stream != c->stream && guard.done &&
        hipEventQuery(guard.done) == hipErrorNotReady)
        return fail(c, MCULL_E_STATE, "the static scratch of device %d is in use by a cull on another stream", c->device);
    uint64_t *mask = reinterpret_cast<uint64_t *>(ptr + kStaticMaskOff);
    uint32_t *tile_counts = reinterpret_cast<uint32_t *>(ptr + kStaticTileOff);
    uint64_t *tile_offsets = reinterpret_cast<uint64_t *>(ptr + kStaticOffsetsOff);
    uint32_t *inst_first = reinterpret_cast<uint32_t *>(ptr + kStaticFirstOff);
    pl.total = params->total;
    pl.list = params->list;
    pl.capacity = params->list_capacity;
    pl.commands = params->commands;
    if (pl.s.n_groups == 0) {
        // no pair at all: zero commands from zero per-instance counts (kept in the inst_first region)
        HIPCHK(c, mcull::launch_zero_words((uint32_t *)pl.total, 2, c->stream));
        HIPCHK(c, mcull::launch_zero_words(inst_first, c->n_instances + 1, c->stream));
        HIPCHK(c, mcull::launch_emit_commands(pl.s, pl.uniform_count, nullptr, inst_first, (const uint64_t *)pl.total, nullptr,
                                              0, pl.lay, pl.commands, nullptr, c->stream));
    } else {
        // the array keeps the previous call's contents (any context of this GPU may have made it): the
        // counters the mask pass adds into start at zero
        HIPCHK(c, mcull::launch_zero_words(tile_counts, pl.n_tiles, c->stream));
        HIPCHK(c, mcull::launch_cull_mask(pl.s, pl.v, mask, tile_counts, pl.streaming, c->stream, nullptr));
        HIPCHK(c, mcull::launch_compact(pl.s, mask, tile_counts, pl.two_level ? tile_offsets : nullptr, pl.list, pl.capacity,
                                        inst_first, pl.total, pl.n_pairs * 8ull > (64ull << 20), c->stream, c->compact_short == 1));
        HIPCHK(c, mcull::launch_emit_commands(pl.s, pl.uniform_count, inst_first, nullptr, (const uint64_t *)pl.total, nullptr, 0,
                                              pl.lay, pl.commands, tile_counts, c->stream, pl.two_level ? tile_offsets : nullptr,
                                              inst_first));
    }
    if (capturing == hipStreamCaptureStatusNone && guard.done) {
        HIPCHK(c, hipEventRecord(guard.done, c->stream));
        guard.recorded = true;
        guard.stream = c->stream;
    }
    if (result) {
        result->list = pl.list;
        result->commands = pl.commands;
        result->total = (const uint64_t *)pl.total;
        result->mask = pl.s.n_groups > 0 ? mask : nullptr;
        result->n_groups = pl.s.n_groups;
        result->n_pairs = pl.n_pairs;
        result->first_task = pl.s.n_groups > 0 ? inst_first : nullptr;
    }
    return MCULL_OK;
}

namespace {

// globals_dev != nullptr: n_views Globals in device memory, read when the stream gets there
int views_impl(mcull_ctx *c, uint32_t n_views, const mcull_globals *globals, const mcull_globals *globals_dev,
               uint32_t meshlet_count, const mcull_cull_params *params, mcull_cull_result *results) {
    if (!c) return MCULL_E_INVALID;
    if (n_views == 0 || n_views > MCULL_MAX_VIEWS) return fail(c, MCULL_E_INVALID, "n_views must be 1..%d", MCULL_MAX_VIEWS);
    if (!globals && !globals_dev) return fail(c, MCULL_E_INVALID, "globals is null");
    HIPCHK(c, hipSetDevice(c->device));
    CullPlan pl[MCULL_MAX_VIEWS];
    for (uint32_t i = 0; i < n_views; i++) {
        if (int rc = plan_cull(c, globals ? &globals[i] : nullptr, meshlet_count, params ? &params[i] : nullptr, c->out[i], &pl[i], true))
            return rc;
        if (pl[i].p.compaction != MCULL_COMPACT_ORDERED)
            return fail(c, MCULL_E_INVALID, "mcull_cull_views supports ordered compaction only (view %u)", i);
        if (pl[i].uniform_count != pl[0].uniform_count)
            return fail(c, MCULL_E_INVALID, "view %u: globals.meshletCount differs from view 0", i);
    }
    const mcull::ViewConst *view_dev = nullptr;
    if (globals_dev) {
        if (int rc = ensure(c, c->view_dev, kViewDevBytes)) return rc;
        view_dev = (const mcull::ViewConst *)c->view_dev.p;
    }
    mcull::ViewHot2 *hot_dev = view_dev ? (mcull::ViewHot2 *)((char *)c->view_dev.p + kViewDevHotOff) : nullptr;
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (view_dev) {
        for (uint32_t i = 0; i < n_views; i++) {
            hipLaunchKernelGGL(derive_view_kernel, dim3(1), dim3(64), 0, c->stream, globals_dev + i, pl[i].v,
                               (mcull::ViewConst *)c->view_dev.p + i);
            HIPCHK(c, hipGetLastError());
        }
        HIPCHK(c, mcull::launch_pack_view_pairs(view_dev, n_views, hot_dev, c->stream));
    }
    if (pl[0].s.n_groups == 0) {
        for (uint32_t i = 0; i < n_views; i++) {
            mcull_ctx::ViewOut &o = c->out[i];
            HIPCHK(c, mcull::launch_zero_words((uint32_t *)pl[i].total, 2, c->stream));
            HIPCHK(c, mcull::launch_zero_words((uint32_t *)o.inst_count.p, c->n_instances + 1, c->stream));
            HIPCHK(c, mcull::launch_emit_commands(pl[i].s, pl[i].uniform_count, nullptr, (const uint32_t *)o.inst_count.p,
                                                  (const uint64_t *)pl[i].total, nullptr, 0, pl[i].lay, pl[i].commands, nullptr,
                                                  c->stream));
        }
    } else {
        // one pass over the bounds for every view, then each view's scatter + commands
        const mcull::ViewConst *vc[MCULL_MAX_VIEWS];
        uint64_t *masks[MCULL_MAX_VIEWS];
        uint32_t *counts[MCULL_MAX_VIEWS];
        for (uint32_t i = 0; i < n_views; i++) {
            vc[i] = &pl[i].v;
            masks[i] = (uint64_t *)c->out[i].mask.p;
            counts[i] = (uint32_t *)c->out[i].tile_counts.p;
        }
        // instance level of the multi-view pass: one byte per instance and view (dropped / no frustum test / no occlusion
        // test); backs off like the single-view one when fewer than a tenth of the bytes say anything
        bool classify = c->precull_flags && pl[0].s.n_groups >= c->precull_min_groups;
        bool any_test = false;
        for (uint32_t i = 0; i < n_views; i++) {
            any_test |= (pl[i].p.tests & (MCULL_TEST_FRUSTUM | MCULL_TEST_HIZ)) != 0;
            if (pl[i].p.flags & MCULL_CULL_NO_INSTANCE_PRECULL) classify = false;
        }
        classify = classify && any_test;
        // (a captured call is replayed for cameras this call knows nothing about: no back-off, no probe - see plan_cull)
        hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
        const bool views_adaptive = c->precull_adaptive &&
                                    !(hipStreamIsCapturing(c->stream, &cap_status) == hipSuccess && cap_status != hipStreamCaptureStatusNone);
        if (classify && views_adaptive) {
            if (c->views_probe_pending) {
                const uint32_t useful = c->kept_host[1];
                if (useful != 0xffffffffu) {
                    c->views_probe_pending = false;
                    if ((uint64_t)useful * 10 < c->views_probe_units) c->views_cooldown = kPrecullCooldown;
                }
            }
            if (c->views_cooldown) {
                c->views_cooldown--;
                classify = false;
            }
        }
        uint32_t *useful_dev = nullptr, *useful_report = nullptr;
        if (classify) {
            if (int rc = ensure_range_spheres(c, pl[0].s, pl[0].uniform_count)) return rc;
            if (int rc = ensure(c, c->view_flags, (size_t)c->n_instances * MCULL_MAX_VIEWS)) return rc;
            useful_dev = (uint32_t *)c->out[0].total.p + 48;  // a spare word of the context-owned count buffer
            HIPCHK(c, mcull::launch_zero_words(useful_dev, 1, c->stream));
            HIPCHK(c, mcull::launch_classify_views(pl[0].s, (const float4 *)c->range_spheres.p, c->have_ranges, n_views, vc, view_dev,
                                                   (uint8_t *)c->view_flags.p, useful_dev, c->stream));
            if (views_adaptive && !c->views_probe_pending) {
                c->kept_host[1] = 0xffffffffu;  // until this cull's mask pass reports
                c->views_probe_pending = true;
                c->views_probe_units = c->n_instances * n_views;
                useful_report = c->kept_dev + 1;
            }
        }
        HIPCHK(c, mcull::launch_cull_mask_views(pl[0].s, n_views, vc, masks, counts, pl[0].streaming, c->stream, view_dev, hot_dev,
                                                classify ? (const uint8_t *)c->view_flags.p : nullptr, useful_dev, useful_report));
        if (n_views == 1) {
            if (int rc = launch_ordered_tail(c, c->out[0], pl[0])) return rc;
        } else {
            // every view's scatter and commands in three launches in all (8 views at 8192^2: 24 launches took 1.15 ms)
            mcull::ViewTail tails[MCULL_MAX_VIEWS];
            for (uint32_t i = 0; i < n_views; i++) {
                mcull_ctx::ViewOut &o = c->out[i];
                tails[i] = mcull::ViewTail{(const uint64_t *)o.mask.p, (uint32_t *)o.tile_counts.p, (uint64_t *)o.tile_offsets.p, pl[i].list,
                                           pl[i].capacity,           (uint32_t *)o.inst_first.p,  pl[i].total,                 pl[i].commands,
                                           pl[i].lay};
            }
            HIPCHK(c, mcull::launch_ordered_tails(pl[0].s, pl[0].uniform_count, n_views, tails, pl[0].n_pairs * 8ull > (64ull << 20), c->stream));
            for (uint32_t i = 0; i < n_views; i++) c->out[i].counters_clean = true;
        }
    }
    if (c->timing) {
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        c->timed = true;
    }
    for (uint32_t i = 0; i < n_views; i++) finish_cull(c->out[i], pl[i], results ? &results[i] : nullptr);
    c->cur = 0;
    return MCULL_OK;
}

}  // namespace

int mcull_cull_views(mcull_ctx *c, uint32_t n_views, const mcull_globals *globals, const mcull_cull_params *params,
                     mcull_cull_result *results) {
    if (c && !globals) return fail(c, MCULL_E_INVALID, "globals is null");
    return views_impl(c, n_views, globals, nullptr, 0, params, results);
}

int mcull_cull_views_device_globals(mcull_ctx *c, uint32_t n_views, const mcull_globals *device_globals, uint32_t meshlet_count,
                                    const mcull_cull_params *params, mcull_cull_result *results) {
    if (c && !device_globals) return fail(c, MCULL_E_INVALID, "device_globals is null");
    return views_impl(c, n_views, nullptr, device_globals, meshlet_count, params, results);
}

int mcull_select_result(mcull_ctx *c, uint32_t view) {
    if (!c) return MCULL_E_INVALID;
    if (view >= MCULL_MAX_VIEWS) return fail(c, MCULL_E_INVALID, "view %u out of range", view);
    if (!c->out[view].last.valid) return fail(c, MCULL_E_STATE, "view slot %u holds no cull result", view);
    c->cur = view;
    return MCULL_OK;
}

int mcull_rebase_commands(mcull_ctx *c, const uint64_t *device_totals, uint32_t rank) {
    if (!c) return MCULL_E_INVALID;
    mcull_ctx::ViewOut &o = c->out[c->cur];
    if (!o.last.valid) return fail(c, MCULL_E_STATE, "no cull result to rebase");
    if (!device_totals && rank) return fail(c, MCULL_E_INVALID, "device_totals is null");
    if (o.last.lay.form == MCULL_CMD_EXT_GROUPS)
        return fail(c, MCULL_E_INVALID, "EXT records carry no firstTask to rebase; offset first_task on the consumer side");
    // (NV_TASK_GROUPS records carry the same firstTask as NV_MESHLETS records and are rewritten the same way)
    HIPCHK(c, hipSetDevice(c->device));
    const bool ordered = o.last.compaction == MCULL_COMPACT_ORDERED && o.last.scene.n_groups > 0;
    HIPCHK(c, mcull::launch_emit_commands(o.last.scene, o.last.uniform_count,
                                          ordered ? (const uint32_t *)o.inst_first.p : nullptr,
                                          ordered ? nullptr : (const uint32_t *)o.inst_count.p,
                                          (const uint64_t *)o.total_ptr, device_totals, rank, o.last.lay,
                                          o.last.commands, nullptr, c->stream));
    return MCULL_OK;
}

int mcull_rebase_gathered_commands(void *stream, void *gathered, uint32_t world, uint64_t rows_per_rank, uint64_t *totals,
                                   uint64_t *bases) {
    if (!gathered || !totals || !bases || world == 0 || world > 4096) return MCULL_E_INVALID;
    if (mcull::launch_rebase_gathered(gathered, world, rows_per_rank, totals, bases, (hipStream_t)stream) != hipSuccess)
        return MCULL_E_HIP;
    return MCULL_OK;
}

uint64_t mcull_task_payload_count(const mcull_ctx *c) {
    if (!c) return 0;
    const mcull_ctx::ViewOut &o = c->out[c->cur];
    if (!o.last.valid || o.last.scene.ranges) return 0;
    return (uint64_t)((o.last.uniform_count + 31u) / 32u) * o.last.scene.n_instances;
}

int mcull_emit_task_payloads(mcull_ctx *c, mcull_task_payload *out) {
    if (!c || !out) return MCULL_E_INVALID;
    mcull_ctx::ViewOut &o = c->out[c->cur];
    if (!o.last.valid || !o.last.have_mask) return fail(c, MCULL_E_STATE, "no cull result with a visibility mask");
    if (o.last.scene.ranges) return fail(c, MCULL_E_INVALID, "task payloads need a uniform meshlet count");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t ppi = (o.last.uniform_count + 31u) / 32u;
    HIPCHK(c, mcull::launch_task_payloads(o.last.scene, (const uint64_t *)o.mask.p, out,
                                          (uint64_t)ppi * o.last.scene.n_instances, ppi, c->stream));
    return MCULL_OK;
}

int mcull_download_count(mcull_ctx *c, uint64_t *total) {
    if (!c || !total) return MCULL_E_INVALID;
    mcull_ctx::ViewOut &o = c->out[c->cur];
    if (!o.last.valid) return fail(c, MCULL_E_STATE, "no cull result");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(total, o.total_ptr, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // firstTask and the per-instance list offsets are 32-bit fields (VkDrawMeshTasksIndirectCommandNV): an
    // ordered cull that keeps 2^32 pairs or more has wrapped them, whatever the list capacity was
    if (o.last.compaction == MCULL_COMPACT_ORDERED && *total > 0xffffffffull)
        return fail(c, MCULL_E_CAPACITY, "%llu visible pairs: the 32-bit firstTask offsets of the commands have wrapped",
                    (unsigned long long)*total);
    return MCULL_OK;
}

int mcull_download_list(mcull_ctx *c, mcull_visible *out, uint64_t capacity, uint64_t *total) {
    if (!c) return MCULL_E_INVALID;
    mcull_ctx::ViewOut &o = c->out[c->cur];
    uint64_t n = 0;
    int rc = mcull_download_count(c, &n);
    if (rc) return rc;
    if (total) *total = n;
    if (n > o.last.capacity)
        return fail(c, MCULL_E_CAPACITY, "visible list needs %llu entries, device capacity is %llu",
                    (unsigned long long)n, (unsigned long long)o.last.capacity);
    if (n > capacity)
        return fail(c, MCULL_E_CAPACITY, "visible list needs %llu entries, host capacity is %llu",
                    (unsigned long long)n, (unsigned long long)capacity);
    if (n && !out) return fail(c, MCULL_E_INVALID, "out is null");
    if (n) {
        HIPCHK(c, hipMemcpyAsync(out, o.last.list, n * sizeof(mcull_visible), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MCULL_OK;
}

int mcull_download_commands(mcull_ctx *c, void *out, uint64_t bytes) {
    if (!c || !out) return MCULL_E_INVALID;
    mcull_ctx::ViewOut &o = c->out[c->cur];
    if (!o.last.valid) return fail(c, MCULL_E_STATE, "no cull result");
    const uint64_t have = (uint64_t)o.last.scene.n_instances * o.last.lay.stride;
    if (bytes > have) return fail(c, MCULL_E_INVALID, "requested %llu bytes of %llu", (unsigned long long)bytes, (unsigned long long)have);
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t n = 0;
    if (int rc = mcull_download_count(c, &n)) return rc;  // MCULL_E_CAPACITY: the commands' firstTask values have wrapped
    if (bytes) {
        HIPCHK(c, hipMemcpyAsync(out, o.last.commands, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MCULL_OK;
}

int mcull_download_mask(mcull_ctx *c, uint64_t *out, uint64_t words) {
    if (!c || !out) return MCULL_E_INVALID;
    mcull_ctx::ViewOut &o = c->out[c->cur];
    if (!o.last.valid || !o.last.have_mask) return fail(c, MCULL_E_STATE, "no visibility mask kept");
    if (words > o.last.scene.n_groups) return fail(c, MCULL_E_INVALID, "requested more words than groups");
    HIPCHK(c, hipSetDevice(c->device));
    if (words) {
        HIPCHK(c, hipMemcpyAsync(out, o.mask.p, words * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MCULL_OK;
}

int mcull_slow_path_pairs(mcull_ctx *c, uint64_t *pairs) {
    if (!c || !pairs) return MCULL_E_INVALID;
    mcull_ctx::ViewOut &o = c->out[c->cur];
    if (!o.last.valid) return fail(c, MCULL_E_STATE, "no cull result");
    if (!o.counts_slow_pairs) return fail(c, MCULL_E_STATE, "the last cull did not run with MCULL_CULL_COUNT_SLOW_PAIRS");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(pairs, (uint64_t *)o.total.p + 16, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCULL_OK;
}

int mcull_precull_stats(mcull_ctx *c, uint64_t *instances_or_groups_marked, uint64_t *bytes_tested, int *per_group) {
    if (!c) return MCULL_E_INVALID;
    mcull_ctx::ViewOut &o = c->out[c->cur];
    if (!o.last.valid) return fail(c, MCULL_E_STATE, "no cull result");
    if (instances_or_groups_marked) *instances_or_groups_marked = 0;
    if (bytes_tested) *bytes_tested = o.last.precull_units;
    if (per_group) *per_group = o.last.scene.group_rec != nullptr;
    if (!o.last.work) return MCULL_OK;  // the pre-cull did not run for this cull
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t kept = 0;  // the list's length is the number of survivors
    HIPCHK(c, hipMemcpyAsync(&kept, o.last.work + 1, 4, hipMemcpyDeviceToHost, c->stream));  // word 1: see emit_commands_body
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint64_t n = o.last.precull_units - kept;
    if (instances_or_groups_marked) *instances_or_groups_marked = n;
    return MCULL_OK;
}

int mcull_precull_group_stats(mcull_ctx *c, uint64_t *groups_dropped, uint64_t *groups_tested) {
    if (!c) return MCULL_E_INVALID;
    mcull_ctx::ViewOut &o = c->out[c->cur];
    if (!o.last.valid) return fail(c, MCULL_E_STATE, "no cull result");
    if (groups_dropped) *groups_dropped = 0;
    if (groups_tested) *groups_tested = 0;
    if (!o.last.group_work) return MCULL_OK;  // the group level did not run for this cull
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t listed = 0, kept = 0;  // entries the instance level handed over, groups that survived
    HIPCHK(c, hipMemcpyAsync(&listed, o.last.work + 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&kept, o.last.group_work, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint64_t tested = o.last.scene.group_rec ? (uint64_t)listed : (uint64_t)listed * o.last.scene.groups_per_instance;
    if (groups_tested) *groups_tested = tested;
    if (groups_dropped) *groups_dropped = tested - kept;
    return MCULL_OK;
}

// ------------------------------------------------------------------------------------ depth source

namespace {

int raster_impl(mcull_ctx *c, const mcull_globals *g, const mcull_globals *g_dev, uint32_t width, uint32_t height, int clear) {
    if (!c) return MCULL_E_INVALID;
    mcull_ctx::ViewOut &o = c->out[c->cur];
    if (!g && !g_dev) return fail(c, MCULL_E_INVALID, "globals is null");
    if (width == 0 || height == 0 || width > 16384 || height > 16384) return fail(c, MCULL_E_INVALID, "depth size %ux%u", width, height);
    if (!o.last.valid) return fail(c, MCULL_E_STATE, "no cull result to rasterise");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_geometry(c)) return rc;
    // the list's meshlet ids index the AoS records: they must be the array the cull ran on
    if (o.last.scene.cone_axis_w != (const float4 *)c->cone.p)
        return fail(c, MCULL_E_STATE, "the last cull ran on adopted SoA bounds, not on the uploaded meshlets");
    // ... and the list's drawIDs index the transforms (and the per-instance matrices derived from them here)
    if ((uint64_t)o.last.scene.n_instances != c->n_instances)
        return fail(c, MCULL_E_STATE, "the transforms were replaced by %llu since the last cull ran on %u",
                    (unsigned long long)c->n_instances, o.last.scene.n_instances);
    const size_t bytes = (size_t)width * height * 4;
    const bool resized = c->depth_w != width || c->depth_h != height;
    if (int rc = ensure(c, c->depth, bytes)) return rc;
    c->depth_w = width;
    c->depth_h = height;
    const bool zero_first = clear || resized;  // clear value 0 (Engine.cpp:2404); the rasteriser's first launch does it
    float vp[16];
    const mcull::ViewConst *view_dev = nullptr;
    if (g) {
        mat4_mul(g->proj, g->view, vp);
    } else {
        if (int rc = ensure(c, c->view_dev, kViewDevBytes)) return rc;
        view_dev = (const mcull::ViewConst *)c->view_dev.p + MCULL_MAX_VIEWS;  // the rasteriser's own record
    }
    if (!c->raster_scratch.p) {
        if (int rc = ensure(c, c->raster_scratch, mcull::depth_raster_scratch_bytes())) return rc;
        HIPCHK(c, mcull::launch_zero_words((uint32_t *)c->raster_scratch.p, 64, c->stream));  // the queue counters; the kernels keep them zero between calls
    }
    if (int rc = ensure(c, c->raster_mvp, (size_t)(c->n_instances ? c->n_instances : 1) * 64)) return rc;
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (view_dev) {
        hipLaunchKernelGGL(derive_view_kernel, dim3(1), dim3(64), 0, c->stream, g_dev, mcull::ViewConst{},
                           (mcull::ViewConst *)c->view_dev.p + MCULL_MAX_VIEWS);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, mcull::launch_depth_raster(o.last.list, (const uint64_t *)o.total_ptr, o.last.capacity,
                                         (const mcull_meshlet *)c->aos.p, (const uint32_t *)c->mdata.p,
                                         (const mcull_vertex *)c->vertices.p, (const float *)c->transforms.p, c->n_instances,
                                         (float *)c->raster_mvp.p, g ? vp : nullptr, width, height, (float *)c->depth.p,
                                         zero_first, c->raster_scratch.p, c->stream, view_dev));
    if (c->timing) {
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        c->timed = true;
    }
    return MCULL_OK;
}

}  // namespace

int mcull_depth_rasterize(mcull_ctx *c, const mcull_globals *g, uint32_t width, uint32_t height, int clear) {
    if (c && !g) return fail(c, MCULL_E_INVALID, "globals is null");
    return raster_impl(c, g, nullptr, width, height, clear);
}

int mcull_depth_rasterize_device_globals(mcull_ctx *c, const mcull_globals *device_globals, uint32_t width, uint32_t height,
                                         int clear) {
    if (c && !device_globals) return fail(c, MCULL_E_INVALID, "device_globals is null");
    return raster_impl(c, nullptr, device_globals, width, height, clear);
}

int mcull_depth_buffer(mcull_ctx *c, const float **device_ptr, uint32_t *width, uint32_t *height) {
    if (!c) return MCULL_E_INVALID;
    if (!c->depth.p || !c->depth_w) return fail(c, MCULL_E_STATE, "no depth buffer has been rasterised");
    if (device_ptr) *device_ptr = (const float *)c->depth.p;
    if (width) *width = c->depth_w;
    if (height) *height = c->depth_h;
    return MCULL_OK;
}

int mcull_depth_download(mcull_ctx *c, float *out) {
    if (!c || !out) return MCULL_E_INVALID;
    if (!c->depth.p || !c->depth_w) return fail(c, MCULL_E_STATE, "no depth buffer has been rasterised");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->depth.p, (size_t)c->depth_w * c->depth_h * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCULL_OK;
}

int mcull_enable_timing(mcull_ctx *c, int enable) {
    if (!c) return MCULL_E_INVALID;
    c->timing = enable != 0;
    if (!c->timing) c->timed = false;
    return MCULL_OK;
}

int mcull_last_kernel_ms(mcull_ctx *c, float *ms) {
    if (!c || !ms) return MCULL_E_INVALID;
    if (!c->timed) return fail(c, MCULL_E_STATE, "nothing has been timed yet");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return MCULL_OK;
}

}  // extern "C"
