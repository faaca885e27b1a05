// Cull + compact kernels for gfx950 (wave64).
//
// Work unit: a *group* = 64 consecutive meshlets of one instance, one meshlet per lane, so the
// instance transform is wave-uniform (scalar loads -> SGPRs) and the two SoA bounds streams are
// read with one coalesced 16-byte load per lane each (1 KiB per wave-instruction).
//
// Arithmetic: this file is compiled with -ffp-contract=off and default (correctly rounded)
// division / sqrt, and every expression below fixes its association order, so that each test
// evaluates to the same IEEE binary32 value as the scalar CPU transcription of
// shaders/shader.task:37-101.
//
// Compaction strategies:
//   ordered      cull_mask_kernel  -> one 64-bit ballot word per group + one count per 256-group tile
//                scan_tiles_kernel -> (scenes above 1024 tiles) exclusive 64-bit prefix of the tile counts + the
//                                     total, one workgroup per 8192 tiles; smaller scenes sum the lower counters
//                                     inside the scatter pass instead of paying a launch
//                compact_kernel    -> per-tile block scan + scatter; list sorted by (drawID, meshletID)
//                emit_commands_kernel -> {taskCount, firstTask} per instance
//   wave/block atomic  cull_atomic_kernel: ballot + mbcnt prefix, one atomic per wavefront (or per
//                workgroup step), list order unspecified.
#include <cstring>
#include <type_traits>

#include "mcull_internal.h"

namespace mcull {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr uint32_t kTileGroups = 256;  // groups per compaction tile = 16384 meshlet-instances
// Groups staged in LDS per output round of the scatter pass (compact_body).  The pass stages either the eight bytes of
// an entry (WIDE: 32 KiB per 64 groups, four workgroups per CU) or two bytes per entry, from which the entry is put together
// when it is stored (8 KiB per 64 groups: the staging no longer limits how many workgroups share a CU, the stores cost a
// look-up more).  Measured against each other on one GPU, eight bytes / two bytes:
//   dense list of one view, one workgroup per tile (100 M pairs, half visible)   0.5001 / 0.5019 ms per step  -> WIDE, 64 groups
//   dense lists of eight views (37 - 77 % visible)                               2.773 / 2.740 ms            -> two bytes, 128 groups
//     (two bytes with 64 / 128 / 256 groups per round: 2.746 / 2.731 / 2.743 ms)
//   the forms that wait for latency rather than for bytes - the tile-walking form (scenes that skip empty tiles) and
//   SHORT, one workgroup per tile behind a work list expected to hold under a quarter of the scene's groups:
//     1 M instances of one mesh 16.10 / 16.00 ms, 100 M pairs with the occlusion test (4 % visible) 0.199 / 0.196 ms
//                                                                                 -> two bytes, 64 groups (128: 16.05, 0.197)
//   (with eight bytes these forms wanted rounds of 32 groups, seven workgroups per CU: 1.72 -> 1.37 ms and 51 -> 37 us for the pass)
constexpr uint32_t kRoundGroups = 128;       // two-byte staging, dense lists
constexpr uint32_t kShortRoundGroups = 64;   // every other form
constexpr uint32_t kScanMinTiles = 1024;  // scenes with more tiles get their tile offsets from scan_tiles_kernel
constexpr int kScanBlock = 1024;          // threads of that kernel's single workgroup
constexpr int kScanPerThread = 8;         // consecutive counters per thread and iteration

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;
}

// clamp(ceil(log2(max(px, 1))), 0, levels-1) evaluated exactly from the float's bits
// (shader.task:71-72).  NaN and px <= 1 give level 0.
__device__ __forceinline__ uint32_t mip_level(float px, uint32_t levels) {
    if (!(px > 1.0f)) return 0u;
    uint32_t bits = __float_as_uint(px);
    uint32_t e = (bits >> 23) & 0xffu, mant = bits & 0x7fffffu;
    uint32_t l = (e == 0xffu) ? 0xffu : (e - 127u) + (mant ? 1u : 0u);
    return l > levels - 1u ? levels - 1u : l;
}

// nearest + REPEAT addressing (Engine.cpp:1623-1627): floor(u*size) mod size; a scaled
// coordinate that is not finite and below 1e9 in magnitude addresses texel 0.
__device__ __forceinline__ uint32_t wrap_texel(float scaled_floor, uint32_t size) {
    if (!(fabsf(scaled_floor) < 1.0e9f)) return 0u;
    int32_t i = (int32_t)scaled_floor;
    int32_t m = i % (int32_t)size;
    if (m < 0) m += (int32_t)size;
    return (uint32_t)m;
}

struct Xform {
    float4 c0, c1, c2, c3;  // columns of the model matrix
};

// length(transform.model[0].xyz) (shader.task:99): uniform per instance, so it is evaluated once
// per uploaded transform (same expression, same rounding) instead of once per pair.
__device__ __forceinline__ float xform_scale(float4 c0) { return sqrtf(dot3(c0.x, c0.y, c0.z, c0.x, c0.y, c0.z)); }

// normalize(mat3(model) * cone.xyz) (shader.task:95): independent of the view
__device__ __forceinline__ float4 world_axis(const Xform &m, float4 a) {
    float ax = (m.c0.x * a.x + m.c1.x * a.y) + m.c2.x * a.z;
    float ay = (m.c0.y * a.x + m.c1.y * a.y) + m.c2.y * a.z;
    float az = (m.c0.z * a.x + m.c1.z * a.y) + m.c2.z * a.z;
    const float ia = 1.0f / sqrtf(dot3(ax, ay, az, ax, ay, az));
    return make_float4(ax * ia, ay * ia, az * ia, a.w);
}

// ------------------------------------------------------------------------------------------------
// Guarded-exact evaluation of the cone and occlusion tests.
// What must equal the scalar CPU transcription is every DECISION (the selected mip level, the texel, the
// two comparisons), not every intermediate.  The tests are therefore evaluated first with the hardware's
// approximate reciprocal and reciprocal square root (v_rcp_f32 / v_rsq_f32, one instruction each, accurate
// to 1 ulp; the bounds below allow 2) in place of the correctly rounded divisions and square roots (about
// a dozen instructions each), together with a bound on how far each quantity can lie from the value of the
// IEEE expression.  Where that bound cannot change a decision the decision stands; the lanes for which it
// could - a dot product within 1e-5 of the cone boundary, a scaled texture coordinate within the bound of
// a texel edge, a pixel radius next to a power of two, a depth next to the threshold, anything that is
// not an ordinary finite number - re-evaluate that quantity with the IEEE expression, behind a wave-wide
// branch that is not taken when no lane needs it.  Results are bit-identical to the all-IEEE evaluation.
__device__ __forceinline__ bool ordinary(float x) {  // finite, not near under- or overflow; false for NaN
    const float a = fabsf(x);
    return a > 1.0e-18f && a < 1.0e18f;
}

// lanes of this wave that left the fast path, for the statistics of a cull that asks for them
__device__ __forceinline__ void note_slow(const ViewConst &v, bool slow) {
    if (v.slow_pairs) {
        const uint64_t b = __ballot(slow);
        if (slow && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(b)) atomicAdd(v.slow_pairs, (unsigned long long)__popcll(b));
    }
}

// coneCull (shader.task:37-41) of a pair that is still `live`: dot(axis_w, normalize(cam - centre_w)) < -coneW.
// AXIS_READY: `a` already holds the world-space unit axis (world_axis, IEEE), as when several views test the
// same pair; otherwise it is the meshlet's local axis.
// Fast form: both normalisations with v_rsq_f32.  With |axis| and |view| within roundoff of 1 the dot product
// differs from the IEEE expression's by less than 2e-6 (two reciprocal square roots at 2 ulp, nine roundings
// of products and sums of magnitude <= 1), so a margin dot + coneW beyond 1e-5 decides; squared lengths
// outside [1e-18, 1e18], NaNs and infinities fail the guard.
template <bool AXIS_READY>
__device__ __forceinline__ bool cone_culled(const ViewConst &v, const Xform &m, float wx, float wy, float wz, float4 a, bool live) {
    float nx = a.x, ny = a.y, nz = a.z, ax = a.x, ay = a.y, az = a.z, la = 1.0f;
    if (!AXIS_READY) {
        nx = (m.c0.x * a.x + m.c1.x * a.y) + m.c2.x * a.z;
        ny = (m.c0.y * a.x + m.c1.y * a.y) + m.c2.y * a.z;
        nz = (m.c0.z * a.x + m.c1.z * a.y) + m.c2.z * a.z;
        la = dot3(nx, ny, nz, nx, ny, nz);
        const float ia = __builtin_amdgcn_rsqf(la);
        ax = nx * ia; ay = ny * ia; az = nz * ia;
    }
    const float vx = v.cam[0] - wx, vy = v.cam[1] - wy, vz = v.cam[2] - wz;
    const float lv = dot3(vx, vy, vz, vx, vy, vz);
    const float iv = __builtin_amdgcn_rsqf(lv);
    const float d = dot3(ax, ay, az, vx * iv, vy * iv, vz * iv);
    bool culled = d < -a.w;
    const bool sure = fabsf(d + a.w) > 1.0e-5f && ordinary(lv) && ordinary(la) && !v.force_exact;
    if (__any(live && !sure)) {
        if (live && !sure) {  // the IEEE expression
            if (!AXIS_READY) {
                const float ia = 1.0f / sqrtf(la);
                ax = nx * ia; ay = ny * ia; az = nz * ia;
            }
            const float ie = 1.0f / sqrtf(lv);
            culled = dot3(ax, ay, az, vx * ie, vy * ie, vz * ie) < -a.w;
        }
        note_slow(v, live && !sure);
    }
    return culled;
}

// the mip level changes at powers of two: is px, known to a relative 2^-20, next to one?  (px <= 0.75: level 0 for certain)
__device__ __forceinline__ bool near_power_of_two(float px) {
    const uint32_t mant = __float_as_uint(px) & 0x7fffffu;
    return px > 0.75f && (mant < 16u || mant > 0x7fffffu - 16u);
}

// nearest + REPEAT addressing, the common case first: a texel index already inside the level needs no division
__device__ __forceinline__ uint32_t wrap_texel_fast(float scaled_floor, uint32_t size) {
    if (!(fabsf(scaled_floor) < 1.0e9f)) return 0u;
    const int32_t i = (int32_t)scaled_floor;
    if ((uint32_t)i < size) return (uint32_t)i;
    int32_t mm = i % (int32_t)size;
    if (mm < 0) mm += (int32_t)size;
    return (uint32_t)mm;
}

// occlusionCull (shader.task:58-77) of a pair that is still `live`, from its clip-space centre (cx, cy, cz, cw)
// and pr = proj[1][1] * r.
// Fast form: one v_rcp_f32 of cw instead of four divisions by it.  A quotient q = c * rcp(cw) differs from the
// correctly rounded c / cw by at most |q| 2^-22 (reciprocal at 1 ulp; 1.25 * 2^-22 at 2 ulp), hence
//   pixel radius: relative 2^-21 at most  -> certain unless next to a power of two (near_power_of_two);
//   texel:        tu = q/2 + 1/2 by |q| 2^-23 + |tu| 2^-23, s = tu * width by width 2^-23 (1.5 |q| + 2 |tu|)
//                 <= E = width 2^-22 (|q| + |tu|)  -> certain when s is farther than E from both neighbouring integers;
//   depth:        certain when |nz - (d - 0.05)| > |nz| 2^-21.
// cw outside [1e-18, 1e18] or a numerator beyond 1e18 (or NaN) sends the lane to the divisions for everything.
__device__ __forceinline__ bool occluded(const ViewConst &v, float cx, float cy, float cz, float cw, float pr, bool live) {
    const bool plain = ordinary(cw) && fabsf(cx) < 1.0e18f && fabsf(cy) < 1.0e18f && fabsf(cz) < 1.0e18f && fabsf(pr) < 1.0e18f &&
                       !v.force_exact;
    const float rc = __builtin_amdgcn_rcpf(cw);
    bool slow = false;
    // mip level (shader.task:63, :68-72)
    float px = ((pr * rc) * 2.0f) * v.tex_max;
    {
        const bool unsure = !plain || near_power_of_two(px);
        if (__any(live && unsure)) {
            if (unsure) px = ((pr / cw) * 2.0f) * v.tex_max;
            slow |= unsure;
        }
    }
    const uint32_t mip = mip_level(px, v.levels);
    // level extents are arithmetic; only the level's offset in the flat pyramid comes from the table
    const uint32_t lw = max(1u, v.w0 >> mip), lh = max(1u, v.h0 >> mip), lo = v.level_table[32 + mip];
    const float fw = (float)lw, fh = (float)lh;
    // texel column (shader.task:62, :65, :74)
    float nx = cx * rc;
    float tu = nx * 0.5f + 0.5f;
    float sx = tu * fw, fx = floorf(sx);
    {
        const float e = fw * 2.384185791015625e-7f * (fabsf(nx) + fabsf(tu)), f = sx - fx;  // 2^-22
        const bool unsure = !plain || !(f > e && f < 1.0f - e);
        if (__any(live && unsure)) {
            if (unsure) {
                nx = cx / cw;
                tu = nx * 0.5f + 0.5f;
                fx = floorf(tu * fw);
            }
            slow |= unsure;
        }
    }
    // texel row (shader.task:62, :65-66, :74)
    float ny = cy * rc;
    float th = ny * 0.5f + 0.5f;
    float tv = 1.0f - th;
    float sy = tv * fh, fy = floorf(sy);
    {
        // 1 - th adds one more rounding: |ny| 2^-23 + |th| 2^-23 + |tv| 2^-23, times the height, plus the product's
        const float e = fh * 2.384185791015625e-7f * ((fabsf(ny) + fabsf(th)) + fabsf(tv)), f = sy - fy;
        const bool unsure = !plain || !(f > e && f < 1.0f - e);
        if (__any(live && unsure)) {
            if (unsure) {
                ny = cy / cw;
                tv = 1.0f - (ny * 0.5f + 0.5f);
                fy = floorf(tv * fh);
            }
            slow |= unsure;
        }
    }
    const uint32_t tx = wrap_texel_fast(fx, lw), ty = wrap_texel_fast(fy, lh);
    float nz = cz * rc;
    // Before the texel, the largest depth of the 64 x 8 strip of the depth image that holds its footprint (levels
    // 0-3: at most 8 x 8 texels, aligned; the table is 1/512 of level 0 and stays in the caches): the texel is not
    // above it, so a pair whose depth is certainly not below that bound minus the bias is not occluded, whatever
    // the texel holds, and does not read it.  (A NaN texel compares false in the test itself: visible either way.)
    bool need = live;
    if (v.strips_per_pair) {
        const bool fine = mip <= 3u;
        float smax = 0.0f;
        if (live && fine) smax = v.strips[(size_t)((ty << mip) >> 3) * v.strips_w + ((tx << mip) >> 6)];
        const bool clear = plain && fine && ((nz - fabsf(nz) * 4.76837158203125e-7f) - 1.0e-37f >= smax - 0.05f);
        need = live && !clear;
    }
    float d = 0.0f;
    if (need) d = v.pyramid[(size_t)lo + (size_t)ty * lw + tx];  // the one data-dependent read: only for pairs still in the running
    // depth comparison (shader.task:62, :76)
    const float thr = d - 0.05f;
    {
        const bool unsure = !plain || !(fabsf(nz - thr) > fabsf(nz) * 4.76837158203125e-7f + 1.0e-37f);  // 2^-21
        if (__any(need && unsure)) {
            if (unsure) nz = cz / cw;
            slow |= need && unsure;
        }
    }
    if (v.slow_pairs && __any(live && slow)) note_slow(v, live && slow);
    return need && nz < thr;
}

// The cone and occlusion tests of one pair whose world-space centre (wx, wy, wz) and radius r are
// known; `culled` is the frustum test's verdict.  Returns the final verdict.
// accept = !frustum && !occlusion && !cone (shader.task:101) is a conjunction, so the order of the
// tests cannot change it, only the cost.  The occlusion test's pyramid read is the one
// data-dependent memory access; the cone test is arithmetic only.  CONE_FIRST (bounds streamed from HBM,
// the pass is memory-bound): the cone test runs first and about half the pairs never issue the read
// (8 views at 8192^2: 11.1 -> 10.3 ms).  Otherwise (bounds cache-resident, the pass is arithmetic-bound): the
// occlusion test runs first, because it empties whole waves, which then skip the cone arithmetic (1 M
// instances of one mesh: 68 ms against 88 ms with the cone test first).
template <bool CONE_FIRST, bool AXIS_READY = false>
__device__ __forceinline__ bool cone_and_occlusion(const uint32_t TESTS, const ViewConst &v, const Xform &m, float wx, float wy,
                                                   float wz, float r, float4 a, bool culled) {
    auto cone_test = [&]() {
        if ((TESTS & MCULL_TEST_CONE) && __any(!culled)) {
            const bool c = cone_culled<AXIS_READY>(v, m, wx, wy, wz, a, !culled);
            culled |= c;
        }
    };
    if (CONE_FIRST) cone_test();
    if ((TESTS & MCULL_TEST_HIZ) && __any(!culled)) {
        const float cx = ((v.vp[0] * wx + v.vp[4] * wy) + v.vp[8] * wz) + v.vp[12];
        const float cy = ((v.vp[1] * wx + v.vp[5] * wy) + v.vp[9] * wz) + v.vp[13];
        const float cz = ((v.vp[2] * wx + v.vp[6] * wy) + v.vp[10] * wz) + v.vp[14];
        const float cw = ((v.vp[3] * wx + v.vp[7] * wy) + v.vp[11] * wz) + v.vp[15];
        const bool o = occluded(v, cx, cy, cz, cw, v.proj11 * r, !culled);
        culled |= o;
    }
    if (!CONE_FIRST) cone_test();
    return culled;
}

// shader.task:85-101 for one pair.  `a` = {axis.xyz, coneW}, `c` = {centre.xyz, radius}.
template <bool CONE_FIRST>
__device__ __forceinline__ bool pair_visible_impl(const uint32_t TESTS, const ViewConst &v, const Xform &m, float scale, float4 a,
                                                  float4 c) {
    // world-space centre (shader.task:91), axis (:95), radius (:99)
    const float wx = ((m.c0.x * c.x + m.c1.x * c.y) + m.c2.x * c.z) + m.c3.x;
    const float wy = ((m.c0.y * c.x + m.c1.y * c.y) + m.c2.y * c.z) + m.c3.y;
    const float wz = ((m.c0.z * c.x + m.c1.z * c.y) + m.c2.z * c.z) + m.c3.z;
    const float r = c.w * scale;  // scale = length(model[0].xyz), evaluated once per instance

    bool culled = false;
    if (TESTS & MCULL_TEST_FRUSTUM) {  // frustumCull, shader.task:43-56
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const float d = (dot3(v.frustum[i][0], v.frustum[i][1], v.frustum[i][2], wx, wy, wz) + v.frustum[i][3]) + r;
            culled |= d < 0.0f;
        }
    }
    return !cone_and_occlusion<CONE_FIRST>(TESTS, v, m, wx, wy, wz, r, a, culled);
}

// One pair against one of several views: world-space centre {wx, wy, wz, r} and unit axis {ax, ay, az, coneW}
// were evaluated once for all views.
template <bool CONE_FIRST>
__device__ __forceinline__ bool pair_visible_prepared(const uint32_t TESTS, const ViewConst &v, const Xform &m, float4 w, float4 wa) {
    bool culled = false;
    if (TESTS & MCULL_TEST_FRUSTUM) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const float d = (dot3(v.frustum[i][0], v.frustum[i][1], v.frustum[i][2], w.x, w.y, w.z) + v.frustum[i][3]) + w.w;
            culled |= d < 0.0f;
        }
    }
    return !cone_and_occlusion<CONE_FIRST, true>(TESTS, v, m, w.x, w.y, w.z, w.w, wa, culled);
}

__device__ __forceinline__ float4 world_centre(const Xform &m, float scale, float4 c) {
    return make_float4(((m.c0.x * c.x + m.c1.x * c.y) + m.c2.x * c.z) + m.c3.x, ((m.c0.y * c.x + m.c1.y * c.y) + m.c2.y * c.z) + m.c3.y,
                       ((m.c0.z * c.x + m.c1.z * c.y) + m.c2.z * c.z) + m.c3.z, c.w * scale);
}

// The same tests for two pairs at once, written on two-element vectors: on gfx950 the compiler emits
// the packed binary32 instructions (v_pk_mul_f32 / v_pk_add_f32: two independent IEEE operations per
// instruction, each rounded exactly like the scalar form), which halves the multiply/add count of a
// pass that is arithmetic-bound when the bounds are cache-resident.  Element e of every vector
// belongs to pair e; divisions, square roots, comparisons and the texel fetch stay per element.
// Expression trees and association order are those of pair_visible_impl, term for term.
// in0 / in1: the lane's pair exists.  The cone and occlusion stages are skipped by a wave only when
// no lane of either group is left, which is why this form is used for streamed bounds only: there
// it is 3 % faster (100 M pairs: 0.574 ms against 0.594 ms), while a cache-resident, arithmetic-bound
// scene loses more from the coarser skipping than it gains (1 M instances of one mesh: 72.8 ms
// against 69.3 ms).  Choosing per wave between this form and the scalar one (packed only while
// both groups have live lanes) was slower than either on both scenes (0.626 ms, 80.4 ms).
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v dot3v(f2v ax, f2v ay, f2v az, f2v bx, f2v by, f2v bz) { return (ax * bx + ay * by) + az * bz; }
__device__ __forceinline__ f2v splat(float x) { return f2v{x, x}; }

struct Centre2 {
    f2v wx, wy, wz, r;
};
struct Axis2 {
    f2v x, y, z;
};
#define MCULL_M2(col, comp) (f2v{m0.col.comp, m1.col.comp})
__device__ __forceinline__ Centre2 world_centre2(const Xform &m0, const Xform &m1, float scale0, float scale1, float4 c0, float4 c1) {
    const f2v cx = {c0.x, c1.x}, cy = {c0.y, c1.y}, cz = {c0.z, c1.z};
    Centre2 w;
    w.wx = ((MCULL_M2(c0, x) * cx + MCULL_M2(c1, x) * cy) + MCULL_M2(c2, x) * cz) + MCULL_M2(c3, x);
    w.wy = ((MCULL_M2(c0, y) * cx + MCULL_M2(c1, y) * cy) + MCULL_M2(c2, y) * cz) + MCULL_M2(c3, y);
    w.wz = ((MCULL_M2(c0, z) * cx + MCULL_M2(c1, z) * cy) + MCULL_M2(c2, z) * cz) + MCULL_M2(c3, z);
    w.r = f2v{c0.w, c1.w} * f2v{scale0, scale1};
    return w;
}
__device__ __forceinline__ Axis2 world_axis2(const Xform &m0, const Xform &m1, float4 a0, float4 a1) {
    const f2v Ax = {a0.x, a1.x}, Ay = {a0.y, a1.y}, Az = {a0.z, a1.z};
    Axis2 n;
    n.x = (MCULL_M2(c0, x) * Ax + MCULL_M2(c1, x) * Ay) + MCULL_M2(c2, x) * Az;
    n.y = (MCULL_M2(c0, y) * Ax + MCULL_M2(c1, y) * Ay) + MCULL_M2(c2, y) * Az;
    n.z = (MCULL_M2(c0, z) * Ax + MCULL_M2(c1, z) * Ay) + MCULL_M2(c2, z) * Az;
    const f2v la = dot3v(n.x, n.y, n.z, n.x, n.y, n.z);
    const f2v ia = {1.0f / sqrtf(la.x), 1.0f / sqrtf(la.y)};
    n.x *= ia; n.y *= ia; n.z *= ia;
    return n;
}
#undef MCULL_M2

// AXIS_READY: `axis` holds the unit world-space axes (world_axis2, IEEE), evaluated once for several views;
// otherwise they are evaluated here, and only if the cone test runs.  The cone and occlusion tests are the
// guarded-exact forms (cone_culled, occluded): packed arithmetic up to the reciprocals, then per element.
template <bool CONE_FIRST, bool AXIS_READY>
__device__ __forceinline__ void tests2(const uint32_t TESTS, const ViewConst &v, const Xform &m0, const Xform &m1, const Centre2 &w,
                                       const Axis2 &axis, float4 a0, float4 a1, bool in0, bool in1, bool &vis0, bool &vis1) {
    const f2v wx = w.wx, wy = w.wy, wz = w.wz, r = w.r;
    bool k0 = !in0, k1 = !in1;  // culled (a pair that does not exist counts as culled from the start)
    if (TESTS & MCULL_TEST_FRUSTUM) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const f2v d = (dot3v(splat(v.frustum[i][0]), splat(v.frustum[i][1]), splat(v.frustum[i][2]), wx, wy, wz) +
                           splat(v.frustum[i][3])) + r;
            k0 |= d.x < 0.0f;
            k1 |= d.y < 0.0f;
        }
    }
    auto cone_test = [&]() {
        if ((TESTS & MCULL_TEST_CONE) && __any(!(k0 && k1))) {
            f2v nx, ny, nz, la = splat(1.0f), ax, ay, az;
            if (AXIS_READY) {
                ax = axis.x; ay = axis.y; az = axis.z;
                nx = ax; ny = ay; nz = az;
            } else {
                const f2v Ax = {a0.x, a1.x}, Ay = {a0.y, a1.y}, Az = {a0.z, a1.z};
#define MCULL_M2(col, comp) (f2v{m0.col.comp, m1.col.comp})
                nx = (MCULL_M2(c0, x) * Ax + MCULL_M2(c1, x) * Ay) + MCULL_M2(c2, x) * Az;
                ny = (MCULL_M2(c0, y) * Ax + MCULL_M2(c1, y) * Ay) + MCULL_M2(c2, y) * Az;
                nz = (MCULL_M2(c0, z) * Ax + MCULL_M2(c1, z) * Ay) + MCULL_M2(c2, z) * Az;
#undef MCULL_M2
                la = dot3v(nx, ny, nz, nx, ny, nz);
                const f2v ia = {__builtin_amdgcn_rsqf(la.x), __builtin_amdgcn_rsqf(la.y)};
                ax = nx * ia; ay = ny * ia; az = nz * ia;
            }
            const f2v vx = splat(v.cam[0]) - wx, vy = splat(v.cam[1]) - wy, vz = splat(v.cam[2]) - wz;
            const f2v lv = dot3v(vx, vy, vz, vx, vy, vz);
            const f2v iv = {__builtin_amdgcn_rsqf(lv.x), __builtin_amdgcn_rsqf(lv.y)};
            const f2v d = dot3v(ax, ay, az, vx * iv, vy * iv, vz * iv);
            const f2v cw = {a0.w, a1.w};
            const f2v margin = d + cw;
            bool c0 = d.x < -a0.w, c1 = d.y < -a1.w;
            const bool sure0 = fabsf(margin.x) > 1.0e-5f && ordinary(lv.x) && ordinary(la.x) && !v.force_exact;
            const bool sure1 = fabsf(margin.y) > 1.0e-5f && ordinary(lv.y) && ordinary(la.y) && !v.force_exact;
            const bool slow0 = !k0 && !sure0, slow1 = !k1 && !sure1;
            if (__any(slow0 || slow1)) {  // the IEEE expression for the lanes that need it (see cone_culled)
                auto exact = [&](float nx_, float ny_, float nz_, float la_, float ax_, float ay_, float az_, float vx_, float vy_,
                                 float vz_, float lv_, float cw_) {
                    if (!AXIS_READY) {
                        const float ia = 1.0f / sqrtf(la_);
                        ax_ = nx_ * ia; ay_ = ny_ * ia; az_ = nz_ * ia;
                    }
                    const float ie = 1.0f / sqrtf(lv_);
                    return dot3(ax_, ay_, az_, vx_ * ie, vy_ * ie, vz_ * ie) < -cw_;
                };
                if (slow0) c0 = exact(nx.x, ny.x, nz.x, la.x, ax.x, ay.x, az.x, vx.x, vy.x, vz.x, lv.x, a0.w);
                if (slow1) c1 = exact(nx.y, ny.y, nz.y, la.y, ax.y, ay.y, az.y, vx.y, vy.y, vz.y, lv.y, a1.w);
                note_slow(v, slow0);
                note_slow(v, slow1);
            }
            if (!k0) k0 = c0;
            if (!k1) k1 = c1;
        }
    };
    if (CONE_FIRST) cone_test();
    if ((TESTS & MCULL_TEST_HIZ) && __any(!(k0 && k1))) {
        const f2v px_ = ((splat(v.vp[0]) * wx + splat(v.vp[4]) * wy) + splat(v.vp[8]) * wz) + splat(v.vp[12]);
        const f2v py_ = ((splat(v.vp[1]) * wx + splat(v.vp[5]) * wy) + splat(v.vp[9]) * wz) + splat(v.vp[13]);
        const f2v pz_ = ((splat(v.vp[2]) * wx + splat(v.vp[6]) * wy) + splat(v.vp[10]) * wz) + splat(v.vp[14]);
        const f2v pw_ = ((splat(v.vp[3]) * wx + splat(v.vp[7]) * wy) + splat(v.vp[11]) * wz) + splat(v.vp[15]);
        const f2v pr = splat(v.proj11) * r;
        if (__any(!k0)) {
            const bool o = occluded(v, px_.x, py_.x, pz_.x, pw_.x, pr.x, !k0);
            if (!k0) k0 = o;
        }
        if (__any(!k1)) {
            const bool o = occluded(v, px_.y, py_.y, pz_.y, pw_.y, pr.y, !k1);
            if (!k1) k1 = o;
        }
    }
    if (!CONE_FIRST) cone_test();
    vis0 = !k0;
    vis1 = !k1;
}

template <bool CONE_FIRST>
__device__ __forceinline__ void pair_visible2_impl(const uint32_t TESTS, const ViewConst &v, const Xform &m0, const Xform &m1,
                                                   float scale0, float scale1, float4 a0, float4 a1, float4 c0, float4 c1,
                                                   bool in0, bool in1, bool &vis0, bool &vis1) {
    const Centre2 w = world_centre2(m0, m1, scale0, scale1, c0, c1);
    tests2<CONE_FIRST, false>(TESTS, v, m0, m1, w, Axis2{}, a0, a1, in0, in1, vis0, vis1);
}

// The tests as a compile-time constant: the single-view kernels are instantiated per combination.
template <uint32_t TESTS, bool CONE_FIRST>
__device__ __forceinline__ bool pair_visible(const ViewConst &v, const Xform &m, float scale, float4 a, float4 c) {
    return pair_visible_impl<CONE_FIRST>(TESTS, v, m, scale, a, c);
}

// Which instance / meshlet range a group belongs to.  All values are wave-uniform.
// Scenes with per-instance ranges carry one precomputed 16-byte record per group
// {instance, first meshlet, meshlets in this group (<= 64), local group index}; scenes where
// every instance draws [0, meshletCount) (the reference's case, shader.task:85) derive it.
struct GroupInfo {
    uint32_t inst, first_meshlet, count, local_group;
};

__device__ __forceinline__ GroupInfo group_lookup(uint32_t g, const uint4 *__restrict__ group_rec,
                                                  uint32_t groups_per_instance, uint32_t uniform_count) {
    GroupInfo gi;
    if (group_rec) {
        const uint4 r = group_rec[g];
        gi.inst = r.x; gi.first_meshlet = r.y; gi.count = r.z; gi.local_group = r.w;
    } else {
        gi.inst = g / groups_per_instance;
        gi.local_group = g - gi.inst * groups_per_instance;
        gi.first_meshlet = gi.local_group * 64u;
        const uint32_t left = uniform_count - gi.first_meshlet;
        gi.count = left < 64u ? left : 64u;
    }
    return gi;
}

__device__ __forceinline__ Xform load_xform(const float4 *__restrict__ transforms, uint32_t inst) {
    const float4 *t = transforms + (size_t)inst * 4;
    Xform m;
    m.c0 = t[0]; m.c1 = t[1]; m.c2 = t[2]; m.c3 = t[3];
    return m;
}

// 16-byte load of one SoA element; NT = streaming (non-temporal) hint for bounds arrays far larger
// than the caches, which are read exactly once per cull.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned long long ull2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ float4 load_bounds(const float4 *__restrict__ p) {
    if (NT) {
        const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
        return make_float4(q.x, q.y, q.z, q.w);
    }
    return *p;
}

// One visible-list entry as a single 8-byte store.  NT = streaming hint: a list far larger than the
// caches would otherwise sit dirty in L2 / Infinity Cache and its write-back would collide with the
// next kernel's reads (measured: the following cull pass runs 20 % slower after plain stores).
template <bool NT>
__device__ __forceinline__ void store_entry(mcull_visible *__restrict__ list, unsigned long long slot, uint32_t draw,
                                            uint32_t meshlet) {
    const unsigned long long e = ((unsigned long long)meshlet << 32) | draw;
    if (NT) __builtin_nontemporal_store(e, reinterpret_cast<unsigned long long *>(list) + slot);
    else reinterpret_cast<unsigned long long *>(list)[slot] = e;
}

// exclusive scan of one value per thread over a workgroup of WAVES waves; returns the exclusive
// prefix and writes the workgroup total to *total_out (uniform).
template <int WAVES = kWaves>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t x, uint32_t *s_wave /*[WAVES]*/, uint32_t *total_out) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
        const uint32_t t = s_wave[w];
        if ((uint32_t)w < wave) wave_base += t;
        total += t;
    }
    *total_out = total;
    __syncthreads();  // s_wave may be reused by the caller's next scan
    return wave_base + incl - x;
}

// ------------------------------------------------------------------------------------------------
// Instance-level pre-cull (no counterpart in the reference, which launches every task group of every
// draw, Engine.cpp:1475-1483).  Each meshlet range gets one enclosing sphere in mesh space; a cull first
// tests every instance's sphere against the five planes and marks the instances that are certainly
// outside, and the mask pass then writes zero ballots for their groups without loading or testing a
// single meshlet.  "Certainly" covers the rounding of the per-meshlet expression, so the ballots are
// identical to those of the full evaluation (see instance_classify).

// Enclosing sphere {C.xyz, R} of the meshlet spheres of one range: C = mean of the centres,
// R >= max(|c_m - C| + r_m) (rounded up).  One workgroup per range.  A non-finite centre or radius
// anywhere in the range, or a range above 2^20 meshlets (one workgroup would take too long), gives
// R = +inf: such a range is never pre-culled.
// rmax[i] >= every meshlet radius of the range (what bounds the mip level its meshlets can select); +inf with R,
// and when a radius of the range is negative (the sign of the projected radius then differs inside the range).
// boxes[i] = {hx, hy, hz, 0}: no meshlet CENTRE of the range is farther than h from C along a mesh-space axis (rounded up;
// +inf with R) - the box that a flat or elongated mesh fills far better than the sphere (range_boxes).
// sliding: sphere i encloses the meshlets [i, i + 64) of an array of uniform_count meshlets instead - whatever group
// of at most 64 meshlets starts at meshlet i, of any range, lies inside it.
// PER_WAVE: one wave per range instead (kWaves ranges per workgroup, n_ranges of them in all; nothing goes through
// shared memory and no wave waits for another) - the shape for many short ranges: 781 250 ranges of 128 meshlets
// take 2.3 ms with a workgroup each, most of it workgroup launches and barriers.
template <bool PER_WAVE>
__global__ __launch_bounds__(kBlock) void range_sphere_kernel(const float4 *__restrict__ centre_radius,
                                                              const uint2 *__restrict__ ranges, uint32_t uniform_count,
                                                              uint32_t sliding, uint32_t n_ranges, float4 *__restrict__ spheres,
                                                              float *__restrict__ rmax, float4 *__restrict__ boxes) {
    __shared__ float s_red[PER_WAVE ? 1 : kWaves][6];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t i = PER_WAVE ? blockIdx.x * kWaves + wave : blockIdx.x;
    if (PER_WAVE && i >= n_ranges) return;  // whole waves leave, and this form has no barrier
    const uint32_t rank = PER_WAVE ? lane : threadIdx.x, stride = PER_WAVE ? 64u : (uint32_t)kBlock;
    uint32_t first = ranges ? ranges[i].x : 0u, count = ranges ? ranges[i].y : uniform_count;
    if (sliding) {
        first = i;
        count = uniform_count - i < 64u ? uniform_count - i : 64u;
    }
    if (count == 0u || count > (1u << 20)) {
        if (rank == 0) {
            spheres[i] = make_float4(0.0f, 0.0f, 0.0f, count ? __builtin_inff() : 0.0f);
            rmax[i] = count ? __builtin_inff() : 0.0f;
            const float h = count ? __builtin_inff() : 0.0f;
            boxes[i] = make_float4(h, h, h, 0.0f);
        }
        return;
    }
    float sx = 0.0f, sy = 0.0f, sz = 0.0f, bad = 0.0f, neg = 0.0f;
    for (uint32_t j = rank; j < count; j += stride) {
        const float4 c = centre_radius[first + j];
        sx += c.x; sy += c.y; sz += c.z;
        if (!(fabsf(c.x) < 1.0e18f && fabsf(c.y) < 1.0e18f && fabsf(c.z) < 1.0e18f && fabsf(c.w) < 1.0e18f)) bad = 1.0f;
        if (c.w < 0.0f) neg = 1.0f;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sx += __shfl_xor(sx, d, 64); sy += __shfl_xor(sy, d, 64); sz += __shfl_xor(sz, d, 64);
        bad = fmaxf(bad, __shfl_xor(bad, d, 64));
    }
    const float inv = 1.0f / (float)count;
    float cx, cy, cz;
    if (PER_WAVE) {  // the xor reduction left every lane with the wave's sums
        cx = sx * inv; cy = sy * inv; cz = sz * inv;
    } else {
        if (lane == 0) { s_red[wave][0] = sx; s_red[wave][1] = sy; s_red[wave][2] = sz; s_red[wave][3] = bad; }
        __syncthreads();
        cx = ((s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0])) * inv;
        cy = ((s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1])) * inv;
        cz = ((s_red[0][2] + s_red[1][2]) + (s_red[2][2] + s_red[3][2])) * inv;
        bad = fmaxf(fmaxf(s_red[0][3], s_red[1][3]), fmaxf(s_red[2][3], s_red[3][3]));
        __syncthreads();
    }
    // any point works as the centre; what must hold is the radius, measured from the centre as stored
    float r = 0.0f, rm = 0.0f, hx = 0.0f, hy = 0.0f, hz = 0.0f;
    for (uint32_t j = rank; j < count; j += stride) {
        const float4 c = centre_radius[first + j];
        const float dx = c.x - cx, dy = c.y - cy, dz = c.z - cz;
        // |c - C| + r, each step rounded to nearest: the factor below covers those roundings
        r = fmaxf(r, sqrtf((dx * dx + dy * dy) + dz * dz) + fabsf(c.w));
        rm = fmaxf(rm, fabsf(c.w));
        hx = fmaxf(hx, fabsf(dx)); hy = fmaxf(hy, fabsf(dy)); hz = fmaxf(hz, fabsf(dz));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        r = fmaxf(r, __shfl_xor(r, d, 64));
        rm = fmaxf(rm, __shfl_xor(rm, d, 64));
        neg = fmaxf(neg, __shfl_xor(neg, d, 64));
        hx = fmaxf(hx, __shfl_xor(hx, d, 64)); hy = fmaxf(hy, __shfl_xor(hy, d, 64)); hz = fmaxf(hz, __shfl_xor(hz, d, 64));
    }
    if (!PER_WAVE) {
        if (lane == 0) { s_red[wave][0] = r; s_red[wave][1] = rm; s_red[wave][2] = neg; s_red[wave][3] = hx; s_red[wave][4] = hy; s_red[wave][5] = hz; }
        __syncthreads();
    }
    if (rank == 0) {
        const float own[6] = {r, rm, neg, hx, hy, hz};
        auto top = [&](int k) { return PER_WAVE ? own[k] : fmaxf(fmaxf(s_red[0][k], s_red[1][k]), fmaxf(s_red[2][k], s_red[3][k])); };
        r = top(0); rm = top(1); neg = top(2);
        r = r * 1.00001f + 1.0e-30f;
        if (bad != 0.0f || !(r < 1.0e18f)) r = __builtin_inff();
        spheres[i] = make_float4(cx, cy, cz, r);
        rmax[i] = (neg != 0.0f || !(r < 1.0e18f)) ? __builtin_inff() : rm;
        // the differences are rounded once each (relative 2^-24): the factor covers it
        const float inf = __builtin_inff();
        const bool ok = r < 1.0e18f;
        boxes[i] = make_float4(ok ? top(3) * 1.00001f + 1.0e-30f : inf, ok ? top(4) * 1.00001f + 1.0e-30f : inf,
                               ok ? top(5) * 1.00001f + 1.0e-30f : inf, 0.0f);
    }
}

// ranges up to this many meshlets get a wave each (16 loads per lane and pass), longer ones a workgroup each
constexpr uint32_t kRangePerWaveMax = 1024;

constexpr uint32_t kWorkHeader = 64;  // uint32 words in front of a work list: [0] = number of entries (256-byte header)
// Shape of the pre-pass: one instance per thread, 1 024 threads per workgroup, one list reservation per workgroup.
// Every reservation is a returning atomic on the same word, and one word takes about 90 of them per microsecond, so few
// of them are wanted - but not at the price of few waves: 256 threads with 2 / 8 instances each (1 526 / 381 reservations
// for 781 250 instances) gave 0.531 / 0.525 ms per step with the frustum test alone and 0.250 / 0.265 ms with the occlusion
// test, whose threads wait for texels and want many waves in flight; 1 024 threads with one instance each (763
// reservations, four times the waves) gives 0.517 and 0.229 ms (2 or 4 instances per thread at 1 024, and 512 threads:
// 0.518 - 0.524 / 0.229 - 0.244 ms).
constexpr int kPreItems = 1, kPreBlock = 1024;

// Layout of a work list over `units` entries: header, then `units` uint32 (ranged scenes: the group's index
// in the scene; uniform scenes: the surviving instance's index), then - ranged scenes only - `units` uint4
// group records {instance, first meshlet, count, local group}, 16-byte aligned, so that the mask pass gets a
// listed group's record with one load instead of two dependent ones.
__host__ __device__ inline size_t work_rec_offset_words(uint64_t units) { return (kWorkHeader + units + 3) & ~size_t(3); }

// instance_occlusion, kInstCulled: does the occlusion test certainly reject every meshlet of the instance?
// A meshlet is rejected when  cz/cw < d - 0.05  for the texel d that its projected centre selects at the mip
// level its projected radius selects (occlusionCull, shader.task:58-77).  With W the instance sphere's world
// centre, every meshlet centre w_m (as evaluated in binary32) lies within rho of W, so each clip coordinate of
// w_m lies within e_k = |row k of VP| rho + roundoff of W's; rmax_w bounds the meshlets' world radii.  From
// those intervals (cw must stay positive):
//   depth:  cz/cw <= nz_ub;
//   level:  at most lmax, the level of the largest possible pixel radius (level 0 when proj[1][1] <= 0: the
//           radius is then never positive);
//   texel:  u, v inside [u_lo, u_hi] x [v_lo, v_hi], a sub-range of (0, 1): no REPEAT wrap.  Level l's texel
//           x = floor(u (w0 >> l)) is the minimum over the level-0 columns [x 2^l, (x+1) 2^l) (the 2x2 reduction
//           of floor-halved levels nests exactly), and these lie inside [u w0 - 2^l - (w0 mod 2^l), u w0 + 2^l].
// So every texel any meshlet can select is a minimum over level-0 texels inside a known rectangle, hence not
// below dmin, the minimum of the texels of one coarser level k that cover the rectangle (at most 4 x 4 reads;
// levels drop the odd last column and row of their source, so the rectangle must lie inside level k's cover).
// The instance is hidden when nz_ub < dmin - 0.05f.  That holds as long as no texel is a NaN: the reduction
// min(x, y) = y < x ? y : x can drop one, and the fine level would then answer "visible" where the coarse one
// does not.  Word 48 of a pyramid's level table says whether the depth image it was last built from held a NaN;
// the instance test is then off for that pyramid.  Anything non-finite or out of range among
// the intermediate values fails a comparison below, and the instance is not marked.
// The same intervals also prove the opposite for instances in the open (kInstOpen): every texel a meshlet can
// select is a minimum over texels of the rectangle, hence not ABOVE the largest depth of the 64 x 8 strips of the
// depth image that cover it (v.strips, at most 32 reads); when the smallest possible cz/cw is not below that
// maximum minus the bias, no meshlet of the instance is occluded and the per-meshlet pass skips the test.  (A NaN
// texel answers "not occluded" by itself, so this half does not depend on the NaN word.)
constexpr uint32_t kInstCulled = 1u;  // every meshlet is certainly rejected
constexpr uint32_t kInstInside = 2u;  // no meshlet is rejected by the frustum test
constexpr uint32_t kInstOpen = 4u;    // no meshlet is rejected by the occlusion test

// what instance_classify needs of an instance, whatever the view: world-space sphere centre, the bound F on the
// stretch of its matrix, the range's radius and largest meshlet radius, and the magnitudes the margins scale with
// BOX: also the world-space images of the box's half axes, b[i] = column i of the matrix times h_i, and hs = F (hx + hy + hz)
struct InstanceGeom {
    float wx, wy, wz, F, R, rmax_local, ax, ay, az;
    float b[3][3], hs;
};

// where launch_range_spheres / launch_group_spheres put the boxes of n ranges: behind the n spheres and n largest radii
__host__ __device__ inline size_t range_box_offset_bytes(uint32_t n) { return ((size_t)n * 20 + 15) & ~(size_t)15; }
__host__ __device__ inline const float4 *range_boxes(const float4 *spheres, uint32_t n) {
    return reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(spheres) + range_box_offset_bytes(n));
}

// how far the box reaches from its centre along (nx, ny, nz), not necessarily a unit vector: sum_i |n . b_i|.  Each dot
// product is off by a few 2^-24 of |n| |b_i| <= |n| F h_i; the caller adds |n| hs 2^-18 for that.
__device__ __forceinline__ float box_reach(const InstanceGeom &q, float nx, float ny, float nz) {
    return (fabsf((nx * q.b[0][0] + ny * q.b[0][1]) + nz * q.b[0][2]) + fabsf((nx * q.b[1][0] + ny * q.b[1][1]) + nz * q.b[1][2])) +
           fabsf((nx * q.b[2][0] + ny * q.b[2][1]) + nz * q.b[2][2]);
}

// BOX: geo also holds the range's box; a clip coordinate of a meshlet centre then differs from the centre's by at most the
// box's reach along that row of VP (box_reach), plus the same roundoff terms - whichever bound is smaller is used.
template <bool BOX>
__device__ __forceinline__ uint32_t instance_occlusion(const ViewConst &v, const InstanceGeom &geo, float rho, float rmax_w, bool want_open) {
    const float wx = geo.wx, wy = geo.wy, wz = geo.wz;
    float c[4], e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float m0 = v.vp[k], m1 = v.vp[4 + k], m2 = v.vp[8 + k], m3 = v.vp[12 + k];
        c[k] = ((m0 * wx + m1 * wy) + m2 * wz) + m3;
        const float len = v.vp_row_len[k];  // >= |row k of the linear part|
        const float mag = ((fabsf(m0) * (fabsf(wx) + rho) + fabsf(m1) * (fabsf(wy) + rho)) + fabsf(m2) * (fabsf(wz) + rho)) + fabsf(m3);
        e[k] = len * rho * 1.0001f + mag * 1.52587890625e-5f;  // 2^-16: a thousand times the roundoff of both evaluations
        if (BOX) {
            // centres as the per-meshlet path evaluates them: inside the box up to the roundoff of both evaluations
            const float round_w = ((geo.ax + geo.ay) + geo.az) * 1.52587890625e-5f;
            const float eb = (box_reach(geo, m0, m1, m2) * 1.0001f + len * (geo.hs * 3.814697265625e-6f + round_w)) + mag * 1.52587890625e-5f;
            e[k] = fminf(e[k], eb);  // a box that is not finite gives NaN or +inf here: the sphere's bound stays
        }
    }
    const float cw_lo = c[3] - e[3], cw_hi = c[3] + e[3];
    if (!(cw_lo > 1.0e-12f && cw_hi < 1.0e18f)) return 0u;
    // 1/cw of any meshlet lies between them (v_rcp_f32 is good to 1 ulp; the factors move the bounds outwards by 2^-20)
    const float r_lo = __builtin_amdgcn_rcpf(cw_hi) * 0.99999905f, r_hi = __builtin_amdgcn_rcpf(cw_lo) * 1.00000095f;
    auto upper = [&](float x) { return x >= 0.0f ? x * r_hi : x * r_lo; };  // of x / cw
    auto lower = [&](float x) { return x >= 0.0f ? x * r_lo : x * r_hi; };
    const float slack = 2.44140625e-4f;  // 2^-12, relative: the roundings of the per-meshlet quotients and sums
    const float q = upper(c[2] + e[2]);
    const float nz_ub = q + fabsf(q) * slack + 1.0e-30f;
    const float nx_lo = lower(c[0] - e[0]), nx_hi = upper(c[0] + e[0]);
    const float ny_lo = lower(c[1] - e[1]), ny_hi = upper(c[1] + e[1]);
    const float u_lo = (nx_lo * 0.5f + 0.5f) - (fabsf(nx_lo) + 1.0f) * slack, u_hi = (nx_hi * 0.5f + 0.5f) + (fabsf(nx_hi) + 1.0f) * slack;
    const float v_lo = (1.0f - (ny_hi * 0.5f + 0.5f)) - (fabsf(ny_hi) + 1.0f) * slack;
    const float v_hi = (1.0f - (ny_lo * 0.5f + 0.5f)) + (fabsf(ny_lo) + 1.0f) * slack;
    if (!(u_lo > 0.001f && u_hi < 0.999f && v_lo > 0.001f && v_hi < 0.999f)) return 0u;
    uint32_t lmax = 0u;
    if (!(v.proj11 <= 0.0f)) {
        const float px_ub = (((v.proj11 * rmax_w) * r_hi) * 2.0f) * v.tex_max * 1.001f;
        if (!(px_ub < 1.0e9f)) return 0u;
        lmax = mip_level(px_ub, v.levels);
    } else if (!(rmax_w < 1.0e18f)) {
        return 0u;
    }
    if ((v.w0 >> lmax) == 0u || (v.h0 >> lmax) == 0u || v.w0 > 32768u || v.h0 > 32768u) return 0u;
    // (w0 >> l) 2^l = w0 - (w0 mod 2^l), and w0 mod 2^l <= w0 mod 2^lmax for l <= lmax: the second 2^l of the lower bound
    // above is really that remainder - nothing for the usual image sizes, which halves the reach to the left and upwards
    const int32_t reach = 1 << lmax;
    const int32_t rem_x = (int32_t)(v.w0 & (uint32_t)(reach - 1)), rem_y = (int32_t)(v.h0 & (uint32_t)(reach - 1));
    int32_t x_lo = (int32_t)floorf(u_lo * (float)v.w0) - reach - rem_x - 2, x_hi = (int32_t)(u_hi * (float)v.w0) + reach + 2;
    int32_t y_lo = (int32_t)floorf(v_lo * (float)v.h0) - reach - rem_y - 2, y_hi = (int32_t)(v_hi * (float)v.h0) + reach + 2;
    x_lo = max(x_lo, 0); y_lo = max(y_lo, 0);
    x_hi = min(x_hi, (int32_t)v.w0 - 1); y_hi = min(y_hi, (int32_t)v.h0 - 1);
    // Both tables are read with a fixed number of independent loads (coordinates past the rectangle repeat its last
    // row / column), so that a thread waits for memory once per table, not once per texel.
    if (want_open && v.strips) {
        // up to 4 x 8 cells of the strips (64 x 8 pixels each), or - when the rectangle is larger than that and lies
        // inside what they cover - of the tables behind them: cells of 64, 128, 256 and 512 pixels square
        const uint32_t sw = v.strips_w, sh = (v.h0 + 7u) >> 3;
        int32_t cx0 = x_lo >> 6, ncx = (x_hi >> 6) - cx0 + 1;
        int32_t cy0 = y_lo >> 3, ncy = (y_hi >> 3) - cy0 + 1;
        const float *table = v.strips;
        uint32_t row_len = sw;
        if (ncx > 4 || ncy > 8) {
            const uint32_t cover_w = ((v.w0 >> 3) + 63u) / 64u * 512u, cover_h = ((v.h0 >> 3) + 63u) / 64u * 512u;
            ncx = ncy = 99;
            if ((uint32_t)x_hi < cover_w && (uint32_t)y_hi < cover_h) {
                size_t off = (size_t)sw * sh;
                uint32_t cw = sw, ch = (v.h0 + 63u) >> 6;
#pragma unroll
                for (uint32_t j = 0; j <= 3; j++) {
                    const int32_t nx = (x_hi >> (6 + j)) - (x_lo >> (6 + j)) + 1, ny = (y_hi >> (6 + j)) - (y_lo >> (6 + j)) + 1;
                    if (ncx > 4 && nx <= 4 && ny <= 8) {  // the finest table that will do
                        cx0 = x_lo >> (6 + j); cy0 = y_lo >> (6 + j); ncx = nx; ncy = ny;
                        table = v.strips + off;
                        row_len = cw;
                    }
                    off += (size_t)cw * ch;
                    cw = (cw + 1u) >> 1;
                    ch = (ch + 1u) >> 1;
                }
            }
        }
        if (ncx <= 4 && ncy <= 8) {
            const float ql = lower(c[2] - e[2]);
            const float nz_lb = (ql - fabsf(ql) * slack) - 1.0e-30f;
            float smax = -__builtin_inff();
            const float *row0 = table + (size_t)cy0 * row_len + cx0;
            // one row of cells per iteration, its (at most four) loads independent of each other: scattered 4-byte
            // loads are what this kernel waits for, so none is issued for rows the rectangle does not have
            const int32_t c1 = min(1, ncx - 1), c2 = min(2, ncx - 1), c3 = ncx - 1;
            for (int32_t j = 0; j < ncy; j++) {
                const float *row = row0 + (size_t)j * row_len;
                smax = fmaxf(fmaxf(smax, fmaxf(row[0], row[c1])), fmaxf(row[c2], row[c3]));
            }
            if (nz_lb >= smax - 0.05f) return kInstOpen;  // it cannot be hidden as well
        }
    }
    uint32_t k = 0;
    while (k < 16u && (((x_hi >> k) - (x_lo >> k)) > 3 || ((y_hi >> k) - (y_lo >> k)) > 3)) k++;
    const uint32_t lw = v.w0 >> k, lh = v.h0 >> k;
    if (v.level_table[48] != 0u) return 0u;
    // A rectangle that needs a level coarser than the window tables' cells: the tables' windows cover it within one cell
    // each way (see kHizWinBase), where 4 x 4 texels of level k may reach twice as far - and every level-0 texel they
    // reach beyond the rectangle that is background undoes the proof.
    if (v.windows && k > kHizWinBase && v.level_table[kHizWinWord] != 0u) {
        const uint32_t cw = v.w0 >> kHizWinBase, ch = v.h0 >> kHizWinBase;
        const int32_t X0 = x_lo >> kHizWinBase, X1 = x_hi >> kHizWinBase, Y0 = y_lo >> kHizWinBase, Y1 = y_hi >> kHizWinBase;
        const uint32_t W = (uint32_t)(X1 - X0 + 1), H = (uint32_t)(Y1 - Y0 + 1);
        uint32_t j = 31u - (uint32_t)__clz((int)min(W, H));  // W, H >= 1
        j = min(j, v.win_levels);
        const uint32_t win = 1u << j, nx = (W + win - 1u) >> j, ny = (H + win - 1u) >> j;
        if ((uint32_t)X1 < cw && (uint32_t)Y1 < ch && nx <= 4u && ny <= 4u) {
            const float *table = j == 0u ? v.pyramid + v.level_table[32 + kHizWinBase] : v.windows + (size_t)(j - 1u) * cw * ch;
            const int32_t xl = X1 - (int32_t)win + 1, yl = Y1 - (int32_t)win + 1;  // the last window's corner; >= X0, Y0
            float dmin = __builtin_inff();
#pragma unroll
            for (int32_t b = 0; b < 4; b++)
#pragma unroll
                for (int32_t a = 0; a < 4; a++)
                    dmin = fminf(dmin, table[(size_t)min(Y0 + b * (int32_t)win, yl) * cw + min(X0 + a * (int32_t)win, xl)]);
            return nz_ub < dmin - 0.05f ? kInstCulled : 0u;
        }
    }
    if (k < 16u && (uint32_t)(x_hi >> k) < lw && (uint32_t)(y_hi >> k) < lh) {
        const float *level = v.pyramid + v.level_table[32 + k];
        const int32_t x0 = x_lo >> k, x1 = x_hi >> k, y0 = y_lo >> k, y1 = y_hi >> k;
        float dmin = __builtin_inff();
#pragma unroll
        for (int32_t j = 0; j < 4; j++)
#pragma unroll
            for (int32_t i = 0; i < 4; i++) dmin = fminf(dmin, level[(size_t)min(y0 + j, y1) * lw + min(x0 + i, x1)]);
        if (nz_ub < dmin - 0.05f) return kInstCulled;
    }
    return 0u;
}

// instance_classify, kInstCulled: is the instance certainly outside one of the five planes?
// For every meshlet m of the instance the per-pair test evaluates, in binary32,
//     e_m = (dot(n, M c_m + t) + d) + r_m |M col 0|          (frustumCull, shader.task:43-56)
// and culls when e_m < 0 for some plane.  In real numbers e_m <= dot(n, M C + t) + d + max(|n|, 1) F R,
// where F bounds the largest singular value of M's linear part (and |col 0|) from above and
// R is the range's radius.  The binary32 evaluation of e_m differs from its real value by at most a few
// units of roundoff times A = sum_j |n_j| (sum_k |M_jk| (|C_k| + R) + |t_j|) + |d| + F R; so does the
// evaluation of the bound here.  The instance is marked only when  bound + 2^-16 A < 0, a margin 500
// times the worst case of both.  Any NaN or infinity among the inputs makes A or the bound non-finite
// and the comparison false, so such instances go through the per-meshlet path like the oracle's.
// What was tried for handing the verdicts to the mask pass (100 M pairs, 32 % of the instances outside, mask
// pass 484 us without any of it): one byte per group, the pass branching around a marked group's loads and
// tests - 433 us, and 513 us when nothing is marked; a list cut into segments of 2048 groups with static
// workgroup-to-segment mapping (ordered, no reservation) - 420 us, the workgroups behind a segment's count
// still start and leave; one compacted list with one reservation per workgroup of the pre-pass - 392 us,
// kept.  In each form the pre-pass itself takes 13 - 44 us.
// The same bound from the other side gives kInstInside: when (dot(n, W) + d) - reach - 2^-16 A > 0 for all five planes
// and no radius of the range is negative, e_m >= 0 for every meshlet and the per-meshlet pass skips the frustum test.
// want_more: also work out kInstInside / kInstOpen (the caller hands them on); otherwise only kInstCulled.
template <bool BOX = false>
__device__ __forceinline__ InstanceGeom sphere_geometry(const float4 *__restrict__ transforms, uint32_t i, float4 sp, float rmax_local,
                                                        float4 box = float4{0.0f, 0.0f, 0.0f, 0.0f}) {
    const float4 *t = transforms + (size_t)i * 4;
    const float4 c0 = t[0], c1 = t[1], c2 = t[2], c3 = t[3];
    const float R = sp.w;
    const float wx = ((c0.x * sp.x + c1.x * sp.y) + c2.x * sp.z) + c3.x;
    const float wy = ((c0.y * sp.x + c1.y * sp.y) + c2.y * sp.z) + c3.y;
    const float wz = ((c0.z * sp.x + c1.z * sp.y) + c2.z * sp.z) + c3.z;
    // F >= the largest singular value of M's linear part: Gershgorin's bound on the largest eigenvalue of
    // G = M^T M (max row sum of |G|), exact for the usual rotation x uniform scale (G diagonal), where it is
    // also |col 0|; never below |col 0| (G_00 is part of row 0's sum)
    const float g00 = (c0.x * c0.x + c0.y * c0.y) + c0.z * c0.z, g11 = (c1.x * c1.x + c1.y * c1.y) + c1.z * c1.z;
    const float g22 = (c2.x * c2.x + c2.y * c2.y) + c2.z * c2.z;
    const float g01 = fabsf((c0.x * c1.x + c0.y * c1.y) + c0.z * c1.z), g02 = fabsf((c0.x * c2.x + c0.y * c2.y) + c0.z * c2.z);
    const float g12 = fabsf((c1.x * c2.x + c1.y * c2.y) + c1.z * c2.z);
    // the products and sums above carry a few units of roundoff relative to |col i||col j| <= the trace
    const float slack = ((g00 + g11) + g22) * 1.0e-6f;
    // (the hardware square root, 1 ulp and denormals flushed: the factor and the absolute term cover both)
    const float F = __builtin_amdgcn_sqrtf(fmaxf(fmaxf((g00 + g01) + g02, (g01 + g11) + g12), (g02 + g12) + g22) + slack) * 1.00001f + 1.0e-18f;
    const float ex = fabsf(sp.x) + R, ey = fabsf(sp.y) + R, ez = fabsf(sp.z) + R;
    const float ax = ((fabsf(c0.x) * ex + fabsf(c1.x) * ey) + fabsf(c2.x) * ez) + fabsf(c3.x);
    const float ay = ((fabsf(c0.y) * ex + fabsf(c1.y) * ey) + fabsf(c2.y) * ez) + fabsf(c3.y);
    const float az = ((fabsf(c0.z) * ex + fabsf(c1.z) * ey) + fabsf(c2.z) * ez) + fabsf(c3.z);
    InstanceGeom q{wx, wy, wz, F, R, rmax_local, ax, ay, az, {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}}, 0.0f};
    if (BOX) {
        q.b[0][0] = c0.x * box.x; q.b[0][1] = c0.y * box.x; q.b[0][2] = c0.z * box.x;
        q.b[1][0] = c1.x * box.y; q.b[1][1] = c1.y * box.y; q.b[1][2] = c1.z * box.y;
        q.b[2][0] = c2.x * box.z; q.b[2][1] = c2.y * box.z; q.b[2][2] = c2.z * box.z;
        q.hs = F * ((box.x + box.y) + box.z);
    }
    return q;
}


__device__ __forceinline__ InstanceGeom instance_geometry(const float4 *__restrict__ transforms, const float4 *__restrict__ spheres,
                                                          const float *__restrict__ rmax, uint32_t per_instance_sphere, uint32_t i) {
    const uint32_t k = per_instance_sphere ? i : 0u;
    return sphere_geometry(transforms, i, spheres[k], rmax[k]);
}
// ... with the range's box (boxes = range_boxes(spheres, number of ranges))
__device__ __forceinline__ InstanceGeom instance_geometry_box(const float4 *__restrict__ transforms, const float4 *__restrict__ spheres,
                                                              const float *__restrict__ rmax, const float4 *__restrict__ boxes,
                                                              uint32_t per_instance_sphere, uint32_t i) {
    const uint32_t k = per_instance_sphere ? i : 0u;
    return sphere_geometry<true>(transforms, i, spheres[k], rmax[k], boxes[k]);
}

// tests: the tests to reason about (v.tests, or fewer when a coarser level has already settled some)
template <bool BOX = false>
__device__ __forceinline__ uint32_t classify_geometry(const InstanceGeom &q, const ViewConst &v, bool want_more, uint32_t tests) {
    const float wx = q.wx, wy = q.wy, wz = q.wz, F = q.F, R = q.R, rmax_local = q.rmax_local, ax = q.ax, ay = q.ay, az = q.az;
    bool outside = false, inside = rmax_local < 1.0e18f;  // +inf: a negative or non-finite radius in the range
    if (tests & MCULL_TEST_FRUSTUM) {
#pragma unroll
        for (int p = 0; p < 5; p++) {
            const float nx = v.frustum[p][0], ny = v.frustum[p][1], nz = v.frustum[p][2], d = v.frustum[p][3];
            const float nlen = v.plane_len[p];  // >= max(|n|, 1)
            const float reach_sphere = nlen * F * R;
            float reach = reach_sphere;
            if (BOX) {
                // centres: within the box's reach along n of W; radii: r_m |col 0| <= rmax F.  (Not finite: the sphere's.)
                const float rb = (box_reach(q, nx, ny, nz) * 1.0001f + nlen * q.hs * 3.814697265625e-6f) + (rmax_local * F) * 1.0001f;
                reach = fminf(reach_sphere, rb);
            }
            const float A = (((fabsf(nx) * ax + fabsf(ny) * ay) + fabsf(nz) * az) + fabsf(d)) + reach_sphere;
            const float base = ((nx * wx + ny * wy) + nz * wz) + d, margin = A * 1.52587890625e-5f;  // 2^-16
            outside |= (base + reach) + margin < 0.0f;
            inside = inside && (base - reach) - margin > 0.0f;
        }
    } else {
        inside = false;
    }
    if (outside) return kInstCulled;
    uint32_t flags = (want_more && inside) ? kInstInside : 0u;
    if ((tests & MCULL_TEST_HIZ) && v.pyramid) {
        // every meshlet centre, as the per-meshlet path evaluates it, lies within rho of (wx, wy, wz): F R in real
        // numbers, plus the roundoff of both evaluations (a few 2^-24 of ax + ay + az)
        const float rho = (F * R) * 1.0001f + ((ax + ay) + az) * 1.52587890625e-5f;
        const float rmax_w = (rmax_local * F) * 1.0001f;
        const uint32_t o = instance_occlusion<BOX>(v, q, rho, rmax_w, want_more);
        if (o & kInstCulled) return kInstCulled;
        flags |= o;
    }
    return flags;
}

__device__ __forceinline__ uint32_t instance_classify(const float4 *__restrict__ transforms, const float4 *__restrict__ spheres,
                                                      const float *__restrict__ rmax, uint32_t per_instance_sphere, uint32_t n_ranges,
                                                      uint32_t i, const ViewConst &v, bool want_more) {
    return classify_geometry<true>(instance_geometry_box(transforms, spheres, rmax, range_boxes(spheres, per_instance_sphere ? n_ranges : 1u),
                                                         per_instance_sphere, i), v, want_more, v.tests);
}

// The pre-pass of a single-view cull: the WORK LIST of the mask pass (layout above; work[0] is zero on entry).
// Each workgroup reserves room for the survivors among its 512 instances with one atomic, so the list is
// ordered inside a workgroup's share and in arrival order between workgroups; the mask pass does not depend
// on the order.  It never starts on a group that is not in the list: the ballots of those groups are zeroed
// here (zero_mask; scenes with few groups per instance) or by a memset beforehand.
// The multi-view pass has its own form of this level (classify_views_kernel): no list, one byte per instance and view.
__global__ __launch_bounds__(kPreBlock) void instance_precull_kernel(const float4 *__restrict__ transforms,
                                                                  const float4 *__restrict__ spheres, uint32_t per_instance_sphere,
                                                                  const uint2 *__restrict__ ranges,
                                                                  const uint32_t *__restrict__ group_base, uint32_t n_instances,
                                                                  uint32_t groups_per_instance, uint64_t units, ViewConst v_arg,
                                                                  const ViewConst *__restrict__ view_dev, const float *__restrict__ rmax,
                                                                  uint32_t *__restrict__ work, uint64_t *__restrict__ zero_mask,
                                                                  uint32_t want_flags, uint32_t *__restrict__ next_list) {
    const ViewConst &v = view_dev ? *view_dev : v_arg;
    __shared__ uint32_t s_wave[kPreBlock / 64];
    __shared__ uint32_t s_base;
    if (next_list && blockIdx.x == 0 && threadIdx.x == 0) next_list[0] = 0u;  // the group level's list starts empty
    const uint32_t i0 = (blockIdx.x * kPreBlock + threadIdx.x) * kPreItems;  // this thread's kPreItems consecutive instances
    bool outside[kPreItems];
    uint32_t hand_on[kPreItems];  // kInstInside / kInstOpen in bits 30 and 31 of the survivor's list entries (want_flags)
#pragma unroll
    for (int q = 0; q < kPreItems; q++) {
        const uint32_t f = i0 + q < n_instances ? instance_classify(transforms, spheres, rmax, per_instance_sphere, n_instances, i0 + q, v, want_flags != 0u)
                                               : kInstCulled;
        outside[q] = (f & kInstCulled) != 0u;
        hand_on[q] = want_flags ? (f & (kInstInside | kInstOpen)) << 29 : 0u;
    }
    uint32_t first[kPreItems], n[kPreItems], mine = 0;
#pragma unroll
    for (int q = 0; q < kPreItems; q++) {
        first[q] = i0 + q;
        n[q] = outside[q] ? 0u : 1u;
        if (group_base && i0 + q < n_instances) {
            first[q] = group_base[i0 + q];
            const uint32_t groups = group_base[i0 + q + 1] - first[q];
            if (n[q]) n[q] = groups;
            else if (zero_mask)  // a dropped instance: its groups' ballots are all zero
                for (uint32_t j = 0; j < groups; j++) zero_mask[first[q] + j] = 0ull;
        } else if (zero_mask && !n[q] && i0 + q < n_instances) {
            for (uint32_t j = 0; j < groups_per_instance; j++) zero_mask[(size_t)(i0 + q) * groups_per_instance + j] = 0ull;
        }
        mine += n[q];
    }
    uint32_t total;
    const uint32_t excl = block_exclusive_scan<kPreBlock / 64>(mine, s_wave, &total);
    if (threadIdx.x == 0) s_base = total ? atomicAdd(&work[0], total) : 0u;
    __syncthreads();
    uint32_t pos = s_base + excl;
    uint32_t *out_g = work + kWorkHeader;
    uint4 *out_rec = reinterpret_cast<uint4 *>(work + work_rec_offset_words(units));
#pragma unroll
    for (int q = 0; q < kPreItems; q++) {
        if (!n[q]) continue;
        if (group_base) {
            const uint2 r = ranges[i0 + q];
            for (uint32_t j = 0; j < n[q]; j++) {
                const uint32_t left = r.y - j * 64u;
                out_g[pos + j] = first[q] + j;
                out_rec[pos + j] = make_uint4(i0 + q, r.x + j * 64u, left < 64u ? left : 64u, j | hand_on[q]);
            }
        } else {
            out_g[pos] = first[q] | hand_on[q];
        }
        pos += n[q];
    }
}

// The GROUP level (no counterpart in the reference either): scenes whose instances span many 64-meshlet groups -
// one large mesh drawn many times - leave the instance level with survivors of which most groups are still hidden
// or outside: an instance that straddles an occluder's edge, or the frustum's, keeps all of its groups.  Every group
// that starts at meshlet m lies inside sphere m of a second table (range_sphere_kernel, sliding: the sphere of the
// meshlets [m, m + 64), which depends on the bounds alone), and the same two proofs run per listed group:
// what the instance level has settled for the whole instance (bits 30 / 31 of its entries) is not looked at again,
// a group that is certainly rejected leaves the list, and a group that lies inside all planes or in the open is told
// so.  in: the instance level's list; out: a list of the same layout for ranged scenes, and for uniform scenes one
// uint32 per group, the group's index in the scene with the two bits on top (hence fewer than 2^30 groups).
// A fixed number of workgroups walks the input (its length is device data); each reserves room for the survivors
// among its kBlock * kGrpItems entries with one atomic.  out[0] is zero on entry (instance_precull_kernel).
constexpr int kGrpItems = 4;

template <bool RANGED>
__global__ __launch_bounds__(kBlock) void group_precull_kernel(const float4 *__restrict__ transforms,
                                                               const float4 *__restrict__ gspheres, const float *__restrict__ grmax,
                                                               const float4 *__restrict__ gboxes,
                                                               uint32_t groups_per_instance, uint64_t units, ViewConst v_arg,
                                                               const ViewConst *__restrict__ view_dev,
                                                               const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                               uint32_t *__restrict__ in_report) {
    const ViewConst &v = view_dev ? *view_dev : v_arg;
    __shared__ uint32_t s_wave[kWaves];
    __shared__ uint32_t s_base;
    const uint32_t n_in = RANGED ? in[0] : in[0] * groups_per_instance;
    // how many groups this level was handed, left where the host can look at it later without waiting (host-mapped
    // memory): with what the mask pass reports of the list it got, that says whether this level earns its time
    if (in_report && blockIdx.x == 0 && threadIdx.x == 0) *in_report = n_in;
    const uint4 *in_rec = reinterpret_cast<const uint4 *>(in + work_rec_offset_words(units));
    uint4 *out_rec = reinterpret_cast<uint4 *>(out + work_rec_offset_words(units));
    constexpr uint32_t kPerBlock = kBlock * kGrpItems;
    for (uint32_t b = blockIdx.x; (unsigned long long)b * kPerBlock < n_in; b += gridDim.x) {
        const uint32_t i0 = b * kPerBlock + threadIdx.x * kGrpItems;
        uint32_t entry[kGrpItems], keep[kGrpItems], mine = 0;
        uint4 rec[kGrpItems];
#pragma unroll
        for (int q = 0; q < kGrpItems; q++) {
            keep[q] = 0u;
            entry[q] = 0u;
            rec[q] = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t idx = i0 + q;
            if (idx >= n_in) continue;
            uint32_t inst, settled, sphere, g;  // sphere: the group's first meshlet
            if (RANGED) {
                g = in[kWorkHeader + idx];
                rec[q] = in_rec[idx];
                inst = rec[q].x;
                settled = rec[q].w >> 30;
                sphere = rec[q].y;
            } else {
                const uint32_t item = idx / groups_per_instance, e = in[kWorkHeader + item];
                const uint32_t local_group = idx - item * groups_per_instance;
                inst = e & 0x3fffffffu;
                settled = e >> 30;
                sphere = local_group * 64u;
                g = inst * groups_per_instance + local_group;
            }
            // bit 0: no meshlet of the instance is rejected by the frustum test; bit 1: none by the occlusion test
            const uint32_t tests = v.tests & ~(((settled & 1u) ? (uint32_t)MCULL_TEST_FRUSTUM : 0u) | ((settled & 2u) ? (uint32_t)MCULL_TEST_HIZ : 0u));
            uint32_t f = 0u;
            if (tests & (MCULL_TEST_FRUSTUM | MCULL_TEST_HIZ))
                f = classify_geometry<true>(sphere_geometry<true>(transforms, inst, gspheres[sphere], grmax[sphere], gboxes[sphere]), v, true, tests);
            if (f & kInstCulled) continue;
            settled |= ((f & kInstInside) ? 1u : 0u) | ((f & kInstOpen) ? 2u : 0u);
            keep[q] = 1u;
            entry[q] = RANGED ? g : (g | (settled << 30));
            if (RANGED) rec[q].w = (rec[q].w & 0x3fffffffu) | (settled << 30);
            mine++;
        }
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(mine, s_wave, &total);
        if (threadIdx.x == 0) s_base = total ? atomicAdd(&out[0], total) : 0u;
        __syncthreads();
        uint32_t pos = s_base + excl;
#pragma unroll
        for (int q = 0; q < kGrpItems; q++) {
            if (!keep[q]) continue;
            out[kWorkHeader + pos] = entry[q];
            if (RANGED) out_rec[pos] = rec[q];
            pos++;
        }
        __syncthreads();  // s_base is rewritten by the next block
    }
}

// ------------------------------------------------------------------------------------------------
// ordered, pass 1: visibility ballots.  Small workgroups: each of the 4 waves takes NG consecutive
// groups.  The workgroup first stages the 4*NG group records in LDS, then requests every group's
// bounds (all loads unconditional and up front: NG x 2 KiB in flight per wave) while one wave
// stages the instance transforms in LDS with a coalesced load; after the barrier each wave reads
// its transforms back as LDS broadcasts, evaluates its groups and stores NG ballot words.  The
// workgroup's visible count is added to its 256-group tile's counter (one atomic per workgroup,
// 256/(4*NG) per counter; the counters are zero on entry and re-zeroed by emit_commands_kernel).
// Staging the transforms in LDS and fetching them with wave-uniform scalar loads into SGPRs
// measured the same within run-to-run noise (about 480 us for 100 M pairs); LDS staging keeps the
// SGPR budget small and replaces 4*NG dependent scalar load chains by one coalesced vector load.
// VP: the view constants are read from device memory (`view_dev`, written by an earlier kernel of the
// stream) instead of the kernel arguments, and the tests to run come from there too (TESTS is unused):
// a captured HIP graph can then be replayed with a camera that moves.
// FLAGS: the work list's entries carry kInstInside / kInstOpen in bits 30 and 31 (the pre-pass was asked for them):
// the groups of such an instance run without the frustum / occlusion test.  The tests are then a run-time value
// (v.tests, as with VP), narrowed per group.
// PK (streamed bounds only): two groups per step on packed operations (pair_visible2_impl) - or, PK = false, one group
// per step like the cache-resident form.  The packed step skips a test only when neither of its groups needs it, which
// costs more than the halved multiply/add count saves when the groups come from a work list without the occlusion test
// (three in four of their pairs are visible; 100 M pairs 0.516 -> 0.509 ms); the whole scene (0.575 against 0.592 ms)
// and lists with the occlusion test (0.219 against 0.225 ms) keep the packed step.
template <uint32_t TESTS, int NG, bool NT, bool VP = false, bool FLAGS = false, bool PK = true>
__global__ __launch_bounds__(kBlock) void cull_mask_kernel(
    const float4 *__restrict__ cone_axis_w, const float4 *__restrict__ centre_radius,
    const float4 *__restrict__ transforms, const float *__restrict__ xf_scale,
    const uint4 *__restrict__ group_rec, uint32_t groups_per_instance, uint32_t n_groups, ViewConst v_arg,
    uint64_t *__restrict__ mask, uint32_t *__restrict__ tile_counts, uint32_t xcd_contiguous,
    const ViewConst *__restrict__ view_dev, const uint32_t *__restrict__ work, uint32_t *__restrict__ kept_report,
    uint32_t list_of_groups, uint32_t start_block) {
    const ViewConst &v = VP ? *view_dev : v_arg;
    const uint32_t tests = (VP || FLAGS) ? v.tests : TESTS;
    // FLAGS only: the list of a scene without per-instance ranges holds one entry per group (group_precull_kernel), not
    // one per instance
    const bool per_group = FLAGS && list_of_groups != 0u;
    constexpr uint32_t kGroups = kWaves * NG;  // groups per workgroup
    __shared__ uint4 s_rec[kGroups];
    __shared__ uint32_t s_keep[kGroups];  // FLAGS: the tests each slot's group still needs
    __shared__ float4 s_x[kGroups][4];
    __shared__ float s_scale[kGroups];
    __shared__ uint32_t s_g[kGroups];    // index of each slot's group in the scene (its ballot word, its tile)
    __shared__ uint32_t s_pop[kGroups];  // visible pairs of each slot's group
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    // Workgroups are dealt to the 8 XCDs round-robin by id.  xcd_contiguous != 0 renumbers them so that
    // each XCD walks one contiguous eighth of the scene (the grid is then a multiple of 8).
    uint32_t first_block = blockIdx.x;
    if (xcd_contiguous) first_block = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (FLAGS && kept_report && blockIdx.x == 0 && threadIdx.x == 0) *kept_report = work[0];  // also when the list is empty
    // The FLAGS form is the one for cache-resident scenes behind a work list, where whole scenes' worth of
    // workgroups would start only to find the list shorter than the grid (1 M instances of one mesh, seven in eight
    // dropped: 3 ms of a 11 ms pass; everything dropped: 3.3 ms for nothing).  A fixed number of workgroups walks
    // the list with a stride instead.  (Streamed scenes keep one workgroup per 16 groups: a walking loop issues the
    // next block's loads only after the last one's arithmetic, 4.9 against 6.4 TB/s.)  Every other form runs the
    // loop's body exactly once.
    uint32_t n_list = 0;
    if constexpr (FLAGS) n_list = (group_rec || per_group) ? work[0] : work[0] * groups_per_instance;
    for (uint32_t b = FLAGS ? start_block + blockIdx.x : first_block;; b += gridDim.x) {  // start_block: the walk begins there (FLAGS)
    if (FLAGS && (unsigned long long)b * kGroups >= n_list) break;
    const uint32_t block_g0 = b * kGroups;
    // With a work list (instance_precull_kernel) the pass runs over the list's groups only: the grid is
    // sized for the whole scene and the workgroups beyond the list's end leave at once.
    // The list's length is device data.  So that no workgroup waits for it before it can ask for its records,
    // the slots read the list speculatively - at positions clamped to the list's capacity, which may hold
    // entries of an earlier cull - and what lies past the end is thrown away once the length has arrived,
    // before anything is addressed with it.
    if (block_g0 >= n_groups) return;  // the padding workgroups of a rounded-up grid
    uint32_t g = 0, hand_on = 0;
    GroupInfo gi = {0u, 0u, 0u, 0u};
    const uint32_t w = block_g0 + threadIdx.x;
    if (threadIdx.x < kGroups) {
        const uint32_t wc = w < n_groups ? w : n_groups - 1u;  // slots past the end re-read a valid element and report nothing
        g = wc;
        if (work && group_rec) {  // the list carries the record beside the group's index: two independent loads
            g = work[kWorkHeader + wc];
            const uint4 r = reinterpret_cast<const uint4 *>(work + work_rec_offset_words(n_groups))[wc];
            gi.inst = r.x; gi.first_meshlet = r.y; gi.count = r.z; gi.local_group = r.w;
            if (FLAGS) { hand_on = r.w >> 30; gi.local_group = r.w & 0x3fffffffu; }
        } else {
            if (work && per_group) {
                const uint32_t e = work[kWorkHeader + wc];
                hand_on = e >> 30;
                g = e & 0x3fffffffu;
            } else if (work) {
                const uint32_t item = wc / groups_per_instance;
                uint32_t e = work[kWorkHeader + item];
                if (FLAGS) { hand_on = e >> 30; e &= 0x3fffffffu; }
                g = e * groups_per_instance + (wc - item * groups_per_instance);
            }
            gi = group_lookup(g, group_rec, groups_per_instance, v.uniform_count);
        }
    }
    uint32_t n_run = n_groups;
    if (work) n_run = (group_rec || per_group) ? work[0] : work[0] * groups_per_instance;
    // how much the pre-cull kept, left where the host can look at it later without waiting (host-mapped memory)
    if (!FLAGS && work && kept_report && blockIdx.x == 0 && threadIdx.x == 0) *kept_report = work[0];
    if (block_g0 >= n_run) return;  // workgroups beyond the list's end
    if (threadIdx.x < kGroups) {
        if (w >= n_run) {  // nothing of a slot past the end is used: it re-reads element 0 and reports nothing
            g = 0u;
            gi = GroupInfo{0u, 0u, 0u, 0u};
            hand_on = 0u;
        }
        s_rec[threadIdx.x] = make_uint4(gi.inst, gi.first_meshlet, gi.count, gi.local_group);
        s_g[threadIdx.x] = g;
        if (FLAGS) s_keep[threadIdx.x] = ~(((hand_on & 1u) ? MCULL_TEST_FRUSTUM : 0u) | ((hand_on & 2u) ? MCULL_TEST_HIZ : 0u));
    }
    __syncthreads();
    if (threadIdx.x < kGroups * 4) {
        const uint32_t k = threadIdx.x >> 2, j = threadIdx.x & 3u;
        s_x[k][j] = transforms[(size_t)s_rec[k].x * 4 + j];
        if (j == 0) s_scale[k] = xf_scale[s_rec[k].x];
    }
    float4 a[NG], c[NG];
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const uint4 r = s_rec[wave * NG + k];
        const uint32_t last = r.z ? r.z - 1u : 0u;
        const uint32_t id = r.y + (lane < last ? lane : last);
        a[k] = load_bounds<NT>(cone_axis_w + id);
        c[k] = load_bounds<NT>(centre_radius + id);
    }
    __syncthreads();
    auto load_xform_lds = [&](uint32_t slot) {
        Xform m;
        m.c0 = s_x[slot][0]; m.c1 = s_x[slot][1]; m.c2 = s_x[slot][2]; m.c3 = s_x[slot][3];
        return m;
    };
    auto publish = [&](uint32_t slot, bool vis) {
        const uint32_t n_valid = s_rec[slot].z;
        const uint64_t ballot = __ballot(vis && (lane < n_valid));
        if (lane == 0) {
            if (n_valid) mask[s_g[slot]] = ballot;
            s_pop[slot] = (uint32_t)__popcll(ballot);
        }
    };
    if constexpr (NG % 2 == 0 && NT && PK) {
#pragma unroll
        for (int k = 0; k < NG; k += 2) {  // two groups per step: one pair per vector element
            const uint32_t slot = wave * NG + k;
            bool vis0, vis1;
            uint32_t t2 = tests;  // both groups of the step must be able to do without a test
            if (FLAGS) t2 &= __builtin_amdgcn_readfirstlane(s_keep[slot] | s_keep[slot + 1]);
            pair_visible2_impl<NT>(t2, v, load_xform_lds(slot), load_xform_lds(slot + 1), s_scale[slot], s_scale[slot + 1],
                                   a[k], a[k + 1], c[k], c[k + 1], lane < s_rec[slot].z, lane < s_rec[slot + 1].z, vis0, vis1);
            publish(slot, vis0);
            publish(slot + 1, vis1);
            // The compiler moves the cone loads of the first step into the cone test's block, which a wave skips
            // when the frustum test leaves it nothing: 100 M pairs then move 3.03 GB instead of 3.28 GB (counted:
            // 23.66 M 128-byte requests per launch) in 484 us.  Pinned in front of the barrier with the other
            // loads (an unconditional use here) the pass took 512 us: it is bound by bytes, not by that round trip.
        }
    } else {
#pragma unroll
        for (int k = 0; k < NG; k++) {
            const uint32_t slot = wave * NG + k;
            uint32_t t1 = tests;
            if (FLAGS) t1 &= __builtin_amdgcn_readfirstlane(s_keep[slot]);
            publish(slot, pair_visible_impl<NT>(t1, v, load_xform_lds(slot), s_scale[slot], a[k], c[k]));
        }
    }
    __syncthreads();
    // One atomic per run of slots that share a 256-group tile: without a work list that is one per
    // workgroup (16 consecutive groups never straddle a tile), with one a handful at most.
    if (threadIdx.x < kGroups) {
        const uint32_t k = threadIdx.x, tile = s_g[k] / kTileGroups;
        if (k == 0 || s_g[k - 1] / kTileGroups != tile) {
            uint32_t sum = 0;
            for (uint32_t j = k; j < kGroups && s_g[j] / kTileGroups == tile; j++) sum += s_pop[j];
            if (sum) atomicAdd(&tile_counts[tile], sum);
        }
    }
    if (!FLAGS) break;
    __syncthreads();  // the staging arrays are rewritten by the next block
    }
}

// ------------------------------------------------------------------------------------------------
// The same pass for several views at once: the bounds, records and transforms are read once and every
// view's tests run on them (the 32 bytes per pair are the dominant traffic of a streamed scene, so
// eight views cost one read instead of eight).  Each view has its own mask and tile counters.
// The views are taken two at a time (ViewHot2: element e of every pair of constants belongs to view 2p + e), so
// that the frustum and cone tests of both run on packed binary32 instructions - two independent IEEE operations
// per instruction, and unlike two GROUPS per step (tests2) a step of two VIEWS gives up nothing: which tests a
// view still needs is known per group and view, and an instruction that serves one view costs what it costs for
// two.  A pair's constants are fetched once per wave, in one batch of scalar loads in front of the groups that
// use them (see pin_sgpr), not per group and per use.
// One byte per instance and view slot, written by the instance level for the mask pass: bit 3 = the instance is still
// in the running for the view (clear: every meshlet is rejected, or the slot is unused), bits 0-2 = the MCULL_TEST_*
// its meshlets still need there.  f = the instance level's verdict (kInst*), tests = the view's tests.
__host__ __device__ inline uint32_t view_need_byte(uint32_t f, uint32_t tests) {
    if (f & kInstCulled) return 0u;
    return 8u | (tests & 7u & ~(((f & kInstInside) ? (uint32_t)MCULL_TEST_FRUSTUM : 0u) | ((f & kInstOpen) ? (uint32_t)MCULL_TEST_HIZ : 0u)));
}

// What is the same for every group of a multi-view cull, worked out once (on the host, or by pack_view_pairs_kernel).
struct ViewsUniform {
    uint32_t def_lo, def_hi;  // the need bytes (view_need_byte) of view slots 0-3 / 4-7 when no instance level has run
    uint32_t exact_all;       // bit i: view i wants every pair through the IEEE expressions
    uint32_t mixed;           // the views in use were not all asked for the same tests, or exact_all != 0
};

__host__ __device__ inline ViewsUniform views_uniform(uint32_t n_views, const uint32_t *tests, const uint32_t *force_exact) {
    ViewsUniform u = {0u, 0u, 0u, 0u};
    for (uint32_t i = 0; i < n_views; i++) {
        const uint32_t b = 8u | (tests[i] & 7u);
        if (i < 4u) u.def_lo |= b << (8u * i);
        else u.def_hi |= b << (8u * (i - 4u));
        if (force_exact[i]) u.exact_all |= 1u << i;
        if ((tests[i] & 7u) != (tests[0] & 7u)) u.mixed = 1u;
    }
    if (u.exact_all) u.mixed = 1u;
    return u;
}

struct ViewSet {
    ViewConst v[MCULL_MAX_VIEWS];
    ViewHot2 hot[MCULL_MAX_VIEWS / 2];
    ViewsUniform uni;
    uint64_t *mask[MCULL_MAX_VIEWS];
    uint32_t *tile_counts[MCULL_MAX_VIEWS];
    uint32_t n;
};

void fill_view_pairs(uint32_t n_views, const ViewConst *const *views, ViewHot2 *hot) {
    std::memset(hot, 0, sizeof(ViewHot2) * (MCULL_MAX_VIEWS / 2));
    for (uint32_t i = 0; i < n_views; i++) {
        ViewHot2 &h = hot[i >> 1];
        const uint32_t e = i & 1u;
        for (int p = 0; p < 5; p++)
            for (int j = 0; j < 4; j++) h.frustum[p][j][e] = views[i]->frustum[p][j];
        for (int j = 0; j < 3; j++) h.cam[j][e] = views[i]->cam[j];
        h.tests[e] = views[i]->tests;
        h.force_exact[e] = views[i]->force_exact;
    }
}

ViewsUniform fill_views_uniform(uint32_t n_views, const ViewConst *const *views) {
    uint32_t tests[MCULL_MAX_VIEWS] = {}, exact[MCULL_MAX_VIEWS] = {};
    for (uint32_t i = 0; i < n_views; i++) { tests[i] = views[i]->tests; exact[i] = views[i]->force_exact; }
    return views_uniform(n_views, tests, exact);
}

// The same from device-resident view constants (the *_device_globals calls): one thread per view slot.
__global__ void pack_view_pairs_kernel(const ViewConst *__restrict__ view_dev, uint32_t n_views, ViewHot2 *__restrict__ hot) {
    const uint32_t i = threadIdx.x;
    if (i >= MCULL_MAX_VIEWS) return;
    ViewHot2 &h = hot[i >> 1];
    const uint32_t e = i & 1u;
    const bool real = i < n_views;
    for (int p = 0; p < 5; p++)
        for (int j = 0; j < 4; j++) h.frustum[p][j][e] = real ? view_dev[i].frustum[p][j] : 0.0f;
    for (int j = 0; j < 3; j++) h.cam[j][e] = real ? view_dev[i].cam[j] : 0.0f;
    h.tests[e] = real ? view_dev[i].tests : 0u;
    h.force_exact[e] = real ? view_dev[i].force_exact : 0u;
    if (i == 0) {  // the uniforms go behind the pairs
        uint32_t tests[MCULL_MAX_VIEWS], exact[MCULL_MAX_VIEWS];
        for (uint32_t q = 0; q < MCULL_MAX_VIEWS; q++) { tests[q] = q < n_views ? view_dev[q].tests : 0u; exact[q] = q < n_views ? view_dev[q].force_exact : 0u; }
        *reinterpret_cast<ViewsUniform *>(hot + MCULL_MAX_VIEWS / 2) = views_uniform(n_views, tests, exact);
    }
}

hipError_t launch_pack_view_pairs(const ViewConst *view_dev, uint32_t n_views, ViewHot2 *hot, hipStream_t st) {
    hipLaunchKernelGGL(pack_view_pairs_kernel, dim3(1), dim3(64), 0, st, view_dev, n_views, hot);
    return hipGetLastError();
}

// The instance level of the multi-view pass: one thread per instance, which works out what does not depend on the view
// once and then instance_classify's verdict for every view (the view's constants are wave-uniform: scalar loads).
// flags[instance][slot] = what the mask pass still has to do for the instance in that view (view_need_byte; zero for
// unused slots), eight bytes stored at once; the mask pass stages them with the instance's group record.  (One thread per instance AND
// view read the view constants per lane and repeated the instance's part eight times: 293 us for 781 250 instances
// and 8 views.  A byte that only says "outside the frustum" did not pay in front of a pass that waited for its texel
// reads: 5.35 against 5.14 ms for 8 views at 8192^2.  With the occlusion verdicts and the tests it lets the pass leave
// out, it does.)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) void classify_views_kernel(const float4 *__restrict__ transforms, const float4 *__restrict__ spheres,
                                                                const float *__restrict__ rmax, uint32_t per_instance_sphere,
                                                                uint32_t n_instances, ViewSet vs, const ViewConst *__restrict__ view_dev,
                                                                uint2 *__restrict__ flags, uint32_t *__restrict__ useful) {
    __shared__ uint32_t s_n[kWaves];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t lo = 0, hi = 0, found = 0;
    if (i < n_instances) {
        const InstanceGeom q = instance_geometry(transforms, spheres, rmax, per_instance_sphere, i);
        for (uint32_t slot = 0; slot < vs.n; slot++) {
            const ViewConst &v = view_dev ? view_dev[slot] : vs.v[slot];
            const uint32_t f = classify_geometry(q, v, true, v.tests);
            found += f != 0u;
            const uint32_t need = view_need_byte(f, v.tests);
            if (slot < 4u) lo |= need << (8u * slot);
            else hi |= need << (8u * (slot - 4u));
        }
        flags[i] = make_uint2(lo, hi);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) found += __shfl_xor(found, d, 64);
    if ((threadIdx.x & 63u) == 0u) s_n[threadIdx.x >> 6] = found;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t sum = (s_n[0] + s_n[1]) + (s_n[2] + s_n[3]);
        if (sum) atomicAdd(useful, sum);
    }
}

// A wave-uniform value is wanted in scalar registers HERE: the empty statement makes the compiler finish the
// scalar loads that produce it at this point - as one batch with one wait, when several values are pinned in a
// row - instead of sinking each load into the conditional block that first uses it, where every wave that gets
// there waits for it on its own.
template <typename T>
__device__ __forceinline__ void pin_sgpr(T &x) {
    asm volatile("" : "+s"(x));
}

__device__ __forceinline__ f2v fma2(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }

// frustumCull and coneCull (shader.task:37-56) of one pair against two views; on entry k0 / k1 say that the pair is
// out of the running for view 0 / 1 already (its lane holds no meshlet, or the instance level has dropped the
// instance for that view), t0 / t1 are the tests each view still needs (wave-uniform).  w = world-space centre and
// radius, wa = unit world-space axis and coneW (world_centre, world_axis: IEEE, once for all views).
// The frustum test is the IEEE expression itself on packed operations.  The cone test is guarded-exact like
// cone_culled: its fast form is  dot(axis, v) * rsq(dot(v, v))  with fused multiply-adds, which lies within 2.5e-6
// of the IEEE expression's value when both are ordinary numbers (each within about 1e-6 of the real quotient: a dozen
// roundings of terms of magnitude <= 1 on one side, five and two units of the reciprocal square root on the other), so a
// margin beyond 1e-5 decides; the other lanes evaluate the IEEE expression.
// Lane masks instead of per-lane booleans: which pairs are still in the running is kept as one 64-bit scalar per
// view, comparisons produce ballots, and the little logic between them runs on the scalar unit.
__device__ __forceinline__ bool lane_of(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// v[lane L] = s, for a compile-time L: one instruction, no comparison and no change of the execution mask
template <int L>
__device__ __forceinline__ void set_lane(uint32_t &v, uint32_t s) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(L));
}

// What the cone test needs of one pair, whatever the view: the world-space axis normalised with v_rsq_f32 (its error
// is part of the fast form's bound), coneW, and -coneW^2 - a NaN when the fast form does not apply to the pair (coneW
// negative, huge or a NaN, or an axis that cannot be normalised), which makes every fast comparison below fail.
// The IEEE axis is evaluated where a lane needs it.
struct ConePair {
    float ax, ay, az, cw, ncw2;
};

__device__ __forceinline__ ConePair cone_pair(const Xform &m, float4 a) {
    const float nx = __builtin_fmaf(m.c2.x, a.z, __builtin_fmaf(m.c1.x, a.y, m.c0.x * a.x));
    const float ny = __builtin_fmaf(m.c2.y, a.z, __builtin_fmaf(m.c1.y, a.y, m.c0.y * a.x));
    const float nz = __builtin_fmaf(m.c2.z, a.z, __builtin_fmaf(m.c1.z, a.y, m.c0.z * a.x));
    const float la = __builtin_fmaf(nz, nz, __builtin_fmaf(ny, ny, nx * nx));
    const float ia = __builtin_amdgcn_rsqf(la);
    ConePair c;
    c.ax = nx * ia; c.ay = ny * ia; c.az = nz * ia;
    c.cw = a.w;
    c.ncw2 = (ordinary(la) && a.w >= 0.0f && a.w < 1.0e9f) ? -(a.w * a.w) : __builtin_nanf("");
    return c;
}

// normalize(mat3(model) * cone.xyz) (shader.task:95) from memory, for the rare lanes that need the IEEE value after
// the registers that held the matrix and the axis have gone to other uses
__device__ __noinline__ float4 world_axis_again(const float4 *__restrict__ t, const float4 *__restrict__ axis) {
    Xform m;
    m.c0 = t[0]; m.c1 = t[1]; m.c2 = t[2]; m.c3 = t[3];
    return world_axis(m, *axis);
}

// coneCull as written (shader.task:37-41), from the unit world-space axis wa = {axis, coneW}, the world-space centre w and
// the camera position: the IEEE expression.  Not inlined: it is the rare path of the cone test below.
__device__ __noinline__ bool cone_exact(float4 wa, float4 w, float camx, float camy, float camz) {
    const float vx = camx - w.x, vy = camy - w.y, vz = camz - w.z;
    const float ie = 1.0f / sqrtf(dot3(vx, vy, vz, vx, vy, vz));
    return dot3(wa.x, wa.y, wa.z, vx * ie, vy * ie, vz * ie) < -wa.w;
}

// frustumCull (shader.task:43-56) of one pair against two views: the IEEE expression itself on packed operations.
// Returns the lanes culled by view 0 / view 1.  The forty plane coefficients come in through scalar loads here, where
// a group needs them (the instance level usually has settled the frustum test for the whole instance).
__device__ __forceinline__ void frustum2(const ViewHot2 *__restrict__ hot, float4 w, uint64_t &c0, uint64_t &c1) {
    const f2v wx = splat(w.x), wy = splat(w.y), wz = splat(w.z), r = splat(w.w);
    f2v dmin = splat(__builtin_inff());
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const f2v nx = {hot->frustum[i][0][0], hot->frustum[i][0][1]}, ny = {hot->frustum[i][1][0], hot->frustum[i][1][1]};
        const f2v nz = {hot->frustum[i][2][0], hot->frustum[i][2][1]}, nw = {hot->frustum[i][3][0], hot->frustum[i][3][1]};
        const f2v d = (dot3v(nx, ny, nz, wx, wy, wz) + nw) + r;
        // "some d < 0" as "the smallest d < 0": fminf skips a NaN, and a NaN d culls nothing either
        dmin.x = fminf(dmin.x, d.x);
        dmin.y = fminf(dmin.y, d.y);
    }
    c0 = __ballot(dmin.x < 0.0f);
    c1 = __ballot(dmin.y < 0.0f);
}

// coneCull (shader.task:37-41) of one pair against two views, guarded-exact:
//   culled  <=>  d < -coneW,  d = dot(axis, v) / |v|,  v = camera - centre.
// Fast form, without a square root: with p = dot(axis, v), L = dot(v, v) and coneW >= 0,
//   d < -coneW  <=>  p < 0 and p^2 > coneW^2 L,          u = p^2 - coneW^2 L = L (d - coneW)(d + coneW),
// and since |d - coneW| <= 2,  |u| > 2.4e-5 L  implies  |d + coneW| > 1.2e-5.  The fast p, L and u (fused
// multiply-adds, the axis normalised with v_rsq_f32) lie within 1e-6 L of their real values and the IEEE
// expression's d within 2.5e-6 of its real value, so beyond that margin both decide alike.  Lanes inside it, and lanes
// whose numbers are not ordinary - L outside [1e-30, 1e30] (its bit pattern is range-checked as an integer), a NaN
// anywhere, a pair whose -coneW^2 is the NaN of cone_pair - fail a comparison and are returned in slow0 / slow1 for the
// IEEE expression.  c0 / c1: the lanes the fast form culls.
__device__ __forceinline__ void cone2_fast(f2v camx, f2v camy, f2v camz, float4 w, const ConePair &cp, uint64_t &c0, uint64_t &c1,
                                           uint64_t &sure0, uint64_t &sure1) {
    const f2v vx = camx - splat(w.x), vy = camy - splat(w.y), vz = camz - splat(w.z);
    const f2v lv = fma2(vz, vz, fma2(vy, vy, vx * vx));
    const f2v p = fma2(splat(cp.az), vz, fma2(splat(cp.ay), vy, splat(cp.ax) * vx));
    const f2v u = fma2(splat(cp.ncw2), lv, p * p);
    const f2v thr = lv * splat(2.4e-5f);
    // min(-p, u) > 0  <=>  p < 0 and u > 0 (a NaN compares false)
    c0 = __ballot(fminf(-p.x, u.x) > 0.0f);
    c1 = __ballot(fminf(-p.y, u.y) > 0.0f);
    constexpr uint32_t kLo = 0x0DA24260u, kSpan = 0x7149F2CAu - 0x0DA24260u;  // bit patterns of 1e-30f and 1e30f
    const bool r0 = __float_as_uint(lv.x) - kLo < kSpan, r1 = __float_as_uint(lv.y) - kLo < kSpan;
    sure0 = __ballot(fabsf(u.x) > (r0 ? thr.x : __builtin_inff()));
    sure1 = __ballot(fabsf(u.y) > (r1 ? thr.y : __builtin_inff()));
}

// occlusionCull of one pair against one view (the clip-space centre is the IEEE expression, as in cone_and_occlusion)
__device__ __forceinline__ bool occlusion1(const ViewConst &v, float4 w, bool live) {
    const float cx = ((v.vp[0] * w.x + v.vp[4] * w.y) + v.vp[8] * w.z) + v.vp[12];
    const float cy = ((v.vp[1] * w.x + v.vp[5] * w.y) + v.vp[9] * w.z) + v.vp[13];
    const float cz = ((v.vp[2] * w.x + v.vp[6] * w.y) + v.vp[10] * w.z) + v.vp[14];
    const float cw = ((v.vp[3] * w.x + v.vp[7] * w.y) + v.vp[11] * w.z) + v.vp[15];
    return occluded(v, cx, cy, cz, cw, v.proj11 * w.w, live);
}

// The multi-view mask pass.  It is bound by instruction issue, vector and scalar alike (a SIMD issues one of each per
// turn, from different waves: measured, per-wave time = 8 waves x the larger of the two counts), so it is written to
// spend few of either:
//  * lane masks in scalar registers instead of per-lane booleans, comparisons as ballots;
//  * the instance level's verdicts arrive as ready-made "tests still needed" bytes (view_need_byte);
//  * the views are taken two at a time on packed operations (ViewHot2), the four pairs unrolled so that every shift,
//    lane index and constant address is an immediate;
//  * the cone test's fast form needs no square root and five comparisons per view (cone2_fast);
//  * the ballots of a wave - one per group and view - are gathered in the lanes of two registers (lane = view * NG +
//    group, v_writelane), go to LDS with one instruction per wave, and leave the workgroup with ONE store
//    instruction, each view's words contiguous; their popcounts are taken from those lanes at the end.
// The occlusion test runs last, after the frustum and cone tests of all views, on the gathered ballots: its texel read
// is the one data-dependent memory access, and the fewer pairs reach it the better.
// VP: the views' constants are read from view_dev[i] / hot_dev[i / 2] (device memory, written by earlier kernels of the
// stream) instead of the argument segment; masks, counters and the uniform meshlet count still come from `vs`.
#ifndef MCULL_VIEWS_NG
#define MCULL_VIEWS_NG 2
#endif
constexpr int kViewsNG = MCULL_VIEWS_NG;  // groups per wave of the multi-view pass

// What a wave holds of one of its groups while the views go by.
struct ViewsGroup {
    float4 wc;        // world-space centre and radius (world_centre: IEEE)
    ConePair cp;
    uint64_t in;      // lanes that hold a meshlet
    uint32_t nb_lo, nb_hi;  // the instance's need bytes for view slots 0-3 / 4-7
};

// Frustum and cone tests of group K of the wave against the views 2P and 2P + 1; the survivors' ballots go to lanes
// (2P) * NG + K and (2P + 1) * NG + K of blo / bhi.
// A test the instance level has settled for the instance ("no meshlet is rejected") may still be run on its meshlets -
// it rejects none - so a block that runs for one view of the pair applies its result to both, unless the views were not
// ASKED for the same tests or one of them wants the IEEE expressions throughout (`mixed`, rare: then every step looks
// at the views' own bits).
template <int P, int K, int NG, typename AxisFn>
__device__ __forceinline__ void views_pair_step(const ViewsGroup &g, const ViewHot2 *__restrict__ hot, const f2v *__restrict__ cam,
                                                uint32_t mixed, uint32_t asked, uint32_t exact_all,
                                                const ViewConst *__restrict__ views, AxisFn &&exact_axis, uint32_t &blo, uint32_t &bhi) {
    const uint32_t nb = (P < 2 ? g.nb_lo : g.nb_hi) >> (16 * (P & 1));  // bytes 2P and 2P + 1 of the instance's eight
    uint64_t live0 = (nb & 8u) ? g.in : 0ull, live1 = (nb & 0x800u) ? g.in : 0ull;
    if (live0 | live1) {
        const uint32_t ask = mixed ? asked >> (16 * (P & 1)) : nb | (nb >> 8) | (nb << 8);  // the tests each view may be given
        if (nb & 0x0101u) {
            uint64_t c0, c1;
            frustum2(hot, g.wc, c0, c1);
            if (mixed) {
                if (!(ask & 0x0001u)) c0 = 0;
                if (!(ask & 0x0100u)) c1 = 0;
            }
            live0 &= ~c0;
            live1 &= ~c1;
        }
        if ((nb & 0x0202u) && (live0 | live1)) {
            const f2v camx = cam[0], camy = cam[1], camz = cam[2];
            uint64_t c0, c1, sure0, sure1;
            cone2_fast(camx, camy, camz, g.wc, g.cp, c0, c1, sure0, sure1);
            uint64_t slow0 = live0 & ~sure0, slow1 = live1 & ~sure1;
            if (mixed) {
                if (!(ask & 0x0002u)) c0 = slow0 = 0;
                if (!(ask & 0x0200u)) c1 = slow1 = 0;
                if ((exact_all & (1u << (2 * P))) && (ask & 0x0002u)) slow0 = live0;
                if ((exact_all & (2u << (2 * P))) && (ask & 0x0200u)) slow1 = live1;  // (bit i of exact_all: view i)
            }
            if (slow0 | slow1) {  // rare: the IEEE expression, in functions of their own so that their registers are not the kernel's
                const float4 wa = exact_axis();
                if (slow0) c0 = (c0 & ~slow0) | (__ballot(cone_exact(wa, g.wc, camx.x, camy.x, camz.x)) & slow0);
                if (slow1) c1 = (c1 & ~slow1) | (__ballot(cone_exact(wa, g.wc, camx.y, camy.y, camz.y)) & slow1);
                if (slow0) note_slow(views[2 * P], lane_of(slow0));
                if (slow1) note_slow(views[2 * P + 1], lane_of(slow1));
            }
            live0 &= ~c0;
            live1 &= ~c1;
        }
    }
    set_lane<(2 * P) * NG + K>(blo, (uint32_t)live0);
    set_lane<(2 * P) * NG + K>(bhi, (uint32_t)(live0 >> 32));
    set_lane<(2 * P + 1) * NG + K>(blo, (uint32_t)live1);
    set_lane<(2 * P + 1) * NG + K>(bhi, (uint32_t)(live1 >> 32));
}

template <int NG, bool NT, bool VP = false>
__global__ __launch_bounds__(kBlock) void cull_mask_views_kernel(
    const float4 *__restrict__ cone_axis_w, const float4 *__restrict__ centre_radius,
    const float4 *__restrict__ transforms, const float *__restrict__ xf_scale,
    const uint4 *__restrict__ group_rec, uint32_t groups_per_instance, uint32_t n_groups, ViewSet vs,
    const ViewConst *__restrict__ view_dev, const ViewHot2 *__restrict__ hot_dev, const uint2 *__restrict__ view_flags,
    const uint32_t *__restrict__ useful, uint32_t *__restrict__ useful_report) {
    static_assert(NG == 1 || NG == 2 || NG == 4, "the lanes of a view's groups are summed with lane shuffles");
    constexpr uint32_t kGroups = kWaves * NG;
    __shared__ uint4 s_rec[kGroups];
    __shared__ float4 s_x[kGroups][4];
    __shared__ float s_scale[kGroups];
    __shared__ uint32_t s_count[MCULL_MAX_VIEWS][kWaves];
    __shared__ uint2 s_vf[kGroups];  // the instance's need byte per view slot (classify_views_kernel)
    __shared__ uint64_t s_ballot[MCULL_MAX_VIEWS][kGroups];
    // the camera position of every view, two views to an element: staged by threads that have no group record to
    // fetch, read back as broadcasts
    __shared__ f2v s_cam[MCULL_MAX_VIEWS / 2][4];
    // how much the pre-pass found, left where the host can look at it later without waiting (host-mapped memory)
    if (useful_report && blockIdx.x == 0 && threadIdx.x == 0) *useful_report = *useful;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t block_g0 = blockIdx.x * kGroups;
    if (threadIdx.x < kGroups) {
        const uint32_t g = block_g0 + threadIdx.x;
        GroupInfo gi = group_lookup(g < n_groups ? g : n_groups - 1u, group_rec, groups_per_instance, vs.v[0].uniform_count);
        if (g >= n_groups) gi.count = 0;
        s_rec[threadIdx.x] = make_uint4(gi.inst, gi.first_meshlet, gi.count, gi.local_group);
    } else if (threadIdx.x >= 64u && threadIdx.x < 64u + (MCULL_MAX_VIEWS / 2) * 3u) {
        const uint32_t p = (threadIdx.x - 64u) / 3u, j = (threadIdx.x - 64u) % 3u;
        const ViewHot2 *hot = VP ? hot_dev + p : vs.hot + p;
        s_cam[p][j] = f2v{hot->cam[j][0], hot->cam[j][1]};
    }
    __syncthreads();
    if (threadIdx.x < kGroups * 4) {
        const uint32_t k = threadIdx.x >> 2, j = threadIdx.x & 3u;
        s_x[k][j] = transforms[(size_t)s_rec[k].x * 4 + j];
        if (j == 0) s_scale[k] = xf_scale[s_rec[k].x];
        if (j == 1 && view_flags) s_vf[k] = view_flags[s_rec[k].x];  // beside the bounds, not in front of them
    }
    float4 a[NG], c[NG];
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const uint4 r = s_rec[wave * NG + k];
        const uint32_t last = r.z ? r.z - 1u : 0u;
        const uint32_t id = r.y + (lane < last ? lane : last);
        a[k] = load_bounds<NT>(cone_axis_w + id);
        c[k] = load_bounds<NT>(centre_radius + id);
    }
    const ViewsUniform *up = VP ? reinterpret_cast<const ViewsUniform *>(hot_dev + MCULL_MAX_VIEWS / 2) : &vs.uni;
    const uint32_t def_lo = __builtin_amdgcn_readfirstlane(up->def_lo), def_hi = __builtin_amdgcn_readfirstlane(up->def_hi);
    const uint32_t exact_all = __builtin_amdgcn_readfirstlane(up->exact_all), mixed = __builtin_amdgcn_readfirstlane(up->mixed);
    __syncthreads();
    // what does not depend on the view, once per group
    ViewsGroup grp[NG];
    uint32_t any_hiz = 0;
    auto xform_of = [&](uint32_t slot) {
        Xform m;
        m.c0 = s_x[slot][0]; m.c1 = s_x[slot][1]; m.c2 = s_x[slot][2]; m.c3 = s_x[slot][3];
        return m;
    };
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const uint32_t slot = wave * NG + k;
        const Xform m = xform_of(slot);
        grp[k].wc = world_centre(m, s_scale[slot], c[k]);
        grp[k].cp = cone_pair(m, a[k]);
        const uint32_t n_valid = __builtin_amdgcn_readfirstlane(s_rec[slot].z);
        grp[k].in = n_valid >= 64u ? ~0ull : (1ull << n_valid) - 1ull;
        if (view_flags) {
            const uint2 f = s_vf[slot];
            grp[k].nb_lo = __builtin_amdgcn_readfirstlane(f.x);
            grp[k].nb_hi = __builtin_amdgcn_readfirstlane(f.y);
        } else {
            grp[k].nb_lo = def_lo;
            grp[k].nb_hi = def_hi;
        }
        any_hiz |= (grp[k].nb_lo | grp[k].nb_hi) & 0x04040404u;
    }
    const ViewConst *views = VP ? view_dev : vs.v;
    const ViewHot2 *hots = VP ? hot_dev : vs.hot;
    uint32_t blo = 0, bhi = 0;
    auto do_pair = [&](auto P) {
        constexpr int Pc = decltype(P)::value;
        if (2u * Pc >= vs.n) return;
        auto step = [&](auto K) {
            constexpr int Kc = decltype(K)::value;
            if constexpr (Kc < NG) {
                const uint32_t slot = wave * NG + Kc;
                // the IEEE axis of the lane's pair, for the lanes that leave the fast cone test: its bounds are read again
                auto exact_axis = [&]() {
                    const uint4 r = s_rec[slot];
                    const uint32_t last = r.z ? r.z - 1u : 0u;
                    return world_axis_again(transforms + (size_t)r.x * 4, cone_axis_w + r.y + (lane < last ? lane : last));
                };
                views_pair_step<Pc, Kc, NG>(grp[Kc], hots + Pc, s_cam[Pc], mixed, Pc < 2 ? def_lo : def_hi, exact_all, views, exact_axis, blo, bhi);
            }
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    };
    do_pair(std::integral_constant<int, 0>{}); do_pair(std::integral_constant<int, 1>{});
    do_pair(std::integral_constant<int, 2>{}); do_pair(std::integral_constant<int, 3>{});
    // ---- the occlusion test of the (view, group) combinations that still want it, on the ballots gathered so far
    if (any_hiz) {
        // Only the (view, group) combinations that still want the test are visited: bit 2 of every need byte, gathered
        // into one word (bit 8 * view + group).  The test itself is compiled ONCE, whichever
        // group it serves: the kernel has to fit the instruction cache.
        uint64_t want = 0;
#pragma unroll
        for (int k = 0; k < NG; k++)
            want |= ((uint64_t)((grp[k].nb_lo >> 2) & 0x01010101u) | ((uint64_t)((grp[k].nb_hi >> 2) & 0x01010101u) << 32)) << k;
        while (want) {
            const uint32_t pos = (uint32_t)__builtin_ctzll(want);
            want &= want - 1ull;
            const uint32_t i = pos >> 3, k = pos & 7u;
            const uint32_t l = i * NG + k;
            const uint64_t live = ((uint64_t)lane_u32(bhi, l) << 32) | lane_u32(blo, l);
            if (!live) continue;
            float4 w = grp[0].wc;
#pragma unroll
            for (int q = 1; q < NG; q++)
                if (k == (uint32_t)q) w = grp[q].wc;
            const uint64_t left = live & ~__ballot(occlusion1(views[i], w, lane_of(live)));
            if (lane == l) { blo = (uint32_t)left; bhi = (uint32_t)(left >> 32); }
        }
    }
    // ---- ballots and counts: lane view * NG + group owns one of each
    if (lane < vs.n * NG) s_ballot[lane / NG][wave * NG + lane % NG] = ((uint64_t)bhi << 32) | blo;
    uint32_t pop = __popc(blo) + __popc(bhi);
    if (NG >= 2) pop += __shfl_xor(pop, 1, 64);
    if (NG >= 4) pop += __shfl_xor(pop, 2, 64);
    if (lane < vs.n * NG && lane % NG == 0u) s_count[lane / NG][wave] = pop;
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < vs.n * kGroups; t += kBlock) {
        const uint32_t i = t / kGroups, slot = t % kGroups;
        if (s_rec[slot].z) vs.mask[i][block_g0 + slot] = s_ballot[i][slot];
    }
    if (threadIdx.x < vs.n) {
        const uint32_t i = threadIdx.x;
        const uint32_t sum = (s_count[i][0] + s_count[i][1]) + (s_count[i][2] + s_count[i][3]);
        if (sum) atomicAdd(&vs.tile_counts[i][block_g0 / kTileGroups], sum);
    }
}

// ------------------------------------------------------------------------------------------------
// Scenes with more than kScanMinTiles tiles: the list offset of every tile (exclusive 64-bit prefix of the
// tile counters) and the total.  One workgroup per 8192 counters: it first sums every counter below its share
// (16-byte loads from L2, all threads), then scans its own.  100 M pairs have 6 104 counters, one workgroup;
// 13.9 G pairs have 851 000, 104 workgroups, of which the last reads 3.4 MB - against one workgroup walking
// all of them with a running carry, two barriers per 8192, which took 0.50 ms there.  The scatter pass then
// starts each tile with one load instead of summing the lower counters itself.
__device__ __forceinline__ void scan_tiles_body(const uint32_t *__restrict__ tile_counts, uint32_t n_tiles,
                                                unsigned long long *__restrict__ tile_offsets,
                                                unsigned long long *__restrict__ total) {
    constexpr int kScanWaves = kScanBlock / 64;
    __shared__ unsigned long long s_wave[kScanWaves];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * (kScanBlock * kScanPerThread);
    unsigned long long carry = 0;
    if (base) {  // counters are 256-byte aligned and base is a multiple of 8192
        const uint4 *lower = reinterpret_cast<const uint4 *>(tile_counts);
        unsigned long long part = 0;
        // eight loads in flight per thread: the last workgroup's walk over the lower counters is one thread's chain of
        // round trips (851 000 counters: 89 us with one load per round trip)
        const uint32_t n4 = base / 4u;
        uint32_t j = threadIdx.x;
        for (; j + 7u * kScanBlock < n4; j += 8u * kScanBlock) {
            uint4 q[8];
#pragma unroll
            for (int k = 0; k < 8; k++) q[k] = lower[j + (uint32_t)k * kScanBlock];
#pragma unroll
            for (int k = 0; k < 8; k++) part += ((unsigned long long)q[k].x + q[k].y) + ((unsigned long long)q[k].z + q[k].w);
        }
        for (; j < n4; j += kScanBlock) {
            const uint4 q = lower[j];
            part += ((unsigned long long)q.x + q.y) + ((unsigned long long)q.z + q.w);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        if (lane == 0) s_wave[wave] = part;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kScanWaves; w++) carry += s_wave[w];
        __syncthreads();
    }
    const uint32_t j0 = base + threadIdx.x * kScanPerThread;
    uint32_t x[kScanPerThread];
    if (j0 + kScanPerThread <= n_tiles) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(tile_counts + j0);
        const uint4 hi = *reinterpret_cast<const uint4 *>(tile_counts + j0 + 4);
        x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w; x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; k++) x[k] = j0 + k < n_tiles ? tile_counts[j0 + k] : 0u;
    }
    unsigned long long sum = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) sum += x[k];
    unsigned long long incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long up = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long wave_base = 0, chunk = 0;
#pragma unroll
    for (int w = 0; w < kScanWaves; w++) {
        const unsigned long long t = s_wave[w];
        if ((uint32_t)w < wave) wave_base += t;
        chunk += t;
    }
    unsigned long long off = carry + wave_base + incl - sum;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) {
        if (j0 + k < n_tiles) tile_offsets[j0 + k] = off;
        off += x[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = carry + chunk;
}

__global__ __launch_bounds__(kScanBlock) void scan_tiles_kernel(const uint32_t *__restrict__ tile_counts, uint32_t n_tiles,
                                                                unsigned long long *__restrict__ tile_offsets,
                                                                unsigned long long *__restrict__ total) {
    scan_tiles_body(tile_counts, n_tiles, tile_offsets, total);
}

// The scatter and command passes of several views in one launch each (blockIdx.y = view): the buffers of every view.
struct ViewTails {
    ViewTail t[MCULL_MAX_VIEWS];
};

__global__ __launch_bounds__(kScanBlock) void scan_tiles_views_kernel(ViewTails tails, uint32_t n_tiles) {
    const ViewTail &t = tails.t[blockIdx.y];
    scan_tiles_body(t.tile_counts, n_tiles, (unsigned long long *)t.tile_offsets, (unsigned long long *)t.total);
}

// ------------------------------------------------------------------------------------------------
// ordered, pass 2: scatter.  `split` workgroups per 256-group tile (each redoes the tile's scan and
// writes the entries of 256/split consecutive groups; the launcher uses 1 except for very small scenes).
// A tile's list offset is the number of visible pairs in all lower tiles: read from tile_offsets
// (scan_tiles_kernel, which has also written the total) or, for scenes of at most kScanMinTiles tiles,
// summed here from the lower tile counters (read from L2).  Then one group per
// thread for the in-tile scan, and one group per wave-iteration for the stores (lane l owns bit l,
// the slots of a group are contiguous).
// LOOP (scenes that skip empty tiles: hundreds of thousands of tiles, most of them empty): a fixed number of
// workgroups walks the tiles with a stride instead of one workgroup per tile, so that an empty tile costs a scalar load
// and a branch, not a workgroup (1 M instances of one mesh, 851 000 tiles: 1.5 ms of this pass were workgroups that
// started and left).
template <bool NT, bool LOOP, bool SHORT, bool WIDE>
__device__ __forceinline__ void compact_body(
    const uint64_t *__restrict__ mask, const uint32_t *__restrict__ tile_counts,
    const unsigned long long *__restrict__ tile_offsets, const uint4 *__restrict__ group_rec, uint32_t groups_per_instance,
    uint32_t n_groups, mcull_visible *__restrict__ list, uint64_t capacity, uint32_t *__restrict__ inst_first,
    unsigned long long *__restrict__ total, uint32_t split, uint32_t skip_empty) {
    __shared__ uint64_t s_word[kBlock];
    __shared__ uint32_t s_off[kBlock];
    __shared__ uint2 s_fi[kBlock];  // {first meshlet, instance} of the tile's groups
    __shared__ uint32_t s_wave[kWaves];
    __shared__ unsigned long long s_sum[kWaves];
    constexpr uint32_t kRound = (LOOP || SHORT || WIDE) ? kShortRoundGroups : kRoundGroups;  // groups staged per output round
    // what is staged per surviving entry is two bytes - its group within the round and its lane; the eight bytes that go
    // out are put together from the group's record when they are stored
    // (WIDE: the eight bytes themselves, for the dense list of a single view - see kRoundGroups)
    __shared__ uint16_t s_code[WIDE ? 1 : kRound * 64];
    __shared__ unsigned long long s_wide[WIDE ? kRound * 64 : 1];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t round_span = kTileGroups / split;  // groups whose entries this workgroup writes
    // (kRound <= round_span: the launchers pair more than one workgroup per tile with the short rounds)
    auto do_tile = [&](const uint32_t tile_idx, const uint32_t part_idx) {
    unsigned long long running;
    if (tile_offsets) {
        // A tile without a visible pair has nothing to write; the list offsets of the instances that begin in it all
        // equal the tile's, and the command pass takes them from the tile offsets itself (skip_empty; 1 M instances of
        // one mesh, nine tiles in ten empty: 1.9 -> 1.5 ms for this pass).  The launcher asks for it when instances span
        // many groups; with a hundred instances per tile the command pass's extra look-ups cost more (2.4 us for
        // 781 250 instances) than the rare empty tile saves.
        if (skip_empty && tile_counts[tile_idx] == 0u) return;
        running = tile_offsets[tile_idx];  // uniform: one scalar load
    } else {
        unsigned long long part = 0;
        for (uint32_t j = threadIdx.x; j < tile_idx; j += kBlock) part += tile_counts[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        if (lane == 0) s_sum[wave] = part;
        __syncthreads();
        running = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = running + tile_counts[tile_idx];
    }
    const uint32_t tile = tile_idx * kTileGroups;
    const uint32_t g_end = min(tile + kTileGroups, n_groups);
    const uint32_t g = tile + threadIdx.x;
    // A tile without a visible pair (its counter says so: whole instances dropped by the pre-cull or rejected
    // by the tests) has no entry to write and every ballot word of it is zero: only the list offsets of the
    // instances that begin in it remain, and they all equal the tile's.
    const bool empty_tile = tile_counts[tile_idx] == 0u;  // uniform
    uint64_t word = 0;
    uint32_t inst = 0, first = 0, is_first_group = 0;
    if (g < g_end) {
        if (!empty_tile) word = mask[g];
        if (group_rec) {
            const uint4 r = group_rec[g];
            inst = r.x; first = r.y; is_first_group = (r.w == 0u);
        } else {
            inst = g / groups_per_instance;
            const uint32_t lg = g - inst * groups_per_instance;
            first = lg * 64u;
            is_first_group = (lg == 0u);
        }
    }
    if (empty_tile) {
        if (g < g_end && is_first_group && threadIdx.x / round_span == part_idx) inst_first[inst] = (uint32_t)running;
        return;
    }
    uint32_t tile_total;
    const uint32_t excl = block_exclusive_scan((uint32_t)__popcll(word), s_wave, &tile_total);
    s_word[threadIdx.x] = word;
    s_off[threadIdx.x] = excl;
    s_fi[threadIdx.x] = make_uint2(first, inst);
    // firstTask of an instance = list offset of its first group (exclusive scan in instance order)
    if (g < g_end && is_first_group && threadIdx.x / round_span == part_idx) inst_first[inst] = (uint32_t)(running + excl);
    __syncthreads();
    const uint32_t n_tile = g_end - tile;
    // The surviving entries are first placed in LDS (each group's slots are contiguous, lane l owns
    // bit l), kRound groups at a time, and then streamed out linearly with aligned 16-byte
    // stores: whole cache lines instead of one partial, unaligned run per group.
    // Where the list starts inside a 128-byte line, in 8-byte entries.  The stores below are dealt to the threads from the
    // line boundary at or in front of a round's first entry, so that every store instruction of a wave (64 x 16 bytes)
    // covers eight whole lines: non-temporal stores that straddle lines are written as partial lines (a probe that
    // streams 67 KB regions per workgroup: 5.4 TB/s from a line boundary, 4.26 TB/s from 48 bytes behind one - the
    // rate this pass ran at while it only aligned to 16 bytes).
    const uint32_t base_lead = (uint32_t)(reinterpret_cast<uintptr_t>(list) >> 3) & 15u;
    const uint32_t span_end = (part_idx + 1u) * round_span, g_stop = n_tile < span_end ? n_tile : span_end;
    for (uint32_t gbeg = part_idx * round_span; gbeg < g_stop; gbeg += kRound) {
        const uint32_t round_off = s_off[gbeg];
        const uint32_t round_end = (gbeg + kRound < n_tile) ? s_off[gbeg + kRound] : tile_total;
        const uint32_t n_r = round_end - round_off;
        if (n_r == 0u) continue;  // uniform: nothing of these groups is visible
#pragma unroll 4
        for (uint32_t j = 0; j < kRound / kWaves; j++) {
            const uint32_t gidx = gbeg + wave * (kRound / kWaves) + j;
            if (gidx < n_tile) {
                const uint64_t w = s_word[gidx];
                if ((w >> lane) & 1ull) {
                    const uint32_t li = s_off[gidx] - round_off + (uint32_t)__popcll(w & ((1ull << lane) - 1ull));
                    if (WIDE) s_wide[li] = ((unsigned long long)(s_fi[gidx].x + lane) << 32) | s_fi[gidx].y;
                    else s_code[li] = (uint16_t)(((gidx - gbeg) << 6) | lane);
                }
            }
        }
        __syncthreads();
        const unsigned long long s0 = running + round_off;
        const uint32_t odd = ((uint32_t)s0 + base_lead) & 15u;  // entries between the line boundary and the round's first one
        unsigned long long *out = reinterpret_cast<unsigned long long *>(list);
        auto entry = [&](int32_t at) {
            if (WIDE) return s_wide[at];
            const uint32_t code = s_code[at];
            const uint2 fi = s_fi[gbeg + (code >> 6)];
            return ((unsigned long long)(fi.x + (code & 63u)) << 32) | fi.y;
        };
        for (uint32_t p = threadIdx.x; 2 * p < n_r + odd; p += kBlock) {
            const int32_t li = (int32_t)(2 * p) - (int32_t)odd;
            const bool lo = li >= 0, hi = (uint32_t)(li + 1) < n_r;
            const unsigned long long slot = s0 + (long long)li;
            if (lo && hi && slot + 1 < capacity) {
                const ull2v e = {entry(li), entry(li + 1)};
                if (NT) __builtin_nontemporal_store(e, reinterpret_cast<ull2v *>(out + slot));
                else *reinterpret_cast<ull2v *>(out + slot) = e;
            } else {
                if (lo && slot < capacity) {
                    if (NT) __builtin_nontemporal_store(entry(li), out + slot);
                    else out[slot] = entry(li);
                }
                if (hi && slot + 1 < capacity) {
                    if (NT) __builtin_nontemporal_store(entry(li + 1), out + slot + 1);
                    else out[slot + 1] = entry(li + 1);
                }
            }
        }
        __syncthreads();
    }
    };
    if (LOOP) {
        // The workgroup's tiles are blockIdx.x + k gridDim.x.  Their counters are looked at kBlock at a time, one per thread
        // and all loads at once, and only the tiles that hold something are visited: asking one counter after the other put
        // a memory round trip in front of every tile, empty or not (1 M instances of one mesh, 104 tiles per workgroup, nine
        // in ten empty: 1.76 ms for a pass that writes 5.4 GB).
        __shared__ uint64_t s_has[kWaves];
        const uint32_t n_tiles = (n_groups + kTileGroups - 1u) / kTileGroups;
        for (uint64_t base = blockIdx.x; base < n_tiles; base += (uint64_t)gridDim.x * kBlock) {
            const uint64_t mine = base + (uint64_t)threadIdx.x * gridDim.x;
            const uint64_t has = __ballot(mine < n_tiles && tile_counts[mine] != 0u);
            if (lane == 0) s_has[wave] = has;
            __syncthreads();
            for (uint32_t w = 0; w < (uint32_t)kWaves; w++) {
                uint64_t m = s_has[w];  // uniform
                while (m) {
                    const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1ull;
                    do_tile((uint32_t)(base + (uint64_t)(w * 64u + bit) * gridDim.x), 0u);
                    __syncthreads();  // the staging arrays are rewritten by the next tile
                }
            }
            __syncthreads();  // s_has is rewritten by the next batch
        }
    } else {
        do_tile(blockIdx.x / split, blockIdx.x % split);
    }
}

template <bool NT, bool LOOP = false, bool SHORT = false>
__global__ __launch_bounds__(kBlock) void compact_kernel(
    const uint64_t *__restrict__ mask, const uint32_t *__restrict__ tile_counts,
    const unsigned long long *__restrict__ tile_offsets, const uint4 *__restrict__ group_rec, uint32_t groups_per_instance,
    uint32_t n_groups, mcull_visible *__restrict__ list, uint64_t capacity, uint32_t *__restrict__ inst_first,
    unsigned long long *__restrict__ total, uint32_t split, uint32_t skip_empty) {
    compact_body<NT, LOOP, SHORT, !LOOP && !SHORT>(mask, tile_counts, tile_offsets, group_rec, groups_per_instance, n_groups, list, capacity, inst_first, total,
                           split, skip_empty);
}

template <bool NT, bool LOOP = false, bool SHORT = false>
__global__ __launch_bounds__(kBlock) void compact_views_kernel(ViewTails tails, uint32_t with_offsets, const uint4 *__restrict__ group_rec,
                                                               uint32_t groups_per_instance, uint32_t n_groups, uint32_t split,
                                                               uint32_t skip_empty) {
    const ViewTail &t = tails.t[blockIdx.y];
    compact_body<NT, LOOP, SHORT, false>(t.mask, t.tile_counts, with_offsets ? (const unsigned long long *)t.tile_offsets : nullptr, group_rec,
                           groups_per_instance, n_groups, t.list, t.capacity, t.inst_first, (unsigned long long *)t.total, split, skip_empty);
}

// ------------------------------------------------------------------------------------------------
// per-instance indirect commands (createIndirectBuffer, Engine.cpp:1466-1484, made dynamic).
// ordered mode: inst_first[] holds the list offset of every instance that has meshlets;
// taskCount_i = firstTask of the next non-empty instance (or the total) - firstTask_i.
// atomic modes: inst_count[] holds visible_i, firstTask = base.
__device__ __forceinline__ void emit_commands_body(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ next_nonempty, uint32_t uniform_count, uint32_t n_instances,
    const uint32_t *__restrict__ inst_first, const uint32_t *__restrict__ inst_count,
    const uint64_t *__restrict__ total, const uint64_t *__restrict__ rank_totals, uint32_t rank,
    uint32_t stride, uint32_t offset, uint32_t form, uint8_t *__restrict__ commands,
    uint32_t *__restrict__ tile_counts, uint32_t n_tiles, const unsigned long long *__restrict__ tile_offsets,
    const uint32_t *__restrict__ group_base, uint32_t groups_per_instance, uint32_t *__restrict__ inst_first_w,
    uint32_t *__restrict__ work_header = nullptr) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    // every consumer of the tile counters has finished (stream order): leave them zero for the next cull
    if (tile_counts && i < n_tiles) tile_counts[i] = 0;
    // ... and of the instance level's work list: its length moves to word 1 (what mcull_precull_stats reads) and word
    // 0, which the next cull's pre-pass reserves from, is zero again without a launch of its own
    if (work_header && i == 0) {
        work_header[1] = work_header[0];
        work_header[0] = 0u;
    }
    if (i >= n_instances) return;
    unsigned long long base = 0;
    if (rank_totals)
        for (uint32_t r = 0; r < rank; r++) base += rank_totals[r];
    uint32_t task_count, first_task;
    if (inst_count) {
        task_count = inst_count[i];
        first_task = (uint32_t)base;
    } else {
        const unsigned long long tot64 = *total;
        const uint32_t tot = (uint32_t)tot64;
        const uint32_t mine = ranges ? ranges[i].y : uniform_count;
        // The list offset of instance j.  With tile offsets (scan_tiles_kernel) the scatter pass skips the tiles
        // without a visible pair; an instance that begins in one starts at that tile's offset, which is also
        // stored into inst_first for later readers of the array (nobody reads that element in this launch: every
        // reader takes this same branch for it).
        auto first_of = [&](uint32_t j, bool store) {
            if (tile_offsets) {
                const uint32_t tile = (ranges ? group_base[j] : j * groups_per_instance) / kTileGroups;
                if (tile < n_tiles) {
                    const unsigned long long lo = tile_offsets[tile], hi = tile + 1u < n_tiles ? tile_offsets[tile + 1u] : tot64;
                    if (hi == lo) {
                        if (store) inst_first_w[j] = (uint32_t)lo;
                        return (uint32_t)lo;
                    }
                }
            }
            return inst_first[j];
        };
        // the next instance that owns meshlets: looked up (precomputed with the ranges: a scene with long runs of
        // empty instances would otherwise cost every thread a walk over the run), or simply the next one
        const uint32_t k = ranges ? next_nonempty[i] : i + 1u;
        const uint32_t next_first = k < n_instances ? first_of(k, false) : tot;
        const uint32_t my_first = mine ? first_of(i, true) : next_first;
        task_count = next_first - my_first;
        first_task = my_first + (uint32_t)base;
    }
    uint32_t *out = (uint32_t *)(commands + (size_t)i * stride + offset);
    // NV_TASK_GROUPS: task workgroups of 32 (the reference's granularity, Engine.cpp:1475-1483, shader.task:12)
    out[0] = form == MCULL_CMD_NV_TASK_GROUPS ? (task_count + 31u) / 32u : task_count;
    if (form == MCULL_CMD_EXT_GROUPS) {  // {groupCountX, groupCountY, groupCountZ}
        out[1] = 1u;
        out[2] = 1u;
    } else {
        out[1] = first_task;
    }
}

__global__ __launch_bounds__(kBlock) void emit_commands_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ next_nonempty, uint32_t uniform_count, uint32_t n_instances,
    const uint32_t *__restrict__ inst_first, const uint32_t *__restrict__ inst_count,
    const uint64_t *__restrict__ total, const uint64_t *__restrict__ rank_totals, uint32_t rank,
    uint32_t stride, uint32_t offset, uint32_t form, uint8_t *__restrict__ commands,
    uint32_t *__restrict__ tile_counts, uint32_t n_tiles, const unsigned long long *__restrict__ tile_offsets,
    const uint32_t *__restrict__ group_base, uint32_t groups_per_instance, uint32_t *__restrict__ inst_first_w,
    uint32_t *__restrict__ work_header) {
    emit_commands_body(ranges, next_nonempty, uniform_count, n_instances, inst_first, inst_count, total, rank_totals, rank, stride, offset,
                       form, commands, tile_counts, n_tiles, tile_offsets, group_base, groups_per_instance, inst_first_w, work_header);
}

// ordered mode only
__global__ __launch_bounds__(kBlock) void emit_commands_views_kernel(ViewTails tails, uint32_t with_offsets, const uint2 *__restrict__ ranges,
                                                                     const uint32_t *__restrict__ next_nonempty, uint32_t uniform_count,
                                                                     uint32_t n_instances, uint32_t n_tiles,
                                                                     const uint32_t *__restrict__ group_base, uint32_t groups_per_instance) {
    const ViewTail &t = tails.t[blockIdx.y];
    emit_commands_body(ranges, next_nonempty, uniform_count, n_instances, t.inst_first, nullptr, t.total, nullptr, 0u, t.lay.stride,
                       t.lay.offset, t.lay.form, (uint8_t *)t.commands, t.tile_counts, n_tiles,
                       with_offsets ? (const unsigned long long *)t.tile_offsets : nullptr, group_base, groups_per_instance, t.inst_first);
}

// ------------------------------------------------------------------------------------------------
// single-pass compaction with atomics.  BLOCK = false: one returning atomic per wavefront
// (ballot -> popcount -> lane 0 atomicAdd -> readfirstlane broadcast -> mbcnt scatter);
// BLOCK = true: the four waves of a workgroup share one atomic per step.
template <uint32_t TESTS, bool BLOCK>
__global__ __launch_bounds__(kBlock) void cull_atomic_kernel(
    const float4 *__restrict__ cone_axis_w, const float4 *__restrict__ centre_radius,
    const float4 *__restrict__ transforms, const float *__restrict__ xf_scale,
    const uint4 *__restrict__ group_rec, uint32_t groups_per_instance, uint32_t n_groups, ViewConst v,
    mcull_visible *__restrict__ list, uint64_t capacity, uint32_t *__restrict__ inst_count,
    unsigned long long *__restrict__ total, uint64_t *__restrict__ mask) {
    __shared__ uint32_t s_pop[2][kWaves];
    __shared__ unsigned long long s_base[2];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g_begin = blockIdx.x * kTileGroups;
    const uint32_t g_end = min(g_begin + kTileGroups, n_groups);
    const uint32_t steps = (g_end - g_begin + kWaves - 1) / kWaves;
    for (uint32_t it = 0; it < steps; it++) {
        const uint32_t g = g_begin + it * kWaves + wave;
        const bool active = g < g_end;
        bool vis = false;
        GroupInfo gi = {0, 0, 0, 0};
        if (active) {
            gi = group_lookup(g, group_rec, groups_per_instance, v.uniform_count);
            const Xform m = load_xform(transforms, gi.inst);
            if (lane < gi.count) {
                const uint32_t id = gi.first_meshlet + lane;
                vis = pair_visible<TESTS, false>(v, m, xf_scale[gi.inst], cone_axis_w[id], centre_radius[id]);
            }
        }
        const uint64_t ballot = __ballot(vis);
        const uint32_t pop = (uint32_t)__popcll(ballot);
        if (active && mask && lane == 0) mask[g] = ballot;
        unsigned long long base = 0;
        if (BLOCK) {
            const uint32_t buf = it & 1u;
            if (lane == 0) s_pop[buf][wave] = pop;
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t t = (s_pop[buf][0] + s_pop[buf][1]) + (s_pop[buf][2] + s_pop[buf][3]);
                s_base[buf] = t ? atomicAdd(total, (unsigned long long)t) : 0ull;
            }
            __syncthreads();
            base = s_base[buf];
            for (uint32_t w = 0; w < wave; w++) base += s_pop[buf][w];
        } else if (pop) {
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(total, (unsigned long long)pop);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
            const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
            base = ((unsigned long long)hi << 32) | lo;
        }
        if (pop) {
            if (lane == 0) atomicAdd(&inst_count[gi.inst], pop);
            if (vis) {
                const unsigned long long slot = base + (unsigned long long)__popcll(ballot & ((1ull << lane) - 1ull));
                if (slot < capacity) store_entry<false>(list, slot, gi.inst, gi.first_meshlet + lane);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void xform_scale_kernel(const float4 *__restrict__ transforms, uint32_t n,
                                                             float *__restrict__ xf_scale) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) xf_scale[i] = xform_scale(transforms[(size_t)i * 4]);
}

// ------------------------------------------------------------------------------------------------
// one 16-byte record per group for scenes with per-instance ranges
__global__ __launch_bounds__(kBlock) void fill_group_rec_kernel(const uint2 *__restrict__ ranges,
                                                                const uint32_t *__restrict__ group_base,
                                                                uint32_t n_instances, uint4 *__restrict__ group_rec) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_instances) return;
    const uint32_t b = group_base[i], e = group_base[i + 1];
    const uint2 r = ranges[i];
    for (uint32_t g = b; g < e; g++) {
        const uint32_t lg = g - b;
        const uint32_t left = r.y - lg * 64u;
        group_rec[g] = make_uint4(i, r.x + lg * 64u, left < 64u ? left : 64u, lg);
    }
}

// The literal payload of every 32-wide task workgroup (shader.task:28-35, :103-118) rebuilt from
// the ballots: payload (draw, wg) takes bits [32*wg, 32*wg+32) of the draw's mask words.
__global__ __launch_bounds__(kBlock) void task_payload_kernel(const uint64_t *__restrict__ mask,
                                                              uint32_t groups_per_instance,
                                                              uint32_t payloads_per_instance, uint64_t n_payloads,
                                                              mcull_task_payload *__restrict__ out) {
    const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_payloads) return;
    const uint32_t draw = (uint32_t)(p / payloads_per_instance);
    const uint32_t wg = (uint32_t)(p - (uint64_t)draw * payloads_per_instance);
    const uint64_t word = mask[(uint64_t)draw * groups_per_instance + (wg >> 1)];
    uint32_t bits = (uint32_t)(word >> ((wg & 1u) * 32u));
    mcull_task_payload *o = out + p;
    uint32_t n = 0;
    for (uint32_t k = 0; k < 32; k++) o->meshletIndices[k] = 0;
    while (bits) {
        const uint32_t b = __builtin_ctz(bits);
        bits &= bits - 1;
        o->meshletIndices[n++] = wg * 32u + b;
    }
    o->drawID = draw;
    o->taskCount = n;
}

// ------------------------------------------------------------------------------------------------
uint32_t tile_count_for(uint64_t n_groups) { return (uint32_t)((n_groups + kTileGroups - 1) / kTileGroups); }

#define MCULL_DISPATCH_TESTS(tests, CALL)                                        \
    switch ((tests) & 7u) {                                                      \
        case 0: { constexpr uint32_t T = 0; CALL; break; }                       \
        case 1: { constexpr uint32_t T = 1; CALL; break; }                       \
        case 2: { constexpr uint32_t T = 2; CALL; break; }                       \
        case 3: { constexpr uint32_t T = 3; CALL; break; }                       \
        case 4: { constexpr uint32_t T = 4; CALL; break; }                       \
        case 5: { constexpr uint32_t T = 5; CALL; break; }                       \
        case 6: { constexpr uint32_t T = 6; CALL; break; }                       \
        default: { constexpr uint32_t T = 7; CALL; break; }                      \
    }

// above 1024 tiles (16.8 M pairs) the scan launch pays for itself
bool needs_tile_offsets(uint64_t n_groups) { return tile_count_for(n_groups) > kScanMinTiles; }

// est_groups (0 = unknown; only with a work list and without `flags`): how many groups the list is expected to hold - what
// an earlier cull of the slot reported, plus a margin.  The grid of the streaming form is then sized for that instead of
// for the whole scene, and a second, small launch of the walking form takes whatever the list holds beyond it, so the
// result does not depend on the estimate.  A grid sized for the scene starts a workgroup per 16 groups of the SCENE, and
// those behind the list's end - a third of them for the 100 M-pair scene, six in seven with its occlusion test on - start,
// wait for the list's length and leave, as a tail during which nothing else runs.
template <int NG, bool NT>
static void launch_cull_mask_t(const SceneDev &s, const ViewConst &v, uint64_t *mask, uint32_t *tile_counts,
                               const ViewConst *view_dev, const uint32_t *work, uint32_t *kept_report, bool flags, bool list_of_groups,
                               uint64_t est_groups, hipStream_t st) {
    const uint32_t per_block = kWaves * NG;
    const uint32_t log = (flags && work && list_of_groups && !s.group_rec) ? 1u : 0u;
    uint32_t grid = (uint32_t)((s.n_groups + per_block - 1) / per_block);
    if (flags && work) {
        const uint32_t walkers = grid < 16384u ? grid : 16384u;  // the FLAGS form walks the list with a stride
        if (view_dev)
            hipLaunchKernelGGL((cull_mask_kernel<7u, NG, NT, true, true>), dim3(walkers), dim3(kBlock), 0, st, s.cone_axis_w, s.centre_radius,
                               s.transforms, s.xf_scale, s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, v, mask,
                               tile_counts, 0u, view_dev, work, kept_report, log, 0u);
        else
            hipLaunchKernelGGL((cull_mask_kernel<7u, NG, NT, false, true>), dim3(walkers), dim3(kBlock), 0, st, s.cone_axis_w, s.centre_radius,
                               s.transforms, s.xf_scale, s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, v, mask,
                               tile_counts, 0u, (const ViewConst *)nullptr, work, kept_report, log, 0u);
        return;
    }
    uint32_t tail_from = 0, tail_grid = 0;
    if (work && est_groups) {
        const uint64_t est_blocks = (est_groups + per_block - 1) / per_block;  // the caller has added its margin
        if (est_blocks < grid) {
            tail_from = (uint32_t)est_blocks;
            tail_grid = grid - tail_from < 1024u ? grid - tail_from : 1024u;
            grid = tail_from;
        }
    }
    if (view_dev) {
        hipLaunchKernelGGL((cull_mask_kernel<7u, NG, NT, true>), dim3(grid), dim3(kBlock), 0, st, s.cone_axis_w, s.centre_radius,
                           s.transforms, s.xf_scale, s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, v, mask,
                           tile_counts, 0u, view_dev, work, kept_report, 0u, 0u);
        if (tail_grid)
            hipLaunchKernelGGL((cull_mask_kernel<7u, NG, NT, true, true>), dim3(tail_grid), dim3(kBlock), 0, st, s.cone_axis_w,
                               s.centre_radius, s.transforms, s.xf_scale, s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, v,
                               mask, tile_counts, 0u, view_dev, work, (uint32_t *)nullptr, 0u, tail_from);
        return;
    }
    // streamed bounds behind a work list, no occlusion test: the one-group step (see PK)
    const bool scalar_step = NT && work != nullptr && !(v.tests & MCULL_TEST_HIZ);
    MCULL_DISPATCH_TESTS(v.tests, {
        if constexpr (NT && (T & MCULL_TEST_HIZ) == 0u) {
            if (scalar_step) {
                hipLaunchKernelGGL((cull_mask_kernel<T, NG, NT, false, false, false>), dim3(grid), dim3(kBlock), 0, st, s.cone_axis_w,
                                   s.centre_radius, s.transforms, s.xf_scale, s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, v,
                                   mask, tile_counts, 0u, (const ViewConst *)nullptr, work, kept_report, 0u, 0u);
                break;
            }
        }
        hipLaunchKernelGGL((cull_mask_kernel<T, NG, NT>), dim3(grid), dim3(kBlock), 0, st, s.cone_axis_w, s.centre_radius, s.transforms,
                           s.xf_scale, s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, v, mask, tile_counts, 0u,
                           (const ViewConst *)nullptr, work, kept_report, 0u, 0u);
    });
    if (tail_grid)
        hipLaunchKernelGGL((cull_mask_kernel<7u, NG, NT, false, true>), dim3(tail_grid), dim3(kBlock), 0, st, s.cone_axis_w,
                           s.centre_radius, s.transforms, s.xf_scale, s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, v, mask,
                           tile_counts, 0u, (const ViewConst *)nullptr, work, (uint32_t *)nullptr, 0u, tail_from);
}

// view_dev != nullptr: the kernels take the view constants (and the tests) from that device record
hipError_t launch_cull_mask(const SceneDev &s, const ViewConst &v, uint64_t *mask, uint32_t *tile_counts,
                            bool streaming, hipStream_t st, const ViewConst *view_dev, const uint32_t *work, uint32_t *kept_report,
                            bool flags, bool list_of_groups, uint64_t est_groups) {
    // groups per wave: more requests in flight per wave for big scenes, enough workgroups to fill
    // 256 CUs for small ones
#define MCULL_MASK_T(NG_, NT_) launch_cull_mask_t<NG_, NT_>(s, v, mask, tile_counts, view_dev, work, kept_report, flags, list_of_groups, est_groups, st)
    if (streaming) {
        // a work list expected to be short, with the occlusion test on, is spread over twice as many waves: 211 000 listed
        // groups of the 100 M-pair scene, 0.230 -> 0.221 ms per step (the 1.06 M groups listed without it: four per wave).
        // Without the occlusion test a short list keeps four per wave too (133 000 listed groups of a 12.5 M-pair shard:
        // 0.0945 -> 0.0916 ms per step).
        const bool short_list = work && !flags && est_groups && est_groups < (1u << 18) && (v.tests & MCULL_TEST_HIZ);
        if (s.n_groups >= (1u << 16) && !short_list) MCULL_MASK_T(4, true);
        else MCULL_MASK_T(2, true);
    } else {
        if (s.n_groups >= (1u << 16)) MCULL_MASK_T(4, false);  // (1 M instances of one mesh behind a work list: 17.9 ms; two per wave 18.5, one 25.6)
        else if (s.n_groups >= (1u << 14)) MCULL_MASK_T(2, false);
        else MCULL_MASK_T(1, false);
    }
#undef MCULL_MASK_T
    return hipGetLastError();
}

hipError_t launch_classify_views(const SceneDev &s, const float4 *spheres, bool per_instance_sphere, uint32_t n_views,
                                 const ViewConst *const *views, const ViewConst *view_dev, uint8_t *flags, uint32_t *useful,
                                 hipStream_t st) {
    if (s.n_instances == 0) return hipSuccess;
    ViewSet vs;
    std::memset(&vs, 0, sizeof vs);
    vs.n = n_views;
    for (uint32_t i = 0; i < n_views; i++) vs.v[i] = *views[i];
    const float *rmax = reinterpret_cast<const float *>(spheres + (per_instance_sphere ? s.n_instances : 1u));
    hipLaunchKernelGGL(classify_views_kernel, dim3((s.n_instances + kBlock - 1) / kBlock), dim3(kBlock), 0, st, s.transforms, spheres,
                       rmax, per_instance_sphere ? 1u : 0u, s.n_instances, vs, view_dev, reinterpret_cast<uint2 *>(flags), useful);
    return hipGetLastError();
}

hipError_t launch_cull_mask_views(const SceneDev &s, uint32_t n_views, const ViewConst *const *views, uint64_t *const *masks,
                                  uint32_t *const *tile_counts, bool streaming, hipStream_t st, const ViewConst *view_dev,
                                  const ViewHot2 *hot_dev, const uint8_t *view_flags, const uint32_t *useful, uint32_t *useful_report) {
    ViewSet vs;
    std::memset(&vs, 0, sizeof vs);
    vs.n = n_views;
    for (uint32_t i = 0; i < n_views; i++) {
        vs.v[i] = *views[i];
        vs.mask[i] = masks[i];
        vs.tile_counts[i] = tile_counts[i];
    }
    fill_view_pairs(n_views, views, vs.hot);
    vs.uni = fill_views_uniform(n_views, views);
    // two groups per wave: each group is evaluated n_views times, which hides the load latency that
    // the single-view kernel hides with four groups in flight
    const uint32_t per_block = kWaves * kViewsNG;
    const uint32_t grid = (uint32_t)((s.n_groups + per_block - 1) / per_block);
#define MCULL_LAUNCH_VIEWS(NT_, VP_)                                                                                   \
    hipLaunchKernelGGL((cull_mask_views_kernel<kViewsNG, NT_, VP_>), dim3(grid), dim3(kBlock), 0, st, s.cone_axis_w, s.centre_radius, \
                       s.transforms, s.xf_scale, s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, vs, view_dev, hot_dev,  \
                       reinterpret_cast<const uint2 *>(view_flags), useful, useful_report)
    if (view_dev) {
        if (streaming) MCULL_LAUNCH_VIEWS(true, true);
        else MCULL_LAUNCH_VIEWS(false, true);
    } else {
        if (streaming) MCULL_LAUNCH_VIEWS(true, false);
        else MCULL_LAUNCH_VIEWS(false, false);
    }
#undef MCULL_LAUNCH_VIEWS
    return hipGetLastError();
}

// tile_offsets (one uint64 per tile) may be nullptr for scenes up to kScanMinTiles tiles
// Does the scatter pass leave the tiles without a visible pair to the command pass (which must then be given the tile
// offsets)?  Scenes with tile offsets whose instances span at least 32 groups each.
bool skips_empty_tiles(const SceneDev &s) { return needs_tile_offsets(s.n_groups) && s.n_groups >= 32ull * s.n_instances; }

hipError_t launch_compact(const SceneDev &s, const uint64_t *mask, const uint32_t *tile_counts,
                          uint64_t *tile_offsets, mcull_visible *list, uint64_t capacity, uint32_t *inst_first,
                          uint64_t *total, bool streaming, hipStream_t st, bool short_rounds) {
    const uint32_t tiles = tile_count_for(s.n_groups);
    const uint32_t skip_empty = skips_empty_tiles(s) ? 1u : 0u;
    // One workgroup per tile: splitting a tile over 2 / 4 workgroups (each redoing the offset sum and
    // the tile scan) measured 118 / 152 us against 107 us for the 100 M-pair scene.  A scene with
    // fewer tiles than a quarter of the CUs is latency-bound instead, and four workgroups per tile, one
    // staging round each, halve the pass (the reference's scene, 20 tiles: 5.5 us against 10.4 us).
    const uint32_t split = tiles <= 64u ? 4u : tiles <= 128u ? 2u : 1u;
    const uint32_t grid = tiles * split;
    const unsigned long long *offs = (const unsigned long long *)tile_offsets;
    if (tile_offsets) {
        const uint32_t per_block = kScanBlock * kScanPerThread;
        hipLaunchKernelGGL(scan_tiles_kernel, dim3((tiles + per_block - 1u) / per_block), dim3(kScanBlock), 0, st, tile_counts, tiles,
                           (unsigned long long *)tile_offsets, (unsigned long long *)total);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (skip_empty) {
        // split is 1 here (more than 1024 tiles); a few workgroups per CU's worth of LDS, each walking the tiles with a stride
        const uint32_t walkers = tiles < 8192u ? tiles : 8192u;
        if (streaming)
            hipLaunchKernelGGL((compact_kernel<true, true>), dim3(walkers), dim3(kBlock), 0, st, mask, tile_counts, offs, s.group_rec,
                               s.groups_per_instance, (uint32_t)s.n_groups, list, capacity, inst_first, (unsigned long long *)total, 1u, 1u);
        else
            hipLaunchKernelGGL((compact_kernel<false, true>), dim3(walkers), dim3(kBlock), 0, st, mask, tile_counts, offs, s.group_rec,
                               s.groups_per_instance, (uint32_t)s.n_groups, list, capacity, inst_first, (unsigned long long *)total, 1u, 1u);
        return hipGetLastError();
    }
    static_assert(kShortRoundGroups <= kTileGroups / 4 && kRoundGroups <= kTileGroups, "a round never exceeds what one workgroup writes");
    if (split > 1u) short_rounds = true;  // split <= 4: a workgroup's 64 groups or more hold a short round, not always a long one
    if (short_rounds && streaming)
        hipLaunchKernelGGL((compact_kernel<true, false, true>), dim3(grid), dim3(kBlock), 0, st, mask, tile_counts, offs,
                           s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, list, capacity, inst_first,
                           (unsigned long long *)total, split, skip_empty);
    else if (short_rounds)
        hipLaunchKernelGGL((compact_kernel<false, false, true>), dim3(grid), dim3(kBlock), 0, st, mask, tile_counts, offs,
                           s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, list, capacity, inst_first,
                           (unsigned long long *)total, split, skip_empty);
    else if (streaming)
        hipLaunchKernelGGL(compact_kernel<true>, dim3(grid), dim3(kBlock), 0, st, mask, tile_counts, offs,
                           s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, list, capacity, inst_first,
                           (unsigned long long *)total, split, skip_empty);
    else
        hipLaunchKernelGGL(compact_kernel<false>, dim3(grid), dim3(kBlock), 0, st, mask, tile_counts, offs,
                           s.group_rec, s.groups_per_instance, (uint32_t)s.n_groups, list, capacity, inst_first,
                           (unsigned long long *)total, split, skip_empty);
    return hipGetLastError();
}

// scatter + commands of n_views views of one scene: three launches in all (tile offsets, scatter, commands)
hipError_t launch_ordered_tails(const SceneDev &s, uint32_t uniform_count, uint32_t n_views, const ViewTail *views, bool streaming,
                                hipStream_t st) {
    ViewTails tails;
    std::memset(&tails, 0, sizeof tails);
    for (uint32_t i = 0; i < n_views; i++) tails.t[i] = views[i];
    const uint32_t tiles = tile_count_for(s.n_groups);
    const bool offsets = needs_tile_offsets(s.n_groups), skip_empty = skips_empty_tiles(s);
    const uint32_t split = tiles <= 64u ? 4u : tiles <= 128u ? 2u : 1u;
    if (offsets) {
        const uint32_t per_block = kScanBlock * kScanPerThread;
        hipLaunchKernelGGL(scan_tiles_views_kernel, dim3((tiles + per_block - 1u) / per_block, n_views), dim3(kScanBlock), 0, st, tails, tiles);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const uint32_t gx = skip_empty ? (tiles < 8192u ? tiles : 8192u) : tiles * split;
#define MCULL_LAUNCH_COMPACT_VIEWS(NT_, LOOP_, SHORT_)                                                                             \
    hipLaunchKernelGGL((compact_views_kernel<NT_, LOOP_, SHORT_>), dim3(gx, n_views), dim3(kBlock), 0, st, tails, offsets ? 1u : 0u, s.group_rec, \
                       s.groups_per_instance, (uint32_t)s.n_groups, skip_empty ? 1u : split, skip_empty ? 1u : 0u)
    if (skip_empty) {
        if (streaming) MCULL_LAUNCH_COMPACT_VIEWS(true, true, false);
        else MCULL_LAUNCH_COMPACT_VIEWS(false, true, false);
    } else if (split > 1u) {  // several workgroups per tile: the short rounds (see launch_compact)
        if (streaming) MCULL_LAUNCH_COMPACT_VIEWS(true, false, true);
        else MCULL_LAUNCH_COMPACT_VIEWS(false, false, true);
    } else {
        if (streaming) MCULL_LAUNCH_COMPACT_VIEWS(true, false, false);
        else MCULL_LAUNCH_COMPACT_VIEWS(false, false, false);
    }
#undef MCULL_LAUNCH_COMPACT_VIEWS
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t work = s.n_instances > tiles ? s.n_instances : tiles;
    hipLaunchKernelGGL(emit_commands_views_kernel, dim3((work + kBlock - 1) / kBlock, n_views), dim3(kBlock), 0, st, tails,
                       skip_empty ? 1u : 0u, s.ranges, s.next_nonempty, uniform_count, s.n_instances, tiles, s.group_base,
                       s.groups_per_instance);
    return hipGetLastError();
}

hipError_t launch_emit_commands(const SceneDev &s, uint32_t uniform_count, const uint32_t *inst_first,
                                const uint32_t *inst_count, const uint64_t *total, const uint64_t *rank_totals,
                                uint32_t rank, CmdLayout lay, void *commands, uint32_t *tile_counts, hipStream_t st,
                                const uint64_t *tile_offsets, uint32_t *inst_first_w, uint32_t *work_header) {
    if (!skips_empty_tiles(s)) tile_offsets = nullptr;  // the scatter pass has written every instance's offset itself
    const uint32_t n_tiles = (tile_counts || tile_offsets) ? tile_count_for(s.n_groups) : 0;
    const uint32_t work = s.n_instances > n_tiles ? s.n_instances : n_tiles;
    if (work == 0) return hipSuccess;
    const uint32_t grid = (work + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(emit_commands_kernel, dim3(grid), dim3(kBlock), 0, st, s.ranges, s.next_nonempty, uniform_count, s.n_instances,
                       inst_first, inst_count, total, rank_totals, rank, lay.stride, lay.offset, lay.form,
                       (uint8_t *)commands, tile_counts, n_tiles, (const unsigned long long *)tile_offsets, s.group_base,
                       s.groups_per_instance, inst_first_w, work_header);
    return hipGetLastError();
}

hipError_t launch_cull_atomic(const SceneDev &s, const ViewConst &v, mcull_visible *list, uint64_t capacity,
                              uint32_t *inst_count, uint64_t *total, uint64_t *mask, bool block_mode, hipStream_t st) {
    const uint32_t grid = tile_count_for(s.n_groups);
    if (block_mode) {
        MCULL_DISPATCH_TESTS(v.tests, hipLaunchKernelGGL((cull_atomic_kernel<T, true>), dim3(grid), dim3(kBlock), 0, st,
                                                         s.cone_axis_w, s.centre_radius, s.transforms, s.xf_scale, s.group_rec,
                                                         s.groups_per_instance, (uint32_t)s.n_groups, v, list, capacity,
                                                         inst_count, (unsigned long long *)total, mask));
    } else {
        MCULL_DISPATCH_TESTS(v.tests, hipLaunchKernelGGL((cull_atomic_kernel<T, false>), dim3(grid), dim3(kBlock), 0, st,
                                                         s.cone_axis_w, s.centre_radius, s.transforms, s.xf_scale, s.group_rec,
                                                         s.groups_per_instance, (uint32_t)s.n_groups, v, list, capacity,
                                                         inst_count, (unsigned long long *)total, mask));
    }
    return hipGetLastError();
}

// spheres[i] for ranges[i] (n_ranges of them), or spheres[0] for [0, uniform_count) when ranges == nullptr
// spheres: n_ranges float4 followed by n_ranges float (range_sphere_bytes)
size_t range_sphere_bytes(uint32_t n_ranges) { return range_box_offset_bytes(n_ranges) + (size_t)n_ranges * 16; }

// longest_range: no range holds more meshlets (what picks one wave or one workgroup per range)
hipError_t launch_range_spheres(const float4 *centre_radius, const uint2 *ranges, uint32_t n_ranges, uint32_t uniform_count,
                                uint32_t longest_range, float4 *spheres, hipStream_t st) {
    if (n_ranges == 0) return hipSuccess;
    float *rmax = reinterpret_cast<float *>(spheres + n_ranges);
    float4 *boxes = const_cast<float4 *>(range_boxes(spheres, n_ranges));
    if (n_ranges >= kWaves && longest_range <= kRangePerWaveMax)
        hipLaunchKernelGGL(range_sphere_kernel<true>, dim3((n_ranges + kWaves - 1) / kWaves), dim3(kBlock), 0, st, centre_radius, ranges,
                           uniform_count, 0u, n_ranges, spheres, rmax, boxes);
    else
        hipLaunchKernelGGL(range_sphere_kernel<false>, dim3(n_ranges), dim3(kBlock), 0, st, centre_radius, ranges, uniform_count, 0u,
                           n_ranges, spheres, rmax, boxes);
    return hipGetLastError();
}

// sphere m = the meshlets [m, m + 64) of the n_meshlets uploaded (range_sphere_bytes(n_meshlets) bytes): encloses every
// group that starts at meshlet m
hipError_t launch_group_spheres(const float4 *centre_radius, uint32_t n_meshlets, float4 *spheres, hipStream_t st) {
    if (n_meshlets == 0) return hipSuccess;
    hipLaunchKernelGGL(range_sphere_kernel<true>, dim3((n_meshlets + kWaves - 1) / kWaves), dim3(kBlock), 0, st, centre_radius,
                       (const uint2 *)nullptr, n_meshlets, 1u, n_meshlets, spheres, reinterpret_cast<float *>(spheres + n_meshlets),
                       const_cast<float4 *>(range_boxes(spheres, n_meshlets)));
    return hipGetLastError();
}

size_t work_list_bytes(uint64_t units, bool ranged) { return work_rec_offset_words(units) * 4 + (ranged ? units * 16 : 0); }

// work = work_list_bytes(groups or instances) bytes whose first word is zero (the command pass of the last cull left it so); zero_mask (optional):
// the ballot words, of which the pre-pass zeroes those of the groups it drops.
hipError_t launch_instance_precull(const SceneDev &s, const float4 *spheres, bool per_instance_sphere, const uint32_t *group_base,
                                   const ViewConst &v, const ViewConst *view_dev, uint32_t *work, uint64_t *zero_mask,
                                   bool want_flags, hipStream_t st, uint32_t *next_list) {
    if (s.n_instances == 0) return hipSuccess;
    const float *rmax = reinterpret_cast<const float *>(spheres + (per_instance_sphere ? s.n_instances : 1u));
    const uint32_t per_block = kPreBlock * kPreItems;
    hipLaunchKernelGGL(instance_precull_kernel, dim3((s.n_instances + per_block - 1) / per_block), dim3(kPreBlock), 0, st, s.transforms,
                       spheres, per_instance_sphere ? 1u : 0u, s.ranges, group_base, s.n_instances, s.groups_per_instance,
                       group_base ? s.n_groups : (uint64_t)s.n_instances, v, view_dev, rmax, work, zero_mask, want_flags ? 1u : 0u, next_list);
    return hipGetLastError();
}

// The group level between the instance level and the mask pass: `in` = the instance level's list (with the bits of
// want_flags), `out` = work_list_bytes(n_groups, ranged) bytes whose first word is zero by then (next_list above);
// gspheres = what launch_group_spheres wrote for the n_meshlets uploaded.
hipError_t launch_group_precull(const SceneDev &s, const float4 *gspheres, uint32_t n_meshlets, const ViewConst &v,
                                const ViewConst *view_dev, const uint32_t *in, uint32_t *out, hipStream_t st, uint32_t *in_report) {
    if (s.n_groups == 0) return hipSuccess;
    const bool ranged = s.group_rec != nullptr;
    const float *grmax = reinterpret_cast<const float *>(gspheres + n_meshlets);
    const uint64_t blocks = (s.n_groups + kBlock * kGrpItems - 1) / (kBlock * kGrpItems);
    const uint32_t walkers = blocks < 4096u ? (uint32_t)blocks : 4096u;
    if (ranged)
        hipLaunchKernelGGL(group_precull_kernel<true>, dim3(walkers), dim3(kBlock), 0, st, s.transforms, gspheres, grmax,
                           range_boxes(gspheres, n_meshlets), s.groups_per_instance, s.n_groups, v, view_dev, in, out, in_report);
    else
        hipLaunchKernelGGL(group_precull_kernel<false>, dim3(walkers), dim3(kBlock), 0, st, s.transforms, gspheres, grmax,
                           range_boxes(gspheres, n_meshlets), s.groups_per_instance, s.n_groups, v, view_dev, in, out, in_report);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU, one collective per frame: `gathered` holds, for each of `world` ranks, `rows` {taskCount, firstTask}
// records with rank-local firstTask followed by one record that is the rank's visible count (uint64).  Adds to
// every firstTask the counts of the lower ranks (modulo 2^32, the field's width) and writes the counts and
// their exclusive prefix to totals[0..world) / bases[0..world): one launch instead of a chain of small ones.
__global__ __launch_bounds__(kBlock) void rebase_gathered_kernel(uint2 *__restrict__ gathered, uint32_t world, uint64_t rows,
                                                                 unsigned long long *__restrict__ totals,
                                                                 unsigned long long *__restrict__ bases) {
    const uint32_t r = blockIdx.y;
    unsigned long long base = 0;
    for (uint32_t q = 0; q < r; q++) {
        const uint2 c = gathered[(size_t)q * (rows + 1) + rows];
        base += ((unsigned long long)c.y << 32) | c.x;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint2 c = gathered[(size_t)r * (rows + 1) + rows];
        totals[r] = ((unsigned long long)c.y << 32) | c.x;
        bases[r] = base;
    }
    uint2 *mine = gathered + (size_t)r * (rows + 1);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < rows; i += (uint64_t)gridDim.x * kBlock)
        mine[i].y += (uint32_t)base;
}

hipError_t launch_rebase_gathered(void *gathered, uint32_t world, uint64_t rows, uint64_t *totals, uint64_t *bases, hipStream_t st) {
    if (world == 0) return hipSuccess;
    uint64_t bx = (rows + kBlock * 4 - 1) / (kBlock * 4);
    if (bx == 0) bx = 1;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(rebase_gathered_kernel, dim3((uint32_t)bx, world), dim3(kBlock), 0, st, (uint2 *)gathered, world, rows,
                       (unsigned long long *)totals, (unsigned long long *)bases);
    return hipGetLastError();
}

// n words of zero, as a kernel: a memset node in a captured graph was seen to leave garbage from the second replay
// on (ROCm 7.2), and this is what callers under stream capture reach for.
__global__ __launch_bounds__(kBlock) void zero_words_kernel(uint32_t *__restrict__ p, uint64_t n) {
    // 16-byte stores over the aligned middle, single words at the ragged ends
    const uint64_t head = (4u - ((reinterpret_cast<uintptr_t>(p) >> 2) & 3u)) & 3u;
    const uint64_t h = head < n ? head : n, quads = (n - h) / 4, tail = h + quads * 4;
    uint4 *q = reinterpret_cast<uint4 *>(p + h);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < quads; i += (uint64_t)gridDim.x * kBlock) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0) {
        if (threadIdx.x < h) p[threadIdx.x] = 0u;
        if (tail + threadIdx.x < n) p[tail + threadIdx.x] = 0u;
    }
}

hipError_t launch_zero_words(uint32_t *p, uint64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n / 4 + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(zero_words_kernel, dim3((uint32_t)(blocks < 1 ? 1 : blocks < 8192 ? blocks : 8192)), dim3(kBlock), 0, st, p, n);
    return hipGetLastError();
}

hipError_t launch_xform_scale(const float4 *transforms, uint32_t n, float *xf_scale, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(xform_scale_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, transforms, n, xf_scale);
    return hipGetLastError();
}

hipError_t launch_fill_group_rec(const uint2 *ranges, const uint32_t *group_base, uint32_t n_instances,
                                 uint4 *group_rec, hipStream_t st) {
    if (n_instances == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_group_rec_kernel, dim3((n_instances + kBlock - 1) / kBlock), dim3(kBlock), 0, st, ranges,
                       group_base, n_instances, group_rec);
    return hipGetLastError();
}

hipError_t launch_task_payloads(const SceneDev &s, const uint64_t *mask, mcull_task_payload *out, uint64_t n_payloads,
                                uint32_t payloads_per_instance, hipStream_t st) {
    if (n_payloads == 0) return hipSuccess;
    hipLaunchKernelGGL(task_payload_kernel, dim3((uint32_t)((n_payloads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       mask, s.groups_per_instance, payloads_per_instance, n_payloads, out);
    return hipGetLastError();
}

}  // namespace mcull
