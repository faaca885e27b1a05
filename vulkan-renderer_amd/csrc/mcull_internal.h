// Internal declarations shared by the HIP translation units of libmcull (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <math.h>
#include <vector>
#include <stdint.h>

#include "../../include/mcull.h"

namespace mcull {

// Per-view constants handed to the cull kernels by value (kernarg segment -> SGPRs).
// Everything here is uniform over the launch; it is the device-side image of the
// reference's 212-byte Globals push-constant block (shaders/common.h:7-12) with the two
// uniform expressions of shader.task hoisted to the host: inverse(view)[3].xyz
// (shader.task:88-89) and proj*view (shader.task:60).
struct ViewConst {
    float frustum[5][4];
    float vp[16];        // column-major proj*view
    float cam[3];
    float proj11;        // globals.proj[1][1]
    float tex_max;       // max(textureSize(pyramid, 0))
    uint32_t w0, h0;     // level-0 extent; level k is max(1, w0 >> k) x max(1, h0 >> k) (Engine.cpp:1845-1846)
    uint32_t levels;
    uint32_t tests;      // MCULL_TEST_*
    uint32_t uniform_count;  // meshlets per instance when there are no per-instance ranges
    // device table of 3 x 16 uint32: level widths, level heights, float offset of each level in
    // the flat pyramid (indexed per lane by the selected mip level); word 48: non-zero when a depth image
    // this pyramid slot was built from held a NaN
    const uint32_t *level_table;
    const float *pyramid;
    // largest depth of every 64 x 8 strip of the depth image the pyramid was built from (strips_w per row), or
    // nullptr: lets the occlusion test skip the texel read of a pair that is in front of everything there
    const float *strips;
    uint32_t strips_w;
    uint32_t strips_per_pair;  // the per-pair occlusion test consults them too (memory-bound passes; the instance level always may)
    // sliding-window minima over the pyramid's level kHizWinBase (launch_hiz_windows), or nullptr; win_levels tables.
    // Valid only while word kHizWinWord of level_table is non-zero (every pyramid build clears it).
    const float *windows;
    uint32_t win_levels;
    // the guarded-exact predicates: force_exact != 0 sends every pair through the IEEE expressions (measurements,
    // tests); slow_pairs != nullptr counts the pairs that had to leave the fast forms
    uint32_t force_exact;
    unsigned long long *slow_pairs;
    // what the instance and group levels need of the view alone (view_derive_bounds): an upper bound of each plane
    // normal's length, not below 1, and of the length of each row of proj*view's linear part - evaluated once per view
    // instead of once per instance and view
    float plane_len[5];
    float vp_row_len[4];
};

// plane_len / vp_row_len from frustum / vp.  The factors cover the rounding of the square root on either side.
__host__ __device__ inline void view_derive_bounds(ViewConst &v) {
    for (int p = 0; p < 5; p++) {
        const float nx = v.frustum[p][0], ny = v.frustum[p][1], nz = v.frustum[p][2];
        v.plane_len[p] = fmaxf(sqrtf((nx * nx + ny * ny) + nz * nz) * 1.00001f, 1.0f);
    }
    for (int k = 0; k < 4; k++) {
        const float m0 = v.vp[k], m1 = v.vp[4 + k], m2 = v.vp[8 + k];
        v.vp_row_len[k] = sqrtf((m0 * m0 + m1 * m1) + m2 * m2) * 1.0001f;
    }
}

// What the frustum and cone tests need of TWO views, interleaved: element e of every pair belongs to the view in
// slot 2p + e of a multi-view cull, so that one packed instruction serves both (cull_mask_views_kernel).
struct ViewHot2 {
    float frustum[5][4][2];
    float cam[3][2];
    uint32_t tests[2];
    uint32_t force_exact[2];
    uint32_t pad[2];  // 208 bytes
};

// Scene pointers of one cull launch.
struct SceneDev {
    const float4 *cone_axis_w;    // {axis.xyz, coneW} per meshlet
    const float4 *centre_radius;  // {centre.xyz, radius} per meshlet
    const float4 *transforms;     // 4 x float4 (column-major mat4) per instance
    const float *xf_scale;        // length(model[0].xyz) per instance
    const uint2 *ranges;          // {meshletOffset, meshletCount} per instance, or nullptr
    const uint32_t *next_nonempty;  // with ranges: index of the next instance after i that owns meshlets (n_instances if none)
    const uint32_t *group_base;     // with ranges: index of each instance's first 64-meshlet group (n_instances + 1 entries)
    const uint4 *group_rec;       // {instance, first meshlet, count, local group} per 64-meshlet group, or nullptr
    uint32_t n_instances;
    uint32_t groups_per_instance; // uniform case
    uint64_t n_groups;
};

struct CmdLayout {
    uint32_t stride, offset, form;
};

// The buffers of one view's scatter + command passes (ordered compaction).
struct ViewTail {
    const uint64_t *mask;
    uint32_t *tile_counts;
    uint64_t *tile_offsets;  // used by scenes that need tile offsets (needs_tile_offsets)
    mcull_visible *list;
    uint64_t capacity;
    uint32_t *inst_first;
    uint64_t *total;
    void *commands;
    CmdLayout lay;
};

// ---- launchers (each returns the hipError_t of the launch) ----
uint32_t tile_count_for(uint64_t n_groups);
bool needs_tile_offsets(uint64_t n_groups);  // the scatter pass wants one uint64 list offset per tile (scan_tiles_kernel)
// One pass over the scene for n_views views (ordered path only): views[i] -> masks[i], tile_counts[i].
hipError_t launch_cull_mask_views(const SceneDev &s, uint32_t n_views, const ViewConst *const *views, uint64_t *const *masks,
                                  uint32_t *const *tile_counts, bool streaming, hipStream_t st,
                                  const ViewConst *view_dev = nullptr, const ViewHot2 *hot_dev = nullptr,
                                  const uint8_t *view_flags = nullptr, const uint32_t *useful = nullptr,
                                  uint32_t *useful_report = nullptr);
// view_dev[0..n_views) (device memory) -> hot[0..MCULL_MAX_VIEWS / 2): what launch_cull_mask_views wants as hot_dev
hipError_t launch_pack_view_pairs(const ViewConst *view_dev, uint32_t n_views, ViewHot2 *hot, hipStream_t st);
// work: the instance pre-cull's work list (the pass then covers the listed groups only; the ballots of the others
// must be zero already) or nullptr; kept_report: where the pass leaves the list's length (device pointer to
// host-mapped memory) or nullptr
hipError_t launch_cull_mask(const SceneDev &s, const ViewConst &v, uint64_t *mask, uint32_t *tile_counts,
                            bool streaming, hipStream_t st, const ViewConst *view_dev = nullptr, const uint32_t *work = nullptr,
                            uint32_t *kept_report = nullptr, bool flags = false, bool list_of_groups = false,
                            uint64_t est_groups = 0);  // est_groups: see launch_cull_mask_t (0 = unknown)
size_t work_list_bytes(uint64_t units, bool ranged);  // units = groups (ranged scenes) or instances
size_t range_sphere_bytes(uint32_t n_ranges);  // what launch_range_spheres writes: n float4 spheres, n float largest radii, n float4 boxes
hipError_t launch_range_spheres(const float4 *centre_radius, const uint2 *ranges, uint32_t n_ranges, uint32_t uniform_count,
                                uint32_t longest_range, float4 *spheres, hipStream_t st);
// sphere m encloses the meshlets [m, m + 64), hence every 64-meshlet group that starts at meshlet m (range_sphere_bytes(n_meshlets))
hipError_t launch_group_spheres(const float4 *centre_radius, uint32_t n_meshlets, float4 *spheres, hipStream_t st);
// The single-view pre-pass: `work` = work_list_bytes(...) bytes whose first word is zero (at allocation, and again by
// the command pass of every cull: launch_emit_commands).  zero_mask != nullptr:
// the pre-pass also zeroes the ballot words of the groups it drops.
hipError_t launch_instance_precull(const SceneDev &s, const float4 *spheres, bool per_instance_sphere, const uint32_t *group_base,
                                   const ViewConst &v, const ViewConst *view_dev, uint32_t *work, uint64_t *zero_mask,
                                   bool want_flags, hipStream_t st, uint32_t *next_list = nullptr);
// The group level: thins the instance level's list `in` (built with want_flags) into `out`, whose first word
// launch_instance_precull has zeroed (next_list); uniform scenes get one uint32 per group (launch_cull_mask: list_of_groups).
hipError_t launch_group_precull(const SceneDev &s, const float4 *gspheres, uint32_t n_meshlets, const ViewConst &v,
                                const ViewConst *view_dev, const uint32_t *in, uint32_t *out, hipStream_t st,
                                uint32_t *in_report = nullptr)  /* in_report: receives the number of groups handed in (host-mapped memory) */;
// The pre-pass of a multi-view cull: one byte per instance and view slot (flags[instance * MCULL_MAX_VIEWS + view]:
// bit 0 every meshlet rejected, bit 1 none rejected by the frustum test, bit 2 none by the occlusion test);
// *useful (zeroed by the caller) receives the number of non-zero bytes.
hipError_t launch_classify_views(const SceneDev &s, const float4 *spheres, bool per_instance_sphere, uint32_t n_views,
                                 const ViewConst *const *views, const ViewConst *view_dev, uint8_t *flags, uint32_t *useful,
                                 hipStream_t st);
hipError_t launch_compact(const SceneDev &s, const uint64_t *mask, const uint32_t *tile_counts,
                          uint64_t *tile_offsets, mcull_visible *list, uint64_t capacity, uint32_t *inst_first,
                          uint64_t *total, bool streaming, hipStream_t st, bool short_rounds = false);  // short_rounds: a sparse list is expected
// scatter + commands of n_views (<= MCULL_MAX_VIEWS) views of one scene in three launches; leaves every view's tile counters zero
hipError_t launch_ordered_tails(const SceneDev &s, uint32_t uniform_count, uint32_t n_views, const ViewTail *views, bool streaming,
                                hipStream_t st);
hipError_t launch_emit_commands(const SceneDev &s, uint32_t uniform_count, const uint32_t *inst_first,
                                const uint32_t *inst_count, const uint64_t *total, const uint64_t *rank_totals,
                                uint32_t rank, CmdLayout lay, void *commands, uint32_t *tile_counts, hipStream_t st,
                                const uint64_t *tile_offsets = nullptr, uint32_t *inst_first_w = nullptr,
                                uint32_t *work_header = nullptr);
// work_header: the instance level's work list of this cull (or nullptr); the command pass moves its length to word 1
// and leaves word 0 zero for the next cull's pre-pass
// tile_offsets (with inst_first_w = the same array as inst_first): the scatter pass ran with tile offsets and skipped the
// tiles without a visible pair; the command pass then takes the offsets of the instances that begin in such a tile from
// the tile offsets and stores them into inst_first_w.
hipError_t launch_cull_atomic(const SceneDev &s, const ViewConst &v, mcull_visible *list, uint64_t capacity,
                              uint32_t *inst_count, uint64_t *total, uint64_t *mask, bool block_mode, hipStream_t st);
hipError_t launch_rebase_gathered(void *gathered, uint32_t world, uint64_t rows, uint64_t *totals, uint64_t *bases, hipStream_t st);
hipError_t launch_zero_words(uint32_t *p, uint64_t n_words, hipStream_t st);  // a kernel, not a memset node (see there)
hipError_t launch_xform_scale(const float4 *transforms, uint32_t n, float *xf_scale, hipStream_t st);
hipError_t launch_fill_group_rec(const uint2 *ranges, const uint32_t *group_base, uint32_t n_instances,
                                 uint4 *group_rec, hipStream_t st);
hipError_t launch_task_payloads(const SceneDev &s, const uint64_t *mask, mcull_task_payload *out, uint64_t n_payloads,
                                uint32_t payloads_per_instance, hipStream_t st);

hipError_t launch_split_meshlets(const mcull_meshlet *aos, uint64_t n, float4 *cone_axis_w, float4 *centre_radius,
                                 hipStream_t st);
hipError_t launch_bounds(const mcull_vertex *vertices, const uint32_t *meshlet_data, mcull_meshlet *meshlets,
                         uint64_t n, float4 *cone_axis_w, float4 *centre_radius, hipStream_t st);

// GPU front half of the mesh pipeline (meshlet_kernels.hip); host arrays in, host vectors out.
hipError_t gpu_dedup_vertices(const float *pos, uint64_t npos, const float *nrm, uint64_t nnrm, const float *uv, uint64_t nuv,
                              const int32_t *corners, uint32_t n_corners, std::vector<mcull_vertex> *vertices,
                              std::vector<uint32_t> *indices, hipStream_t st);
hipError_t gpu_build_meshlets(const uint32_t *indices, uint32_t n_tri, std::vector<mcull_meshlet> *meshlets,
                              std::vector<uint32_t> *meshlet_data, hipStream_t st);

// Depth-only software rasteriser over a visible list (depth_kernels.hip); `total` is a device pointer, `mvp` 64 bytes of
// device memory per instance (the launch fills it: (proj*view)*model of every instance); clear: zero the image first.
hipError_t launch_depth_raster(const mcull_visible *list, const uint64_t *total, uint64_t capacity,
                               const mcull_meshlet *meshlets, const uint32_t *meshlet_data, const mcull_vertex *vertices,
                               const float *transforms, uint64_t n_instances, float *mvp, const float vp[16], uint32_t w,
                               uint32_t h, float *depth, bool clear, void *scratch, hipStream_t st, const ViewConst *view_dev = nullptr);
size_t depth_raster_scratch_bytes();  // device memory launch_depth_raster needs as `scratch`

// `counter`: kHizCounterBytes of ticket words, zero before the first launch (each launch leaves them zero).
// `level_table`: kHizTableWords uint32 - 3 x 16 level widths / heights / offsets, then word kHizNanWord: 1 when the depth
// image of the LAST build held a NaN, else 0 (`nan_seen` points at it; the word behind it is the build's scratch, zero
// between builds).
constexpr size_t kHizCounterBytes = 2048;
constexpr uint32_t kHizTableWords = 64, kHizNanWord = 48;
// SLIDING-WINDOW MINIMA for the instance and group levels of a cull (no counterpart in the reference): with C = level
// kHizWinBase of the pyramid (cw x ch cells of 2^kHizWinBase pixels square), table j = 1 .. win_levels holds, for every
// cell (x, y) with x + 2^j <= cw and y + 2^j <= ch, the minimum of the 2^j x 2^j cells whose top-left one it is; row
// length cw, ch rows per table, table j at float offset (j - 1) cw ch.  A rectangle of cells is then covered exactly by a
// few overlapping windows (at most 4 x 4 of side 2^floor(log2(its shorter side)) when its sides differ by less than a
// factor of four), where the pyramid's own coarser levels cover up to twice its extent each way.  Built on demand by a cull
// that wants them; word kHizWinWord of the level table is cleared by every pyramid build and set by the launch that
// finishes the tables, so a cull that finds it zero - a pyramid rebuilt since, e.g. by a replayed graph - does without.
constexpr uint32_t kHizWinBase = 3, kHizWinMaxLevels = 7, kHizWinWord = 51;
uint32_t hiz_window_levels(uint32_t w, uint32_t h);  // 0: the image is too small for any table
size_t hiz_window_floats(uint32_t w, uint32_t h);
// pyramid / level_table of a finished build (stream order); windows: hiz_window_floats(w, h) floats
hipError_t launch_hiz_windows(const float *pyramid, uint32_t *level_table, uint32_t w, uint32_t h, uint64_t base_offset,
                              float *windows, hipStream_t st);
// `strips` (optional): hiz_strip_floats(w, h) floats - the largest depth of every 64 x 8 strip of the image, row by row
// ((w + 63) / 64 per row), then of every 64 x 64 block (same row length; a block is written when its pixels lie inside
// [0, 512 ceil((w >> 3) / 64)) x [0, 512 ceil((h >> 3) / 64)), the cover of the second launch's tiles); written only
// when hiz_writes_strips says so (the streaming scheme, large images).
bool hiz_writes_strips(uint32_t w, uint32_t h, int mode);
size_t hiz_strip_floats(uint32_t w, uint32_t h);
hipError_t launch_hiz(const float *depth, uint32_t w, uint32_t h, float *pyramid, uint32_t levels,
                      const uint32_t *level_table, uint32_t *counter, uint32_t *nan_seen, float *strips, int mode, hipStream_t st);
// n pyramids of one size built with three launches in all (the streaming scheme, its launches batched over the views):
// only where hiz_views_batchable says so; every view has its own kHizCounterBytes of ticket words.
bool hiz_views_batchable(uint32_t w, uint32_t h, uint32_t levels, int mode);
hipError_t launch_hiz_views(uint32_t n, const float *const *depth, uint32_t w, uint32_t h, float *const *pyramid, uint32_t levels,
                            uint32_t *const *level_table, uint32_t *const *counter, float *const *strips, hipStream_t st);


}  // namespace mcull

// Accessors for the sources that work on a context without seeing its definition (mcull_capi.hip).
hipStream_t mcull_ctx_stream(mcull_ctx *ctx);
int mcull_ctx_device(mcull_ctx *ctx);
uint64_t mcull_ctx_instances(mcull_ctx *ctx);
int mcull_ctx_fail(mcull_ctx *ctx, int code, const char *message);
