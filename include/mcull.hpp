// mcull.hpp — C++17 host-side wrapper over the C ABI (mcull.h), header-only.
//
// The method names mirror the reference's Engine methods for this path so that host code written
// against the reference reads the same, and errors behave like the reference's VK_CHECK
// (utils.hpp:68-79): the failure is printed to stderr and a std::runtime_error is thrown.
//
//   reference (Engine.cpp)                     this wrapper
//   createMeshletBuffer          :1427-1453    Culler::createMeshletBuffer
//   createVertexBuffer/MeshletDataBuffer       Culler::createGeometryBuffers
//   buildMeshletCon              :472-547      Culler::buildMeshletCones        (GPU)
//   updateTransforms (memcpy)    :2378         Culler::updateTransforms
//   createHiZPyramid+buildHiZPyramid :1658,:1834  Culler::buildHiZPyramid
//   vkCmdPushConstants + vkCmdDrawMeshTasksIndirectNV :2535-2539   Culler::cull
//   createIndirectBuffer         :1466-1506    written by Culler::cull (dynamic)
//   depth pre-pass               :2397-2441    Culler::rasterizeDepth (software rasteriser over the visible list)
//   createGlobals                :2323-2352    mcull::createGlobals
//   updateTransforms (generation):2353-2377    mcull::referenceTransforms
//   createIndirectBuffer (static):1466-1484    mcull::createIndirectBuffer
//   recordCmdBuffer              :2381-2551    Culler::recordFrame (pre-pass cull -> depth -> pyramid -> main-pass cull)
//   loadModel                    :243-282      Mesh::loadModel
//   optimizeGeometry             :338-471      Mesh::optimizeGeometry
//   createMeshlets               :548-628      Mesh::createMeshlets (host) / Culler::createMeshlets (GPU, with bounds)
//   ... run once, then Mesh::save / Culler::loadScene (cache file)
//   (no counterpart: one device, :658-666)     MultiGpuCuller (one thread, all GPUs) / RankExchange (one process per GPU)
#pragma once

#include <array>
#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "mcull.h"

#define MCULL_CHECK(ctx, expr)                                                                        \
    do {                                                                                              \
        int mcull_status_ = (expr);                                                                   \
        if (mcull_status_ != MCULL_OK) {                                                              \
            std::string mcull_msg_ = std::string("mcull error: ") + mcull_status_string(mcull_status_) + \
                                     " (" + ((ctx) ? mcull_last_error(ctx) : "no context") + ")";    \
            std::cerr << mcull_msg_ << " in " << __FILE__ << " at line " << __LINE__ << std::endl;   \
            throw std::runtime_error(mcull_msg_);                                                     \
        }                                                                                             \
    } while (0)

#define MCULL_GROUP_CHECK(expr)                                                                       \
    do {                                                                                              \
        int mcull_status_ = (expr);                                                                   \
        if (mcull_status_ != MCULL_OK) {                                                              \
            std::string mcull_msg_ = std::string("mcull error: ") + mcull_status_string(mcull_status_) + \
                                     " (" + (grp_ ? mcull_group_last_error(grp_) : "no group") + ")"; \
            std::cerr << mcull_msg_ << " in " << __FILE__ << " at line " << __LINE__ << std::endl;   \
            throw std::runtime_error(mcull_msg_);                                                     \
        }                                                                                             \
    } while (0)

namespace mcull {

using Vertex = mcull_vertex;
using Meshlet = mcull_meshlet;
using Globals = mcull_globals;
using Transform = mcull_transform;
using DrawMeshTasksIndirectCommandNV = mcull_mesh_tasks_cmd;
using DrawMeshTasksIndirectCommandEXT = mcull_mesh_tasks_cmd_ext;
using DrawIndirect = mcull_draw_indirect;
using Visible = mcull_visible;

static_assert(sizeof(Vertex) == 24 && sizeof(Meshlet) == 48 && sizeof(Globals) == 212 && sizeof(Transform) == 64 &&
                  sizeof(DrawMeshTasksIndirectCommandNV) == 8 && sizeof(DrawMeshTasksIndirectCommandEXT) == 12 && sizeof(DrawIndirect) == 28 && sizeof(Visible) == 8,
              "layout contract of shaders/common.h / Engine.hpp");

// Engine::createGlobals (Engine.cpp:2323-2352) for a width x height target; the defaults are the
// reference's camera, field of view and near plane.
inline Globals createGlobals(uint32_t width, uint32_t height, uint32_t meshletCount, const float *eye = nullptr,
                             const float *centre = nullptr, const float *up = nullptr, float fovDeg = 45.0f,
                             float zNear = 0.1f, bool flipY = true) {
    Globals g{};
    MCULL_CHECK(static_cast<mcull_ctx *>(nullptr),
                mcull_create_globals(eye, centre, up, fovDeg, zNear, width, height, flipY ? 1 : 0, meshletCount, &g));
    return g;
}
// The matrices Engine::updateTransforms generates (Engine.cpp:2353-2377; the reference's seed is 42).
inline std::vector<Transform> referenceTransforms(uint32_t drawCount, uint32_t seed = 42) {
    std::vector<Transform> t(drawCount);
    MCULL_CHECK(static_cast<mcull_ctx *>(nullptr), mcull_reference_transforms(drawCount, seed, t.data()));
    return t;
}
// Engine::createIndirectBuffer (Engine.cpp:1466-1484): the static commands of the reference.
inline std::vector<DrawIndirect> createIndirectBuffer(uint32_t drawCount, uint32_t meshletCount, uint32_t indexCount) {
    std::vector<DrawIndirect> d(drawCount);
    MCULL_CHECK(static_cast<mcull_ctx *>(nullptr), mcull_create_indirect_buffer(drawCount, meshletCount, indexCount, d.data()));
    return d;
}

// The four arrays of Engine.hpp:215-218 and the load-time pipeline that fills them.
class Mesh {
public:
    Mesh() = default;
    ~Mesh() { mcull_mesh_destroy(mesh_); }
    Mesh(const Mesh &) = delete;
    Mesh &operator=(const Mesh &) = delete;
    void loadModel(const std::string &path) {
        mcull_mesh_destroy(mesh_);
        mesh_ = nullptr;
        MCULL_CHECK(static_cast<mcull_ctx *>(nullptr), mcull_mesh_load_obj(path.c_str(), &mesh_));
    }
    void optimizeGeometry() { MCULL_CHECK(static_cast<mcull_ctx *>(nullptr), mcull_mesh_optimize(mesh_)); }
    void createMeshlets() { MCULL_CHECK(static_cast<mcull_ctx *>(nullptr), mcull_mesh_build_meshlets(mesh_)); }
    void save(const std::string &path) const { MCULL_CHECK(static_cast<mcull_ctx *>(nullptr), mcull_mesh_save(mesh_, path.c_str())); }
    uint64_t meshletCount() const {
        mcull_scene_counts n{};
        MCULL_CHECK(static_cast<mcull_ctx *>(nullptr), mcull_mesh_counts(mesh_, &n));
        return n.meshlet_count;
    }
    const Meshlet *meshlets() const { return mcull_mesh_meshlets(mesh_); }
    const Vertex *vertices() const { return mcull_mesh_vertices(mesh_); }
    mcull_mesh *handle() const { return mesh_; }

private:
    mcull_mesh *mesh_ = nullptr;
};

class Culler {
public:
    explicit Culler(int device = 0, void *hip_stream = nullptr) { MCULL_CHECK(ctx_, mcull_create(device, hip_stream, &ctx_)); }
    ~Culler() {
        if (ctx_) mcull_destroy(ctx_);
    }
    Culler(const Culler &) = delete;
    Culler &operator=(const Culler &) = delete;

    // createMeshlets on the GPU (meshlets, packed data and bounds), mesh left uploaded
    void createMeshlets(Mesh &mesh) {
        MCULL_CHECK(ctx_, mcull_mesh_build_meshlets_gpu(ctx_, mesh.handle()));
        meshletCount_ = mesh.meshletCount();
    }
    // upload a host-built mesh and complete its bounds on the GPU
    void uploadMesh(Mesh &mesh) {
        MCULL_CHECK(ctx_, mcull_mesh_upload(ctx_, mesh.handle()));
        meshletCount_ = mesh.meshletCount();
    }
    // binding 2 (shader.task:22-24)
    void createMeshletBuffer(const std::vector<Meshlet> &meshlets) {
        meshletCount_ = meshlets.size();
        MCULL_CHECK(ctx_, mcull_upload_meshlets(ctx_, meshlets.data(), meshlets.size(), MCULL_MEM_HOST));
    }
    // bindings 1 and 3 (shader.mesh:27-37)
    void createGeometryBuffers(const std::vector<Vertex> &vertices, const std::vector<uint32_t> &meshletData) {
        MCULL_CHECK(ctx_, mcull_upload_geometry(ctx_, vertices.data(), vertices.size(), meshletData.data(),
                                                meshletData.size(), MCULL_MEM_HOST));
    }
    // Engine::buildMeshletCon for every meshlet, on the GPU; returns the completed records
    std::vector<Meshlet> buildMeshletCones() {
        MCULL_CHECK(ctx_, mcull_compute_bounds(ctx_));
        std::vector<Meshlet> out(meshletCount_);
        MCULL_CHECK(ctx_, mcull_download_meshlets(ctx_, out.data(), out.size()));
        return out;
    }
    // binding 4 (shader.task:18-20)
    void updateTransforms(const std::vector<Transform> &transforms) {
        instances_ = transforms.size();
        MCULL_CHECK(ctx_, mcull_upload_transforms(ctx_, transforms.data(), transforms.size(), MCULL_MEM_HOST));
    }
    void setInstanceRanges(const std::vector<uint32_t> &offset, const std::vector<uint32_t> &count) {
        MCULL_CHECK(ctx_, mcull_set_instance_ranges(ctx_, offset.data(), count.data(), offset.size()));
    }
    // binding 5 (shader.task:26): every level of the pyramid from one read of the depth image
    void buildHiZPyramid(const float *depth, uint32_t width, uint32_t height, uint32_t view = 0, bool on_device = false) {
        MCULL_CHECK(ctx_, mcull_hiz_build(ctx_, view, depth, width, height, on_device ? MCULL_MEM_DEVICE : MCULL_MEM_HOST));
    }
    // the pyramids of several views of one size at once (slots 0 .. depth.size() - 1), images in device memory
    void buildHiZPyramids(const std::vector<const float *> &depth, uint32_t width, uint32_t height) {
        std::vector<uint32_t> views(depth.size());
        for (size_t i = 0; i < views.size(); i++) views[i] = (uint32_t)i;
        MCULL_CHECK(ctx_, mcull_hiz_build_views(ctx_, (uint32_t)depth.size(), views.data(), depth.data(), width, height, MCULL_MEM_DEVICE));
    }
    // the task shader over every (instance, meshlet) pair + global compaction
    mcull_cull_result cull(const Globals &globals, const mcull_cull_params *params = nullptr) {
        mcull_cull_result r{};
        MCULL_CHECK(ctx_, mcull_cull(ctx_, &globals, params, &r));
        return r;
    }
    // several views of the same scene in one pass over the bounds; view i uses output slot i
    std::vector<mcull_cull_result> cullViews(const std::vector<Globals> &globals, const std::vector<mcull_cull_params> &params = {}) {
        std::vector<mcull_cull_result> r(globals.size());
        MCULL_CHECK(ctx_, mcull_cull_views(ctx_, static_cast<uint32_t>(globals.size()), globals.data(),
                                           params.empty() ? nullptr : params.data(), r.data()));
        return r;
    }
    // the same cull for a caller that must not allocate (HIP graph capture, a loop that may never reach hipMalloc):
    // list, commands and total in `params` are the caller's; limits MCULL_STATIC_MAX_GROUPS / _INSTANCES
    mcull_cull_result cullStatic(const Globals &globals, const mcull_cull_params &params) {
        mcull_cull_result r{};
        MCULL_CHECK(ctx_, mcull_cull_static(ctx_, &globals, &params, &r));
        return r;
    }
    void selectResult(uint32_t view) { MCULL_CHECK(ctx_, mcull_select_result(ctx_, view)); }
    // what the instance level of the last cull removed: {marked, tested}, in instances or (scenes with ranges) groups
    struct PrecullStats { uint64_t marked, tested; bool perGroup; };
    PrecullStats precullStats() {
        PrecullStats s{0, 0, false};
        int per_group = 0;
        MCULL_CHECK(ctx_, mcull_precull_stats(ctx_, &s.marked, &s.tested, &per_group));
        s.perGroup = per_group != 0;
        return s;
    }
    // and its group level (instances of 32 groups and more): {dropped, tested} groups
    struct GroupStats { uint64_t dropped, tested; };
    GroupStats precullGroupStats() {
        GroupStats s{0, 0};
        MCULL_CHECK(ctx_, mcull_precull_group_stats(ctx_, &s.dropped, &s.tested));
        return s;
    }
    uint64_t visibleCount() {
        uint64_t n = 0;
        MCULL_CHECK(ctx_, mcull_download_count(ctx_, &n));
        return n;
    }
    std::vector<Visible> visibleList() {
        std::vector<Visible> out(visibleCount());
        uint64_t n = 0;
        MCULL_CHECK(ctx_, mcull_download_list(ctx_, out.data(), out.size(), &n));
        return out;
    }
    std::vector<DrawMeshTasksIndirectCommandNV> indirectCommands() {
        std::vector<DrawMeshTasksIndirectCommandNV> out(instances_);
        MCULL_CHECK(ctx_, mcull_download_commands(ctx_, out.data(), out.size() * sizeof(out[0])));
        return out;
    }
    // depth pre-pass stand-in over the last cull's visible list; returns the device image
    const float *rasterizeDepth(const Globals &globals, uint32_t width, uint32_t height, bool clear = true) {
        MCULL_CHECK(ctx_, mcull_depth_rasterize(ctx_, &globals, width, height, clear ? 1 : 0));
        const float *ptr = nullptr;
        MCULL_CHECK(ctx_, mcull_depth_buffer(ctx_, &ptr, nullptr, nullptr));
        return ptr;
    }
    // The culling work of one recorded frame (Engine::recordCmdBuffer): the pre-pass cull, the depth
    // pre-pass over its survivors, the pyramid built from that depth, and the main-pass cull with all
    // three tests.  The reference's pre-pass runs the same task shader against the pyramid left by the
    // previous frame; pass prepass_tests = MCULL_TEST_ALL for that, the default leaves occlusion out
    // of the pre-pass so that the first frame needs no history.  Returns the main-pass result.
    mcull_cull_result recordFrame(const Globals &globals, uint32_t width, uint32_t height,
                                  uint32_t prepass_tests = MCULL_TEST_FRUSTUM | MCULL_TEST_CONE, uint32_t view = 0) {
        mcull_cull_params p{};
        p.tests = prepass_tests;
        p.hiz_view = view;
        cull(globals, &p);
        const float *depth = rasterizeDepth(globals, width, height);
        buildHiZPyramid(depth, width, height, view, /*on_device=*/true);
        p.tests = MCULL_TEST_ALL;
        return cull(globals, &p);
    }
    // recordFrame with the Globals in device memory: the whole frame reads them when it executes, so it can be
    // captured once into a HIP graph (hipStreamBeginCapture on the context's stream) and replayed every frame
    // after the caller has refreshed the 212 bytes at deviceGlobals - the push constants of a recorded frame.
    mcull_cull_result recordFrameDeviceGlobals(const Globals *deviceGlobals, uint32_t width, uint32_t height,
                                               uint32_t prepass_tests = MCULL_TEST_FRUSTUM | MCULL_TEST_CONE, uint32_t view = 0) {
        mcull_cull_params p{};
        p.tests = prepass_tests;
        p.hiz_view = view;
        mcull_cull_result r{};
        MCULL_CHECK(ctx_, mcull_cull_device_globals(ctx_, deviceGlobals, static_cast<uint32_t>(meshletCount_), &p, &r));
        MCULL_CHECK(ctx_, mcull_depth_rasterize_device_globals(ctx_, deviceGlobals, width, height, 1));
        const float *depth = nullptr;
        MCULL_CHECK(ctx_, mcull_depth_buffer(ctx_, &depth, nullptr, nullptr));
        buildHiZPyramid(depth, width, height, view, /*on_device=*/true);
        p.tests = MCULL_TEST_ALL;
        MCULL_CHECK(ctx_, mcull_cull_device_globals(ctx_, deviceGlobals, static_cast<uint32_t>(meshletCount_), &p, &r));
        return r;
    }
    // scene cache: the arrays of Engine.hpp:215-218 written once, uploaded from the file afterwards
    static void saveScene(const std::string &path, const std::vector<Vertex> &vertices, const std::vector<uint32_t> &indices,
                          const std::vector<uint32_t> &meshletData, const std::vector<Meshlet> &meshlets) {
        MCULL_CHECK(static_cast<mcull_ctx *>(nullptr),
                    mcull_scene_save(path.c_str(), vertices.data(), vertices.size(), indices.data(), indices.size(),
                                     meshletData.data(), meshletData.size(), meshlets.data(), meshlets.size(), nullptr,
                                     nullptr, 0));
    }
    void loadScene(const std::string &path) {
        mcull_scene_counts info{};
        MCULL_CHECK(ctx_, mcull_scene_info(path.c_str(), &info));
        MCULL_CHECK(ctx_, mcull_scene_upload(ctx_, path.c_str()));
        meshletCount_ = info.meshlet_count;
    }
    uint64_t meshletCount() const { return meshletCount_; }
    uint64_t instanceCount() const { return instances_; }
    void waitIdle() { MCULL_CHECK(ctx_, mcull_synchronize(ctx_)); }
    mcull_ctx *handle() const { return ctx_; }

private:
    mcull_ctx *ctx_ = nullptr;
    uint64_t meshletCount_ = 0, instances_ = 0;
};

// One scene over several GPUs from the Engine's single thread (mcull_group_*): the instances are split into
// contiguous, equally sized ranges, one per device ordinal; every device ends up with the DrawIndirect
// commands of the whole scene.  The reference draws on one device (Engine.cpp:658-666).
class MultiGpuCuller {
public:
    explicit MultiGpuCuller(const std::vector<int> &deviceOrdinals) {
        MCULL_GROUP_CHECK(mcull_group_create(deviceOrdinals.data(), (uint32_t)deviceOrdinals.size(), &grp_));
    }
    ~MultiGpuCuller() {
        if (grp_) mcull_group_destroy(grp_);
    }
    MultiGpuCuller(const MultiGpuCuller &) = delete;
    MultiGpuCuller &operator=(const MultiGpuCuller &) = delete;

    uint32_t shardCount() const { return mcull_group_size(grp_); }
    mcull_ctx *shard(uint32_t r) const { return mcull_group_ctx(grp_, r); }
    // first instance of shard r when `instances` are split evenly
    static uint64_t firstInstance(uint64_t instances, uint32_t shards, uint32_t r) { return instances * r / shards; }

    // binding 2 on every device (each shard draws the same meshes)
    void createMeshletBuffer(const std::vector<Meshlet> &meshlets) {
        for (uint32_t r = 0; r < shardCount(); r++)
            MCULL_CHECK(shard(r), mcull_upload_meshlets(shard(r), meshlets.data(), meshlets.size(), MCULL_MEM_HOST));
    }
    // the cached scene on every device
    void loadScene(const std::string &path) {
        for (uint32_t r = 0; r < shardCount(); r++) MCULL_CHECK(shard(r), mcull_scene_upload(shard(r), path.c_str()));
    }
    // binding 4, split into contiguous ranges: evenly by instance count, or - with the number of meshlets each
    // instance draws - so that every shard tests about the same number of meshlet-instances (scenes mixing meshes)
    void updateTransforms(const std::vector<Transform> &transforms, const std::vector<uint32_t> *meshletsPerInstance = nullptr) {
        const uint32_t n = shardCount();
        cuts_.assign(n + 1, 0);
        if (meshletsPerInstance && meshletsPerInstance->size() == transforms.size()) {
            uint64_t total = 0;
            for (uint32_t c : *meshletsPerInstance) total += c;
            uint64_t run = 0, i = 0;
            for (uint32_t r = 1; r < n; r++) {
                while (i < transforms.size() && run < total * r / n) run += (*meshletsPerInstance)[i++];
                cuts_[r] = i;
            }
        } else {
            for (uint32_t r = 1; r < n; r++) cuts_[r] = firstInstance(transforms.size(), n, r);
        }
        cuts_[n] = transforms.size();
        for (uint32_t r = 0; r < n; r++)
            MCULL_CHECK(shard(r), mcull_upload_transforms(shard(r), transforms.data() + cuts_[r], cuts_[r + 1] - cuts_[r], MCULL_MEM_HOST));
        instances_ = transforms.size();
    }
    // per-instance meshlet ranges of the whole scene, cut like the transforms (call after updateTransforms)
    void setInstanceRanges(const std::vector<uint32_t> &offset, const std::vector<uint32_t> &count) {
        if (offset.size() != instances_ || count.size() != instances_ || cuts_.size() != shardCount() + 1)
            throw std::runtime_error("mcull error: ranges must cover the instances given to updateTransforms");
        for (uint32_t r = 0; r < shardCount(); r++)
            MCULL_CHECK(shard(r), mcull_set_instance_ranges(shard(r), offset.data() + cuts_[r], count.data() + cuts_[r], cuts_[r + 1] - cuts_[r]));
    }
    // first instance of every shard, then the instance count
    const std::vector<uint64_t> &cuts() const { return cuts_; }
    void buildHiZPyramid(const float *hostDepth, uint32_t width, uint32_t height, uint32_t view = 0) {
        for (uint32_t r = 0; r < shardCount(); r++)
            MCULL_CHECK(shard(r), mcull_hiz_build(shard(r), view, hostDepth, width, height, MCULL_MEM_HOST));
    }
    // the task-shader work of recordCmdBuffer on every shard + the exchange; results[d] lives on device d
    std::vector<mcull_group_result> cull(const Globals &g, const mcull_cull_params *params = nullptr) {
        std::vector<mcull_group_result> res(shardCount());
        MCULL_GROUP_CHECK(mcull_group_cull(grp_, &g, params, res.data()));
        stride_ = params && params->cmd_stride ? params->cmd_stride : (params && params->cmd_form == MCULL_CMD_EXT_GROUPS ? 12u : 8u);
        return res;
    }
    // Engine::recordCmdBuffer's culling work over the shards: cull, depth pre-pass stand-in per shard, the shards'
    // depth images merged on every device, pyramid, cull with all three tests + the exchange
    std::vector<mcull_group_result> recordFrame(const Globals &g, uint32_t width, uint32_t height,
                                                uint32_t prepass_tests = MCULL_TEST_FRUSTUM | MCULL_TEST_CONE, uint32_t view = 0) {
        std::vector<mcull_group_result> res(shardCount());
        MCULL_GROUP_CHECK(mcull_group_frame(grp_, &g, width, height, prepass_tests, view, nullptr, res.data()));
        stride_ = 8;
        return res;
    }
    // host copy of the scene-wide command records as device `r` holds them
    std::vector<uint8_t> indirectCommands(uint32_t r) {
        std::vector<uint8_t> raw(instances_ * stride_);
        MCULL_GROUP_CHECK(mcull_group_download(grp_, r, raw.data(), raw.size(), nullptr, nullptr));
        return raw;
    }
    // visible counts of the shards, then their sum
    std::vector<uint64_t> visibleCounts(uint32_t r = 0) {
        std::vector<uint64_t> totals(shardCount() + 1);
        MCULL_GROUP_CHECK(mcull_group_download(grp_, r, nullptr, 0, nullptr, totals.data()));
        return totals;
    }
    void waitIdle() { MCULL_GROUP_CHECK(mcull_group_synchronize(grp_)); }
    mcull_group *handle() const { return grp_; }

private:
    mcull_group *grp_ = nullptr;
    uint64_t instances_ = 0;
    uint32_t stride_ = 8;
    std::vector<uint64_t> cuts_;
};

// The other multi-GPU form: one PROCESS per GPU.  Every rank culls its contiguous shard of the instances with a Culler
// of its own and lets the cull write its visible count behind its n commands (mcull_cull_params.total = record n of
// the command buffer); gatherCommands then exchanges those n + 1 records of every rank with one RCCL all-gather and
// rebases the gathered firstTask values on the same stream, so that every rank holds the scene-wide commands.  The
// reference draws on one device (Engine.cpp:658-666).
class RankExchange {
public:
    using UniqueId = std::array<unsigned char, MCULL_RCCL_UNIQUE_ID_BYTES>;
    // rank 0 makes the id and ships it to the other ranks by whatever means the host has (a file, a socket, MPI)
    static UniqueId createId() {
        UniqueId id{};
        check(mcull_rccl_unique_id(id.data()));
        return id;
    }
    // collective: returns when all `world` ranks have called it; makes deviceOrdinal the current device
    RankExchange(const UniqueId &id, int world, int rank, int deviceOrdinal) : world_(world), rank_(rank) {
        check(mcull_rccl_comm_create(id.data(), world, rank, deviceOrdinal, &comm_));
    }
    ~RankExchange() {
        if (comm_) mcull_rccl_comm_destroy(comm_);
    }
    RankExchange(const RankExchange &) = delete;
    RankExchange &operator=(const RankExchange &) = delete;

    int world() const { return world_; }
    int rank() const { return rank_; }
    // first instance of rank r when `instances` are split evenly
    static uint64_t firstInstance(uint64_t instances, int world, int r) { return instances * (uint64_t)r / (uint64_t)world; }

    // sendBlock: this rank's rowsPerRank + 1 records {taskCount, firstTask}, the last one its count; gathered: room for
    // world such blocks; totals / bases: world uint64 each (all device memory of the current device); stream: a
    // hipStream_t.  Two enqueues, no host synchronisation.
    void gatherCommands(void *stream, const void *sendBlock, void *gathered, uint64_t rowsPerRank, uint64_t *totals, uint64_t *bases) {
        check(mcull_allgather_commands(comm_, stream, sendBlock, gathered, (uint32_t)world_, rowsPerRank, totals, bases));
    }
    void *handle() const { return comm_; }

private:
    static void check(int status) {
        if (status == MCULL_OK) return;
        const std::string msg = std::string("mcull error: ") + mcull_status_string(status) + " (" + mcull_collective_last_error() + ")";
        std::cerr << msg << std::endl;
        throw std::runtime_error(msg);
    }
    void *comm_ = nullptr;
    int world_ = 1, rank_ = 0;
};

}  // namespace mcull
