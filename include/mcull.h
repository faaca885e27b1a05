/*
 * mcull.h — C ABI of libmcull, the MI355X (gfx950) implementation of the GPU-driven
 * meshlet culling path of sagizilla-dev/vulkan-renderer.
 *
 * The reference has no FFI: the path sits behind Vulkan descriptor bindings, a
 * 212-byte push-constant block and an indirect buffer.  Every entry point below names
 * the reference interface it stands in for (file:line relative to the reference
 * checkout).  All structs are byte-identical to the reference's GPU-visible structs
 * (shaders/common.h, Engine.hpp) so that a Vulkan frontend can hand over / consume
 * the buffers unchanged.
 *
 * Conventions
 *   - every function returns MCULL_OK (0) or a negative MCULL_E_* code; nothing
 *     throws across the boundary (the reference's VK_CHECK, utils.hpp:68-79, throws;
 *     the C++ wrapper in mcull.hpp restores that behaviour on top of these codes);
 *   - a context is bound to one device and one stream and is not thread-safe; different
 *     contexts share nothing and may be driven from different threads at the same time;
 *   - calls are stream-ordered and asynchronous unless the name says download /
 *     synchronize;
 *   - matrices are column-major float[16] exactly as glm::mat4 stores them:
 *     element (row r, column c) is m[4*c + r].
 */
#ifndef MCULL_H
#define MCULL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes */
#define MCULL_OK 0
#define MCULL_E_INVALID (-1)  /* bad argument                                         */
#define MCULL_E_HIP (-2)      /* a HIP runtime call failed (see mcull_last_error)     */
#define MCULL_E_CAPACITY (-3) /* visible list larger than the output capacity         */
#define MCULL_E_STATE (-4)    /* a required upload / build has not happened yet       */
#define MCULL_E_NOMEM (-5)    /* allocation failed                                    */
#define MCULL_E_IO (-6)       /* a scene file could not be opened, read or written    */

/* ------------------------------------------------------------- layout contract   */

/* shaders/common.h:1-5, Engine.hpp:18-62.  24 bytes. */
typedef struct mcull_vertex {
    float vx, vy, vz, vw;   /* vw is padding                                          */
    uint8_t nx, ny, nz, nw; /* normal, (n*0.5+0.5)*255 truncated                      */
    uint16_t tu, tv;        /* IEEE binary16 bits                                     */
} mcull_vertex;

/* shaders/common.h:26-34, Engine.hpp:72-87.  48 bytes, 16-byte aligned in GLSL. */
typedef struct mcull_meshlet {
    float cone[4];          /* xyz = unit cone axis, w = sin(half angle) or 1.0       */
    float coneApex[4];      /* xyz = vertex centroid = bounding-sphere centre, w = 0  */
    float radius;           /* bounding-sphere radius                                 */
    uint32_t dataOffset;    /* word offset of this meshlet's block in meshletData     */
    uint8_t triangleCount;  /* <= 124                                                 */
    uint8_t vertexCount;    /* <= 64                                                  */
    uint8_t padding[6];
} mcull_meshlet;

/* shaders/common.h:7-12, Engine.hpp:89-95.  212 bytes (the push-constant block). */
typedef struct mcull_globals {
    float view[16];
    float proj[16];
    float frustum[5][4];    /* world-space planes L,R,B,T,near; inside: dot+w >= 0    */
    uint32_t meshletCount;
} mcull_globals;

/* shaders/common.h:14-16, Engine.hpp:97-99.  64 bytes. */
typedef struct mcull_transform {
    float model[16];
} mcull_transform;

/* VkDrawMeshTasksIndirectCommandNV as used at Engine.cpp:1482-1483, :2439, :2539. */
typedef struct mcull_mesh_tasks_cmd {
    uint32_t taskCount;
    uint32_t firstTask;
} mcull_mesh_tasks_cmd;

/* Engine.hpp:102-105: VkDrawIndexedIndirectCommand (20 B) followed by the NV command.
 * The reference binds it with offset 20 / stride 28 (Engine.cpp:2439). */
typedef struct mcull_draw_indirect {
    uint32_t indexCount;
    uint32_t instanceCount;
    uint32_t firstIndex;
    int32_t vertexOffset;
    uint32_t firstInstance;
    mcull_mesh_tasks_cmd commandMeshIndirect;
} mcull_draw_indirect;

/* One entry of the compacted visible list: the union over all task workgroups of the
 * payload {meshletIndices[i], drawID} of shaders/shader.task:28-35, :110-117. */
typedef struct mcull_visible {
    uint32_t drawID;
    uint32_t meshletID;
} mcull_visible;

/* The literal task payload of one 32-wide task workgroup (shader.task:28-35) plus
 * gl_TaskCountNV (shader.task:117); produced by mcull_emit_task_payloads. 136 bytes. */
typedef struct mcull_task_payload {
    uint32_t meshletIndices[32];
    uint32_t drawID;
    uint32_t taskCount;
} mcull_task_payload;

#define MCULL_MAX_HIZ_LEVELS 16
#define MCULL_MAX_VIEWS 8

/* where a source / destination pointer lives */
#define MCULL_MEM_HOST 0
#define MCULL_MEM_DEVICE 1

/* which tests run (shader.task:101 runs all three) */
#define MCULL_TEST_FRUSTUM 1u
#define MCULL_TEST_CONE 2u
#define MCULL_TEST_HIZ 4u
#define MCULL_TEST_ALL 7u

/* compaction strategy */
#define MCULL_COMPACT_ORDERED 0     /* mask -> scan -> scatter; list sorted by (drawID, meshletID),
                                       firstTask = exclusive scan in instance order               */
#define MCULL_COMPACT_WAVE_ATOMIC 1 /* single pass, one atomic per wavefront; list order unspecified;
                                       per-instance taskCount exact, firstTask = 0 for every draw */
#define MCULL_COMPACT_BLOCK_ATOMIC 2 /* single pass, one atomic per workgroup batch; as above     */

/* indirect-command form written by mcull_cull / mcull_rebase_commands.
 * NV_MESHLETS: the reference's record (VkDrawMeshTasksIndirectCommandNV, Engine.hpp:102-105)
 *   {taskCount = visible_i, firstTask} - one mesh workgroup per surviving meshlet, reading
 *   list[firstTask + gl_WorkGroupID.x].  8 bytes.
 * EXT_GROUPS: the three-word VkDrawMeshTasksIndirectCommandEXT form {groupCountX = visible_i,
 *   groupCountY = 1, groupCountZ = 1} (12 bytes) for a frontend on VK_EXT_mesh_shader; that record
 *   has no firstTask, so the mesh stage reads list[first_task[gl_DrawID] + gl_WorkGroupID.x] with
 *   first_task from mcull_cull_result.  The reference itself only uses the NV form.
 * NV_TASK_GROUPS: the NV record with the reference's own granularity, taskCount = ceil(visible_i / 32) task
 *   workgroups of 32 (Engine.cpp:1475-1483 launches ceil(meshletCount / 32) per draw, shader.task:12), for a frontend
 *   that keeps a pass-through task stage: task workgroup w of draw i forwards list[firstTask_i + 32 w + lane] while
 *   32 w + lane < visible_i, where firstTask_i is the draw's offset in the list (the same value as in NV_MESHLETS
 *   records) and visible_i = first_task[i + 1] - first_task[i] from mcull_cull_result.  8 bytes. */
#define MCULL_CMD_NV_MESHLETS 0
#define MCULL_CMD_EXT_GROUPS 1
#define MCULL_CMD_NV_TASK_GROUPS 2

/* mcull_cull_params.flags.
 * Ordered single-view culls (mcull_cull, mcull_cull_device_globals) first test ONE enclosing sphere per
 * instance (of the instance's whole meshlet range, computed once after an upload):
 *  - with the frustum test on, against the five planes, with a margin that covers the rounding of the
 *    per-meshlet expression;
 *  - with the occlusion test on, against the pyramid: the largest depth any meshlet centre of the instance
 *    can have is compared with the minimum of a few texels of one coarser level that cover every texel any
 *    of its meshlets could select (at every level their radii could select), with the same bias.
 * The groups of an instance that one of the two certainly rejects get zero ballots without a meshlet being
 * loaded or tested.  Results are identical with and without it.  The reference has no such level: it
 * launches every task group of every draw (Engine.cpp:1475-1483, shader.task:79-102).
 * The occlusion half relies on the pyramid being the min-reduction of its own level 0 (mcull_hiz_build's
 * output, not a buffer edited afterwards), and on it holding no NaN: min(x, y) = y < x ? y : x can drop a NaN
 * on the way up, so a coarse texel then says nothing about the finer ones.  mcull_hiz_build notices a NaN in
 * the depth image; the occlusion half is then off for that view slot until the slot's pyramid is built again
 * from an image without one (every build re-evaluates it, inside its own launches).  The results are unaffected.
 * The same bounds also show when an instance lies inside all five planes, or in front of everything the
 * pyramid holds under it (largest depth of the image region, from two small tables mcull_hiz_build leaves
 * behind for images from 6 M texels on): the per-meshlet pass then leaves that test out for the instance's
 * meshlets, where its arithmetic is what the pass waits for (scenes whose bounds stay cache-resident).
 * mcull_cull_views has the level in the form of one verdict per instance and view (dropped / no frustum test /
 * no occlusion test) in front of its single pass over the bounds.
 * MCULL_CULL_NO_INSTANCE_PRECULL switches the level off (measurements, parity tests; for mcull_cull_views: when
 * any view's params carry it); the library also drops it by itself for a while after a cull in which it
 * removed, or had a verdict for, less than a tenth of the scene.  mcull_cull_static has no such level.
 * Bounds that were adopted in place (mcull_upload_bounds_soa with adopt) must be announced again with that
 * call after their contents change: the range spheres are derived from them. */
#define MCULL_CULL_NO_INSTANCE_PRECULL 1u
/* The cone and occlusion tests are evaluated in a guarded-exact form: first with the hardware's approximate
 * reciprocal / reciprocal square root and an error bound, and with the correctly rounded IEEE expressions
 * (the ones the oracle evaluates; GLSL itself leaves the precision of /, sqrt, normalize open,
 * shader.task:39, :62-63) only for the pairs whose decision the bound could change.  Every decision, and so
 * every output, is bit-identical to the all-IEEE evaluation.  MCULL_CULL_EXACT_ARITHMETIC sends every pair
 * through the IEEE expressions (measurements, parity tests); MCULL_CULL_COUNT_SLOW_PAIRS counts the pairs
 * that had to (mcull_slow_path_pairs; costs one atomic per affected wave). */
#define MCULL_CULL_EXACT_ARITHMETIC 2u
#define MCULL_CULL_COUNT_SLOW_PAIRS 4u

/* VkDrawMeshTasksIndirectCommandEXT as written for MCULL_CMD_EXT_GROUPS. 12 bytes. */
typedef struct mcull_mesh_tasks_cmd_ext {
    uint32_t groupCountX, groupCountY, groupCountZ;
} mcull_mesh_tasks_cmd_ext;

typedef struct mcull_cull_params {
    uint32_t tests;          /* MCULL_TEST_* bits                                      */
    uint32_t compaction;     /* MCULL_COMPACT_*                                        */
    uint32_t hiz_view;       /* which pyramid slot the Hi-Z test samples               */
    uint32_t cmd_form;       /* MCULL_CMD_NV_MESHLETS, _EXT_GROUPS or _NV_TASK_GROUPS  */
    uint32_t cmd_stride;     /* bytes between commands; 0 -> the record size (8 / 12)  */
    uint32_t cmd_offset;     /* byte offset of the command in a record; 28/20 = DrawIndirect */
    uint32_t write_mask;     /* non-zero: keep the 1-bit-per-pair visibility mask      */
    uint32_t flags;          /* MCULL_CULL_* bits, 0 for the defaults                  */
    /* optional caller-owned device outputs (NULL -> context-owned buffers)            */
    mcull_visible *list;     /* capacity list_capacity entries                         */
    uint64_t list_capacity;
    void *commands;          /* n_instances records of cmd_stride bytes                */
    uint64_t *total;         /* one uint64: the visible count (e.g. the send buffer of an all-gather) */
} mcull_cull_params;

typedef struct mcull_cull_result {
    const mcull_visible *list;   /* device pointer                                     */
    const void *commands;        /* device pointer, cmd_stride * n_instances bytes     */
    const uint64_t *total;       /* device pointer to the visible count                */
    const uint64_t *mask;        /* device pointer, one 64-bit word per 64-meshlet group, or NULL */
    uint64_t n_groups;           /* number of 64-meshlet groups (mask words)           */
    uint64_t n_pairs;            /* meshlet-instances tested                           */
    const uint32_t *first_task;  /* device pointer, one list offset per instance (defined for instances
                                    that own meshlets), ordered compaction only, else NULL; rank-local */
} mcull_cull_result;

typedef struct mcull_ctx mcull_ctx;

/* ------------------------------------------------------------------ lifecycle    */

/* Replaces the Vulkan bring-up the path depends on (Engine.cpp:14-53: device, queues,
 * command pools).  `stream` is a hipStream_t (NULL = the device's null stream). */
int mcull_create(int device_ordinal, void *stream, mcull_ctx **out);
int mcull_destroy(mcull_ctx *ctx);
/* Last error text of this context (VK_CHECK's stderr line, utils.hpp:72-74). */
const char *mcull_last_error(const mcull_ctx *ctx);
/* Text for a status code; usable without a context. */
const char *mcull_status_string(int status);
/* vkQueueWaitIdle equivalent for the context's stream. */
int mcull_synchronize(mcull_ctx *ctx);
/* Library version: major*10000 + minor*100 + patch. */
int mcull_version(void);

/* ------------------------------------------------------------------ uploads      */

/* createMeshletBuffer (Engine.cpp:1427-1453; binding 2, shader.task:22-24): takes the
 * reference's 48-byte AoS records, keeps an AoS copy and splits the culling fields into
 * two SoA float4 streams {axis.xyz, coneW} and {centre.xyz, radius}. */
int mcull_upload_meshlets(mcull_ctx *ctx, const mcull_meshlet *meshlets, uint64_t count, int mem);

/* Direct SoA upload for scenes generated on the device (no 48-byte records exist):
 * cone_axis_w[i] = {axis.xyz, coneW}, centre_radius[i] = {centre.xyz, radius}.
 * With mem == MCULL_MEM_DEVICE and adopt != 0 the pointers are used in place (the caller
 * keeps them alive); otherwise they are copied.  No reference counterpart (the reference
 * only ever uploads AoS records, Engine.cpp:1442). */
int mcull_upload_bounds_soa(mcull_ctx *ctx, const float *cone_axis_w, const float *centre_radius,
                            uint64_t count, int mem, int adopt);

/* createVertexBuffer + createMeshletDataBuffer (Engine.cpp:1360-1426; bindings 1 and 3,
 * shader.mesh:27-37): inputs of the bounds computation. */
int mcull_upload_geometry(mcull_ctx *ctx, const mcull_vertex *vertices, uint64_t vertex_count,
                          const uint32_t *meshlet_data, uint64_t word_count, int mem);

/* buildMeshletCon (Engine.cpp:472-547) for every uploaded meshlet, on the GPU: rewrites
 * cone, coneApex and radius of the context's AoS records and refreshes the SoA streams. */
int mcull_compute_bounds(mcull_ctx *ctx);

/* Reads the context's AoS records back (after mcull_compute_bounds). Synchronous. */
int mcull_download_meshlets(mcull_ctx *ctx, mcull_meshlet *out, uint64_t count);

/* createTransformBuffers + the memcpy of updateTransforms (Engine.cpp:1454-1465, :2378;
 * binding 4, shader.task:18-20). */
int mcull_upload_transforms(mcull_ctx *ctx, const mcull_transform *transforms, uint64_t count, int mem);

/* Per-instance meshlet range [offset, offset+count) into the uploaded meshlet array.
 * NULL pointers restore the reference behaviour: every instance draws
 * [0, globals.meshletCount) (shader.task:85).  Host pointers.  Extension needed for
 * scenes that mix meshes; no reference counterpart. */
int mcull_set_instance_ranges(mcull_ctx *ctx, const uint32_t *meshlet_offset,
                              const uint32_t *meshlet_count, uint64_t n_instances);

/* ------------------------------------------------------------------ Hi-Z pyramid */

/* hiZMipLevels (Engine.cpp:1651): floor(log2(max(w,h))) + 1. */
uint32_t mcull_hiz_level_count(uint32_t width, uint32_t height);
/* Level extents max(1, w>>i) x max(1, h>>i) (Engine.cpp:1845-1846) and float offsets of
 * each level inside the flat pyramid buffer. Returns the total float count. */
uint64_t mcull_hiz_layout(uint32_t width, uint32_t height, uint32_t *levels,
                          uint32_t level_width[MCULL_MAX_HIZ_LEVELS],
                          uint32_t level_height[MCULL_MAX_HIZ_LEVELS],
                          uint64_t level_offset[MCULL_MAX_HIZ_LEVELS]);

/* createHiZPyramid + buildHiZPyramid + shaders/shader.comp (Engine.cpp:1658-1685,
 * :1834-1866; shader.comp:12-31): level 0 = copy of depth, level i = 2x2 min of level
 * i-1; the depth image is read once and every level written once (two launches, where the
 * reference dispatches once per level).  `view` selects one of MCULL_MAX_VIEWS pyramid slots. */
int mcull_hiz_build(mcull_ctx *ctx, uint32_t view, const float *depth, uint32_t width,
                    uint32_t height, int mem);
/* The same for n_views depth images of ONE size (the shadow cascades or cameras of mcull_cull_views), image i into slot
 * views[i] (distinct slots): exactly what n_views calls of mcull_hiz_build give, but large images in device memory are built
 * with three launches in all instead of two per view (the reference would record buildHiZPyramid once per view,
 * Engine.cpp:1834-1866; it has one view). */
int mcull_hiz_build_views(mcull_ctx *ctx, uint32_t n_views, const uint32_t *views, const float *const *depth,
                          uint32_t width, uint32_t height, int mem);
/* Device pointer of a slot's flat pyramid (layout as mcull_hiz_layout). */
int mcull_hiz_pyramid(mcull_ctx *ctx, uint32_t view, const float **device_ptr, uint64_t *float_count);
/* Copies one level to the host. Synchronous. */
int mcull_hiz_download_level(mcull_ctx *ctx, uint32_t view, uint32_t level, float *out);

/* ------------------------------------------------------------------ cull+compact */

/* vkCmdPushConstants(Globals) + vkCmdDrawMeshTasksIndirectNV (Engine.cpp:2535-2539) and
 * the task shader it launches (shaders/shader.task:37-119): tests every
 * (instance, meshlet) pair, compacts the survivors into one global visible list and
 * writes one indirect command per instance (createIndirectBuffer, Engine.cpp:1466-1484,
 * made dynamic).  `params` may be NULL (all tests, ordered compaction, 8-byte commands).
 * Returns MCULL_E_CAPACITY only from the download/count calls, never asynchronously (list larger than its
 * capacity, or 2^32 visible pairs and more: see mcull_download_count). */
int mcull_cull(mcull_ctx *ctx, const mcull_globals *globals, const mcull_cull_params *params,
               mcull_cull_result *result);

/* mcull_cull for callers that cannot allocate device memory: code under HIP graph capture, tight frame
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same kernels and results as
 * mcull_cull with ordered compaction, but the call performs no hipMalloc / hipFree: `params` must bring the
 * caller-owned device outputs list (+ list_capacity), commands and total, and the intermediate buffers
 * (visibility ballots, tile counters, per-instance list offsets) live in a static device array of
 * MCULL_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.
 * Limits: at most MCULL_STATIC_MAX_GROUPS 64-meshlet groups (134 M meshlet-instances) and
 * MCULL_STATIC_MAX_INSTANCES instances, else MCULL_E_CAPACITY.  All contexts of one GPU share that array, so
 * calls must not overlap on one GPU: a call made while an earlier one on ANOTHER stream of the same GPU has not
 * finished returns MCULL_E_STATE and launches nothing (the guard is a recorded event; it cannot see calls made
 * under stream capture, whose overlap is the caller's to exclude).  params->write_mask and
 * MCULL_CULL_COUNT_SLOW_PAIRS are not supported here (MCULL_E_INVALID).  result->mask / first_task point into
 * the array and stay valid only until the next mcull_cull_static on that GPU.  The context's stored result (what the
 * download / rebase / payload / rasterise calls refer to) is left untouched. */
#define MCULL_STATIC_MAX_GROUPS 2097152ull
#define MCULL_STATIC_MAX_INSTANCES 1048576ull
#define MCULL_STATIC_SCRATCH_BYTES 21070592ull
int mcull_cull_static(mcull_ctx *ctx, const mcull_globals *globals, const mcull_cull_params *params,
                      mcull_cull_result *result);

/* mcull_cull for up to MCULL_MAX_VIEWS views of the same scene in one pass over the meshlet bounds
 * (the cascade / multi-camera case: the reference records one cull per render pass, Engine.cpp:2381-2551,
 * and has no multi-view form).  globals[i] / params[i] (params may be NULL: all tests, ordered) /
 * results[i] describe view i, which uses output slot i of the context; every view's outputs equal
 * those of a separate mcull_cull call.  Ordered compaction only, and globals[i].meshletCount must
 * be the same for all views.  Afterwards the download / rebase / payload / rasterise calls refer to
 * view 0 until mcull_select_result picks another slot; mcull_cull always uses and selects slot 0. */
int mcull_cull_views(mcull_ctx *ctx, uint32_t n_views, const mcull_globals *globals, const mcull_cull_params *params,
                     mcull_cull_result *results);
int mcull_select_result(mcull_ctx *ctx, uint32_t view);

/* mcull_cull with the Globals in DEVICE memory: they are read by a kernel when the stream reaches the
 * call, not by the host when the call is made (the reference pushes its Globals per frame with
 * vkCmdPushConstants, Engine.cpp:2429-2437, :2535; this is the form of that for a recorded command
 * stream).  A sequence captured into a HIP graph therefore replays with whatever the buffer holds at
 * replay time: the camera can move without re-capturing.  The host cannot see globals.meshletCount,
 * so meshlet_count is passed separately (ignored when instance ranges are set).  Ordered compaction
 * only.  Results are bit-identical to mcull_cull on the same Globals. */
int mcull_cull_device_globals(mcull_ctx *ctx, const mcull_globals *device_globals, uint32_t meshlet_count,
                              const mcull_cull_params *params, mcull_cull_result *result);

/* mcull_cull_views with the n_views Globals in one device array (212-byte records, no padding): the cascade
 * form of mcull_cull_device_globals, same rules. */
int mcull_cull_views_device_globals(mcull_ctx *ctx, uint32_t n_views, const mcull_globals *device_globals,
                                    uint32_t meshlet_count, const mcull_cull_params *params, mcull_cull_result *results);

/* Multi-GPU: re-emits the indirect commands of the last mcull_cull with
 * firstTask += sum(totals[0..rank)), where `totals` is the device array produced by an
 * all-gather of every rank's *result.total.  No reference counterpart (single GPU). */
int mcull_rebase_commands(mcull_ctx *ctx, const uint64_t *device_totals, uint32_t rank);

/* Multi-GPU with ONE collective per frame: every rank lets its cull write the visible count into one extra
 * record behind its n commands (mcull_cull_params.total = address of record n of the command buffer) and
 * all-gathers the n + 1 records.  This call then turns the gathered buffer - `world` blocks of
 * rows_per_rank {taskCount, firstTask} records with rank-local firstTask, each followed by its count record -
 * into the global description on the device, in one launch on `stream` (a hipStream_t; the call needs no
 * context: the buffer belongs to the caller's collective): firstTask += the counts of the lower ranks
 * (modulo 2^32, the field's width), totals[r] = rank r's count, bases[r] = the sum of the counts below r
 * (device arrays of `world` uint64).  The device must be current.  No reference counterpart (single GPU). */
int mcull_rebase_gathered_commands(void *stream, void *gathered, uint32_t world, uint64_t rows_per_rank, uint64_t *totals,
                                   uint64_t *bases);

/* The same step with the collective inside: all-gathers this rank's block (rows_per_rank + 1 records, as above) from
 * `send_block` into `gathered` (`world` blocks; device memory, the caller's) with RCCL's ncclAllGather on `stream`,
 * then rebases the gathered firstTask values on that stream as mcull_rebase_gathered_commands does.  `comm` is an
 * ncclComm_t the caller owns (one process per GPU; any way of creating it will do - mcull_rccl_comm_create below
 * is one).  Asynchronous: two enqueues, no host synchronisation, capturable.  RCCL is looked up at run time, not
 * linked: the copy the process already holds (PyTorch's) is reused, else the system's librccl.so.1 is loaded;
 * MCULL_E_STATE when there is none (mcull_collective_last_error() says why, per thread).  No reference counterpart:
 * the reference drives one device (Engine.cpp:658-666). */
int mcull_allgather_commands(void *nccl_comm, void *stream, const void *send_block, void *gathered, uint32_t world,
                             uint64_t rows_per_rank, uint64_t *totals, uint64_t *bases);

/* Communicator plumbing for hosts that have no other RCCL binding: rank 0 calls mcull_rccl_unique_id, ships the
 * MCULL_RCCL_UNIQUE_ID_BYTES bytes to the other ranks by whatever means it has (a file, a socket, a
 * torch.distributed broadcast), and every rank calls mcull_rccl_comm_create (collective: returns when all `world`
 * ranks have called it; makes device_ordinal current).  No reference counterpart. */
#define MCULL_RCCL_UNIQUE_ID_BYTES 128
int mcull_rccl_unique_id(void *id_128_bytes);
int mcull_rccl_comm_create(const void *id_128_bytes, int world, int rank, int device_ordinal, void **nccl_comm);
int mcull_rccl_comm_destroy(void *nccl_comm);
const char *mcull_collective_last_error(void);

/* The literal per-32 task payloads of shader.task:28-35,:103-118 for the last ordered
 * mcull_cull: one record per (instance, group of 32 meshlets).  `out` is a device buffer
 * of at least mcull_task_payload_count() records. */
uint64_t mcull_task_payload_count(const mcull_ctx *ctx);
int mcull_emit_task_payloads(mcull_ctx *ctx, mcull_task_payload *out);

/* Synchronous read-backs of the last mcull_cull.  firstTask and the list offsets behind it are 32-bit
 * (the NV record's field width): when an ordered cull keeps 2^32 pairs or more, all three calls return
 * MCULL_E_CAPACITY (mcull_download_count still stores the exact 64-bit count in *total) instead of
 * handing out wrapped offsets; split such a scene over several culls or GPUs.  A caller that consumes the
 * buffers on the device without these calls (mcull_cull_static, graph replay) checks *result.total itself. */
int mcull_download_count(mcull_ctx *ctx, uint64_t *total);
/* MCULL_E_CAPACITY (and *total = required entries) when the list did not fit. */
int mcull_download_list(mcull_ctx *ctx, mcull_visible *out, uint64_t capacity, uint64_t *total);
int mcull_download_commands(mcull_ctx *ctx, void *out, uint64_t bytes);
int mcull_download_mask(mcull_ctx *ctx, uint64_t *out, uint64_t words);

/* Pairs of the last cull (run with MCULL_CULL_COUNT_SLOW_PAIRS) that left the fast predicate forms for the
 * IEEE expressions, counted once per test that sent them there.  Synchronous.  No reference counterpart. */
int mcull_slow_path_pairs(mcull_ctx *ctx, uint64_t *pairs);

/* What the instance pre-cull of the last cull removed (synchronous: reads the length of the work list the
 * pre-cull left behind).  *marked = instances marked as certainly outside the frustum or certainly occluded (scenes
 * without instance ranges) or 64-meshlet groups of such instances (scenes with ranges: *per_group != 0);
 * *tested = how many instances / groups there were.  0 / 0 when the pre-cull did not run (frustum and occlusion
 * tests both off, switched off or backed off, scene below 65536 groups, unordered compaction, one of several
 * views).  No reference counterpart. */
int mcull_precull_stats(mcull_ctx *ctx, uint64_t *marked, uint64_t *tested, int *per_group);

/* The same for the GROUP level, which follows the instance level in scenes whose instances span 32 and more
 * 64-meshlet groups with bounds small enough to stay in the caches (one large mesh drawn many times): every group of the
 * surviving instances is tested once more with an enclosing sphere of its own, in the same two ways, before any of
 * its meshlets is loaded.  *tested = groups of the instances the instance level kept, *dropped = those of them the
 * group level removed; 0 / 0 when it did not run.  Synchronous.  No reference counterpart. */
int mcull_precull_group_stats(mcull_ctx *ctx, uint64_t *dropped, uint64_t *tested);

/* Device timestamps around the kernels of each mcull_cull / mcull_hiz_build /
 * mcull_compute_bounds (HIP events on the context's stream; the reference's timestamp
 * queries, Engine.cpp:2393-2394, :2548).  Off by default: the two event records cost a few
 * microseconds of stream time per call. */
int mcull_enable_timing(mcull_ctx *ctx, int enable);
/* Milliseconds between the two timestamps of the last timed call. Synchronous. */
int mcull_last_kernel_ms(mcull_ctx *ctx, float *ms);

/* ------------------------------------------------------------------ depth source */

/* Depth pre-pass stand-in (Engine.cpp:2397-2441 running shader.mesh:60-101 on the hardware
 * rasteriser): rasterises every triangle of every meshlet in the LAST cull's visible list into a
 * context-owned width x height float image, reverse-Z, nearest (largest) depth per pixel
 * (Engine.cpp:975 GREATER_OR_EQUAL, clear 0 Engine.cpp:2404), back faces culled (Engine.cpp:961-962).
 * `globals` supplies view/proj (gl_Position = proj*view*model*v, shader.mesh:73); the viewport is
 * (0,0,width,height), depth 0..1 (Engine.cpp:2413-2420).  clear != 0 (or a size change) clears the
 * image first; clear == 0 accumulates into the previous image.  Needs mcull_upload_meshlets and
 * mcull_upload_geometry (not SoA-only scenes), and the transforms the cull ran on: MCULL_E_STATE when a
 * different number of them has been uploaded since.  The image feeds mcull_hiz_build(..., MCULL_MEM_DEVICE):
 * cull -> rasterise -> pyramid -> cull with MCULL_TEST_HIZ is the reference's two-phase frame
 * (Engine.cpp:2381-2551) without the raster hardware.  Stream-ordered. */
int mcull_depth_rasterize(mcull_ctx *ctx, const mcull_globals *globals, uint32_t width, uint32_t height, int clear);
/* Device pointer and size of the image (valid until the next mcull_depth_rasterize with another size). */
/* The same with the Globals in device memory (see mcull_cull_device_globals). */
int mcull_depth_rasterize_device_globals(mcull_ctx *ctx, const mcull_globals *device_globals, uint32_t width, uint32_t height,
                                         int clear);
int mcull_depth_buffer(mcull_ctx *ctx, const float **device_ptr, uint32_t *width, uint32_t *height);
/* Host copy, width*height floats. */
int mcull_depth_download(mcull_ctx *ctx, float *out);

/* ------------------------------------------------------------------ scene cache  */

/* A scene file is the on-disk form of the arrays the path consumes - Engine.hpp:215-218
 * (vertices, indices, meshlets, meshletData) plus the per-instance meshlet ranges - so the
 * load-time pipeline of Engine.cpp:243-628 (loadModel, optimizeGeometry, createMeshlets) runs once
 * per asset.  The reference has no such cache: it rebuilds everything at every start
 * (Engine.cpp:16-17).  Little-endian, checksummed (FNV-1a-64 over the payload); these calls are
 * host code and need no context or GPU, except mcull_scene_upload. */
typedef struct mcull_scene_counts {
    uint64_t vertex_count, index_count, meshlet_data_words, meshlet_count, range_count;
    uint64_t checksum;
} mcull_scene_counts;

/* Any array may be empty (count 0).  MCULL_E_IO when the file cannot be written (a partial file
 * is removed). */
int mcull_scene_save(const char *path, const mcull_vertex *vertices, uint64_t vertex_count,
                     const uint32_t *indices, uint64_t index_count, const uint32_t *meshlet_data,
                     uint64_t meshlet_data_words, const mcull_meshlet *meshlets, uint64_t meshlet_count,
                     const uint32_t *range_offset, const uint32_t *range_count, uint64_t n_ranges);
/* Header only: the counts a caller needs to size its arrays.  MCULL_E_IO / MCULL_E_INVALID (not
 * a scene file, unknown version, inconsistent sizes, a file shorter or longer than its header says). */
int mcull_scene_info(const char *path, mcull_scene_counts *out);
/* Reads every section into caller-allocated arrays sized from mcull_scene_info (a NULL pointer skips that
 * section).  `expected` is what mcull_scene_info returned when the arrays were sized: a file whose counts or
 * checksum differ from it by now (replaced in between) is refused with MCULL_E_INVALID before anything is
 * written.  MCULL_E_INVALID also when the payload's checksum does not match (the outputs then hold unverified
 * bytes) or the file is not as long as its header says, MCULL_E_IO on a short read. */
int mcull_scene_load(const char *path, const mcull_scene_counts *expected, mcull_vertex *vertices, uint32_t *indices,
                     uint32_t *meshlet_data, mcull_meshlet *meshlets, uint32_t *range_offset, uint32_t *range_count);
/* mcull_scene_load + mcull_upload_meshlets + mcull_upload_geometry + mcull_set_instance_ranges
 * (ranges are cleared when the file holds none). */
int mcull_scene_upload(mcull_ctx *ctx, const char *path);

/* ------------------------------------------------------------------ mesh pipeline */

/* The reference's load-time pipeline, host code as it is there: loadModel (Engine.cpp:243-282),
 * optimizeGeometry (Engine.cpp:338-471), createMeshlets (Engine.cpp:548-628).  A mcull_mesh owns the
 * four arrays of Engine.hpp:215-218.  Meshlets are built with zero cone/apex/radius; mcull_mesh_upload
 * computes those on the GPU (buildMeshletCon, Engine.cpp:472-547) and writes them back into the mesh.
 * These calls need no context or GPU, except mcull_mesh_upload. */
typedef struct mcull_mesh mcull_mesh;

/* tinyobj::LoadObj + the dedup/quantise loop (Engine.cpp:250-281).  MCULL_E_IO when the file cannot
 * be opened, MCULL_E_INVALID for an index outside its array. */
int mcull_mesh_load_obj(const char *path, mcull_mesh **out);
/* The same loop from arrays: positions xyz, optional normals xyz, optional texcoords uv, and one
 * (position, texcoord, normal) index triple per face corner (-1 = absent), three corners per triangle. */
int mcull_mesh_from_arrays(const float *positions, uint64_t n_positions, const float *normals, uint64_t n_normals,
                           const float *texcoords, uint64_t n_texcoords, const int32_t *corners, uint64_t n_corners,
                           mcull_mesh **out);
/* optimizeGeometry: reorders the index buffer in place (same order as the reference's algorithm). */
int mcull_mesh_optimize(mcull_mesh *mesh);
/* createMeshlets: fills meshlets + meshletData from the current index order. */
int mcull_mesh_build_meshlets(mcull_mesh *mesh);
/* Sizes (range_count and checksum are zero) and read-only views of the arrays; the pointers stay
 * valid until the next call that modifies or destroys the mesh. */
int mcull_mesh_counts(const mcull_mesh *mesh, mcull_scene_counts *out);
const mcull_vertex *mcull_mesh_vertices(const mcull_mesh *mesh);
const uint32_t *mcull_mesh_indices(const mcull_mesh *mesh);
const uint32_t *mcull_mesh_meshlet_data(const mcull_mesh *mesh);
const mcull_meshlet *mcull_mesh_meshlets(const mcull_mesh *mesh);
/* createMeshletBuffer + createVertexBuffer + createMeshletDataBuffer, then the GPU bounds kernel;
 * the completed meshlet records are copied back into the mesh. */
int mcull_mesh_upload(mcull_ctx *ctx, mcull_mesh *mesh);
/* GPU variants with identical results: the dedup loop as a parallel hash-table pass (lowest corner
 * index per distinct vertex + a scan for the first-seen numbering), and createMeshlets as one thread
 * per candidate start triangle + the chain of real starts; mcull_mesh_build_meshlets_gpu also uploads
 * the mesh and completes the bounds like mcull_mesh_upload.  (The reorderer stays on the host: it is
 * inherently sequential.) */
int mcull_mesh_from_arrays_gpu(mcull_ctx *ctx, const float *positions, uint64_t n_positions, const float *normals,
                               uint64_t n_normals, const float *texcoords, uint64_t n_texcoords, const int32_t *corners,
                               uint64_t n_corners, mcull_mesh **out);
int mcull_mesh_build_meshlets_gpu(mcull_ctx *ctx, mcull_mesh *mesh);
/* Writes the mesh as a scene cache file (no instance ranges). */
int mcull_mesh_save(const mcull_mesh *mesh, const char *path);
void mcull_mesh_destroy(mcull_mesh *mesh);

/* ------------------------------------------------------------------ per-frame host inputs */

/* Host code (no context, no GPU): the buffers the reference's Engine prepares on the CPU and hands
 * to the path.  IEEE binary32 in glm's association order.
 *
 * createGlobals (Engine.cpp:2323-2352): lookAt view, infinite-far reverse-Z projection with
 * proj[1][1] negated when flip_y != 0 (Engine.cpp:2335), meshletCount, and the five normalised
 * frustum planes.  NULL eye/centre/up select the reference's camera: (2,2,2) -> origin, +Z up; the
 * reference's remaining values are fov_deg 45, z_near 0.1, flip_y 1. */
int mcull_create_globals(const float eye[3], const float centre[3], const float up[3], float fov_deg, float z_near,
                         uint32_t width, uint32_t height, int flip_y, uint32_t meshlet_count, mcull_globals *out);
/* The plane extraction alone (Engine.cpp:2337-2349) from the view and proj already stored in g. */
int mcull_extract_frustum(mcull_globals *g);
/* updateTransforms (Engine.cpp:2353-2377): draw_count model matrices from glibc's rand() sequence after
 * srand(seed) (the reference uses 42), generated on a private generator state: the process-wide
 * rand() is left untouched. */
int mcull_reference_transforms(uint32_t draw_count, uint32_t seed, mcull_transform *out);
/* createIndirectBuffer (Engine.cpp:1466-1484): the static commands the reference uploads once,
 * taskCount = ceil(meshlet_count / 32), firstTask = 0.  mcull_cull writes the dynamic ones. */
int mcull_create_indirect_buffer(uint32_t draw_count, uint32_t meshlet_count, uint32_t index_count,
                                 mcull_draw_indirect *out);

/* ------------------------------------------------------------------ several GPUs from one host thread */

/* A group holds one context per shard of a scene (normally one shard per GPU; an ordinal may repeat, which
 * puts several shards on one GPU), each on a stream of its own, all driven by the calling thread.  The
 * instances of the scene are split into contiguous ranges, shard r holding the r-th range: upload each
 * shard's meshlets / transforms / ranges / pyramid through the ordinary calls on mcull_group_ctx(grp, r).
 * The reference selects a single physical device (Engine.cpp:658-666) and has no counterpart; for one
 * process per GPU use mcull_rebase_commands with a collective of the caller's instead. */
#define MCULL_GROUP_MAX_SHARDS 64
typedef struct mcull_group mcull_group;

typedef struct mcull_group_result {
    mcull_cull_result local;     /* this shard's own cull: list (drawIDs count from the shard's first instance),
                                    rank-local commands, count                                              */
    const void *commands;        /* on this shard's device: the records of ALL instances of the scene in instance
                                    order; NV records carry firstTask into the concatenation of the shards' lists */
    const uint32_t *first_task;  /* EXT records only (else NULL): the same offsets, one per instance of the scene */
    const uint64_t *totals;      /* on this shard's device: n_shards visible counts, then their sum               */
    uint64_t n_instances;        /* instances of the whole scene                                                   */
    uint64_t first_instance;     /* global index of this shard's instance 0 (add it to the list's drawIDs)         */
} mcull_group_result;

int mcull_group_create(const int *device_ordinals, uint32_t n_shards, mcull_group **out);
int mcull_group_destroy(mcull_group *grp);
uint32_t mcull_group_size(const mcull_group *grp);
mcull_ctx *mcull_group_ctx(mcull_group *grp, uint32_t shard);
const char *mcull_group_last_error(const mcull_group *grp);
/* mcull_cull on every shard with the same Globals and params (ordered compaction; list / commands / total of
 * params must be NULL), then one exchange: each shard copies its commands and count straight to every other
 * device (point to point, one copy per pair), and every device packs what it received into the scene-wide
 * command buffer with firstTask_i = sum of the visible counts of all instances before i - the buffer one
 * mcull_cull over the unsharded scene would write.  Asynchronous: results[d] (n_shards entries, may be NULL)
 * is valid in stream order on shard d's stream and until the next mcull_group_cull. */
int mcull_group_cull(mcull_group *grp, const mcull_globals *globals, const mcull_cull_params *params,
                     mcull_group_result *results);
/* The culling work of one recorded frame (Engine::recordCmdBuffer, Engine.cpp:2381-2551) over the shards:
 * every shard culls with prepass_tests and rasterises its survivors (mcull_depth_rasterize; geometry must be
 * uploaded on every shard), the shards' depth images are merged on every device into the depth image of the whole
 * scene (element-wise nearest = maximum, read from the other devices in place), every device builds the pyramid
 * of slot `view` from it, and mcull_group_cull runs with `params` (NULL: all three tests) against that pyramid.
 * Results equal the unsharded two-phase frame. */
int mcull_group_frame(mcull_group *grp, const mcull_globals *globals, uint32_t width, uint32_t height,
                      uint32_t prepass_tests, uint32_t view, const mcull_cull_params *params, mcull_group_result *results);
int mcull_group_synchronize(mcull_group *grp);
/* Synchronous read-back of one device's copy of the scene-wide result; any pointer may be NULL.
 * totals receives n_shards + 1 values. */
int mcull_group_download(mcull_group *grp, uint32_t shard, void *commands, uint64_t command_bytes, uint32_t *first_task,
                         uint64_t *totals);

#ifdef __cplusplus
}
#endif
#endif /* MCULL_H */
