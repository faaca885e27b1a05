/*
 * mcull_oracle.h — CPU restatement (plain C) of the GPU-driven culling path of
 * sagizilla-dev/vulkan-renderer.  TEST INFRASTRUCTURE ONLY: nothing in the product
 * library links, loads or calls this; only tests/, the smoke check and the bench's
 * cpu_baseline leg do, and there only as the checker / the CPU number reported beside
 * the GPU one.
 *
 * PARITY PINNED ONLY IN PART: the reference has no tests or golden vectors for this path,
 * its GLSL cannot be compiled here (no glslc / Vulkan device) and Engine.cpp cannot be
 * built (glm, Vulkan, GLFW, stb and tinyobjloader headers are absent; the submodule
 * directories are empty).  The one output of the reference that exists is a pair of
 * screenshots of its window; tests/test_reference_images.py renders the same scene through
 * this restatement and compares.  That pins createGlobals, updateTransforms, the OBJ
 * loader, the reorderer, the meshlet builder (partition and numbering) and the projection
 * conventions, and shows that the frustum and cone predicates remove nothing the
 * reference draws.  PARITY UNPINNED for everything an image cannot show: the bit-level
 * values of the predicates and their margins, the occlusion test, the pyramid, the bounds
 * and cones.  Those are anchored on the reference's source text alone, cross-checked by an
 * independent numpy restatement (oracle/np_oracle.py) and by hand-derived known-answer
 * cases (tests/).
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout).  Where GLSL / glm leave precision implementation-defined the
 * canonical choice is stated at the function and mirrored bit-for-bit by the HIP kernels:
 *   - all arithmetic IEEE binary32, no fused multiply-add (-ffp-contract=off);
 *   - dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;  M*v sums columns in order x,y,z,w;
 *   - normalize(v) = v * (1.0f / sqrtf(dot(v,v)));  length(v) = sqrtf(dot(v,v));
 *   - inverse(view) and proj*view are uniform and evaluated once on the host;
 *   - ceil(log2(x)) is evaluated exactly from the float's exponent and mantissa;
 *   - nearest/REPEAT texel address = floorf(u * w_L) then integer floor-mod.
 */
#ifndef MCULL_ORACLE_H
#define MCULL_ORACLE_H

#include <stdint.h>
#include "../include/mcull.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- geometry pipeline */

typedef struct orc_mesh {
    mcull_vertex *vertices;   uint64_t vertex_count;
    uint32_t *indices;        uint64_t index_count;     /* 3 * triangles */
    mcull_meshlet *meshlets;  uint64_t meshlet_count;
    uint32_t *meshlet_data;   uint64_t meshlet_data_words;
} orc_mesh;

void orc_mesh_free(orc_mesh *m);

/* loadModel up to (not including) optimizeGeometry: Engine.cpp:243-282. Returns 0 or -1. */
int orc_load_obj(const char *path, orc_mesh *out);
/* Same from raw arrays (positions xyz, optional normals xyz, optional uvs, per-corner
 * index triples v/t/n with -1 for absent): the dedup+quantise loop Engine.cpp:252-281. */
int orc_build_vertices(const float *pos, uint64_t npos, const float *nrm, uint64_t nnrm,
                       const float *uv, uint64_t nuv, const int32_t *corners /* v,t,n */,
                       uint64_t ncorners, orc_mesh *out);
/* computeVertexScore: Engine.cpp:308-335. */
float orc_vertex_score(int cache_position, int valence);
/* optimizeGeometry (modified Forsyth): Engine.cpp:338-471.  In-place on m->indices. */
void orc_optimize_geometry(orc_mesh *m);
/* createMeshlets + buildMeshletCon: Engine.cpp:548-628, :472-547. */
void orc_create_meshlets(orc_mesh *m);
/* buildMeshletCon alone, from the packed meshletData block of one meshlet. */
void orc_build_meshlet_cone(const mcull_vertex *vertices, const uint32_t *meshlet_data,
                            mcull_meshlet *meshlet);
/* the same over an array of meshlets, in order, one thread (the loop of Engine.cpp:585,625) */
void orc_build_meshlet_cones(const mcull_vertex *vertices, const uint32_t *meshlet_data,
                             mcull_meshlet *meshlets, uint64_t count);
/* utils.hpp:94-99 (round-to-nearest-even float -> binary16). */
uint16_t orc_float_to_half(float f);

/* ---------------------------------------------------------------- scene inputs      */

/* createGlobals: Engine.cpp:2323-2352.  eye/centre/up as glm::lookAt; flip_y != 0
 * reproduces `proj[1][1] *= -1` (Engine.cpp:2335). */
void orc_create_globals(const float eye[3], const float centre[3], const float up[3],
                        float fov_deg, float z_near, uint32_t width, uint32_t height,
                        int flip_y, uint32_t meshlet_count, mcull_globals *out);
/* Frustum planes from arbitrary view/proj already stored in g (Engine.cpp:2337-2349). */
void orc_extract_frustum(mcull_globals *g);
/* updateTransforms: Engine.cpp:2353-2379 (srand(42), glibc rand()). */
void orc_update_transforms(uint32_t draw_count, mcull_transform *out);
/* createIndirectBuffer: Engine.cpp:1466-1484 (the static reference commands). */
void orc_create_indirect_buffer(uint32_t draw_count, uint32_t meshlet_count, uint32_t index_count,
                                mcull_draw_indirect *out);

/* ---------------------------------------------------------------- Hi-Z pyramid      */

uint32_t orc_hiz_level_count(uint32_t w, uint32_t h);          /* Engine.cpp:1651        */
/* level extents + float offsets into a flat buffer; returns total floats (Engine.cpp:1845-1846) */
uint64_t orc_hiz_layout(uint32_t w, uint32_t h, uint32_t *levels, uint32_t lw[16], uint32_t lh[16],
                        uint64_t off[16]);
/* buildHiZPyramid + shader.comp: Engine.cpp:1834-1866, shader.comp:12-31.  threads<=1 = scalar. */
void orc_hiz_build(const float *depth, uint32_t w, uint32_t h, float *pyramid, int threads);

/* ---------------------------------------------------------------- per-pair tests    */

typedef struct orc_view {
    mcull_globals g;
    float cam_pos[3];   /* inverse(view)[3].xyz, shader.task:88-89                        */
    float vp[16];       /* proj * view, shader.task:60                                    */
    const float *pyramid; uint32_t w0, h0, levels;
    uint32_t lw[16], lh[16]; uint64_t off[16];
} orc_view;

/* Derives cam_pos and vp from g (host-side hoist of shader.task:60,88). */
void orc_view_init(orc_view *v, const mcull_globals *g, const float *pyramid, uint32_t w0, uint32_t h0);

typedef struct orc_pair_result {
    int visible;            /* accept, shader.task:101                                     */
    int frustum_culled, occlusion_culled, cone_culled; /* each evaluated unconditionally    */
    float frustum_margin;   /* min_i(dot_i + w_i + r)                                      */
    float occlusion_margin; /* ndc.z - (d - 0.05)                                          */
    float cone_margin;      /* dot(axis, viewdir) + coneW                                  */
    uint32_t mip, tx, ty;   /* texel sampled                                               */
} orc_pair_result;

/* shader.task:85-101 for one (transform, meshlet) pair; tests = MCULL_TEST_* bits. */
void orc_cull_pair(const orc_view *v, const float model[16], const float cone[4],
                   const float centre[3], float radius, uint32_t tests, orc_pair_result *out);

/* exact ceil(log2(max(x,1))) clamped to [0, levels-1] (shader.task:71-72) */
uint32_t orc_mip_level(float screen_pixels, uint32_t levels);

/* ---------------------------------------------------------------- whole-scene cull  */

typedef struct orc_scene {
    const mcull_meshlet *meshlets; uint64_t meshlet_count;      /* AoS, or NULL with SoA below */
    const float *cone_axis_w; const float *centre_radius;       /* SoA float4 streams          */
    const mcull_transform *transforms; uint64_t instance_count;
    const uint32_t *range_offset; const uint32_t *range_count;  /* NULL -> [0, g.meshletCount) */
} orc_scene;

typedef struct orc_cull_output {
    uint64_t n_groups;      /* 64-meshlet groups over all instances                        */
    uint64_t *mask;         /* n_groups words, bit l of group = visibility (may be NULL)   */
    mcull_visible *list;    /* sorted by (drawID, meshletID); capacity list_capacity       */
    uint64_t list_capacity;
    uint64_t total;         /* visible pairs                                               */
    mcull_mesh_tasks_cmd *commands; /* per instance {visible_i, exclusive scan} (may be NULL) */
    uint64_t n_pairs;
    uint64_t culled_frustum, culled_occlusion, culled_cone; /* per-test counts (unconditional) */
    float *margins;         /* optional, 3 floats per pair in pair order (may be NULL)     */
} orc_cull_output;

/* shader.task:79-119 over every (instance, meshlet) pair + global compaction.
 * threads <= 1 runs the scalar loop; otherwise OpenMP over instances.  Returns 0, or -1 if
 * the list overflowed (total still counts everything). */
int orc_cull_scene(const orc_scene *s, const orc_view *v, uint32_t tests, int threads,
                   orc_cull_output *out);

/* literal shader.task:103-118 payloads per (instance, 32-group); out has
 * instance_count * ceil(meshlets/32) records (uniform meshlet count only). */
uint64_t orc_task_payloads(const orc_scene *s, const orc_view *v, uint32_t tests,
                           mcull_task_payload *out);

/* ---------------------------------------------------------------- depth source      */

/* Software stand-in for the depth pre-pass (Engine.cpp:2397-2441, shader.mesh:60-101): rasterises
 * the triangles of the listed (drawID, meshletID) pairs into `depth` (h rows of w floats, reverse-Z,
 * the caller clears it to 0), keeping the nearest (largest) depth per pixel. */
void orc_rasterize_depth(const mcull_vertex *vertices, const uint32_t *meshlet_data, const mcull_meshlet *meshlets,
                         const mcull_transform *transforms, const mcull_visible *list, uint64_t count,
                         const mcull_globals *g, uint32_t w, uint32_t h, float *depth);

/* ---------------------------------------------------------------- utilities         */
uint64_t orc_fnv1a64(const void *data, uint64_t bytes);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
