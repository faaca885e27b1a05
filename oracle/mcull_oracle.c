/*
 * mcull_oracle.c — CPU restatement of the reference's culling path.  TEST INFRASTRUCTURE
 * ONLY (see mcull_oracle.h).  PARITY PINNED ONLY IN PART: the reference holds no tests or
 * golden vectors for this path and cannot be built or run here; its two screenshots pin the
 * scene inputs, the load-time pipeline and the conservative side of the frustum and cone
 * tests, everything else is PARITY UNPINNED (see the header).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (see oracle/Makefile).  The
 * -ffp-contract=off flag is part of the definition: no multiply-add may be fused.
 */
#define _GNU_SOURCE
#include "mcull_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ====================================================================== small math */

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
/* glm dot for vec3: tmp = a*b; tmp.x + tmp.y + tmp.z */
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* glm::cross: (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y) */
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
/* glm::normalize: x * inversesqrt(dot(x,x)), inversesqrt(x) = 1 / sqrt(x) */
static inline v3 v3_normalize(v3 a) { return v3_scale(a, 1.0f / sqrtf(v3_dot(a, a))); }
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }

/* column-major 4x4: element (row r, col c) = m[4*c + r] */
#define M(m, r, c) ((m)[4 * (c) + (r)])

/* glm mat4 * mat4: Result[c] = ((A[0]*B[c][0] + A[1]*B[c][1]) + A[2]*B[c][2]) + A[3]*B[c][3] */
static void m4_mul(const float *a, const float *b, float *out) {
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++)
            r[4 * c + row] = ((M(a, row, 0) * M(b, 0, c) + M(a, row, 1) * M(b, 1, c)) +
                              M(a, row, 2) * M(b, 2, c)) + M(a, row, 3) * M(b, 3, c);
    memcpy(out, r, sizeof r);
}
static void m4_identity(float *m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.0f; }

/* (M * vec4(p,1)).xyz with column sum order x,y,z,w; the w column is multiplied by exactly 1 */
static inline v3 m4_transform_point(const float *m, v3 p) {
    v3 r;
    r.x = ((M(m, 0, 0) * p.x + M(m, 0, 1) * p.y) + M(m, 0, 2) * p.z) + M(m, 0, 3);
    r.y = ((M(m, 1, 0) * p.x + M(m, 1, 1) * p.y) + M(m, 1, 2) * p.z) + M(m, 1, 3);
    r.z = ((M(m, 2, 0) * p.x + M(m, 2, 1) * p.y) + M(m, 2, 2) * p.z) + M(m, 2, 3);
    return r;
}
/* mat3(M) * v */
static inline v3 m4_transform_dir(const float *m, v3 p) {
    v3 r;
    r.x = (M(m, 0, 0) * p.x + M(m, 0, 1) * p.y) + M(m, 0, 2) * p.z;
    r.y = (M(m, 1, 0) * p.x + M(m, 1, 1) * p.y) + M(m, 1, 2) * p.z;
    r.z = (M(m, 2, 0) * p.x + M(m, 2, 1) * p.y) + M(m, 2, 2) * p.z;
    return r;
}

/* General 4x4 inverse by cofactors (the method glm::inverse / GLSL inverse() use), f32.
 * Only column 3 (the camera position for a view matrix) is consumed by the path. */
static void m4_inverse(const float *m, float *out) {
    /* Plain cofactor expansion, written per element so the operation order is unambiguous:
     * inv(r,c) = cof(c,r) * (1/det), cof(i,j) = (-1)^(i+j) * minor(i,j); each 3x3 minor is
     * expanded along its first row as  e0*(e4*e8 - e5*e7) - e1*(e3*e8 - e5*e6)
     * + e2*(e3*e7 - e4*e6), summed left to right; det expands along row 0. */
    float cof[16];
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) {
            float e[9]; int k = 0;
            for (int r = 0; r < 4; r++) {
                if (r == i) continue;
                for (int c = 0; c < 4; c++) {
                    if (c == j) continue;
                    e[k++] = M(m, r, c);
                }
            }
            float minor = (e[0] * (e[4] * e[8] - e[5] * e[7]) - e[1] * (e[3] * e[8] - e[5] * e[6])) +
                          e[2] * (e[3] * e[7] - e[4] * e[6]);
            cof[4 * j + i] = ((i + j) & 1) ? -minor : minor; /* stored as (row i, col j) */
        }
    }
    float det = ((M(m,0,0) * M(cof,0,0) + M(m,0,1) * M(cof,0,1)) + M(m,0,2) * M(cof,0,2)) +
                M(m,0,3) * M(cof,0,3);
    float inv_det = 1.0f / det;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            M(out, r, c) = M(cof, c, r) * inv_det;
}

/* ---------------------------------------------------------------- depth pre-pass stand-in
 * The reference fills its depth buffer with the hardware rasteriser: depth pre-pass of
 * Engine.cpp:2397-2441 running shader.mesh:60-101 (gl_Position = proj * view * model * vec4(v,1),
 * evaluated left to right, i.e. ((proj*view)*model)*v), viewport (0,0,W,H) with depth 0..1
 * (Engine.cpp:2413-2420), back faces culled with counter-clockwise front faces (Engine.cpp:961-962),
 * depth test GREATER_OR_EQUAL on a buffer cleared to 0 (Engine.cpp:975, :2404).  Sub-pixel
 * snapping and the exact interpolation of a hardware rasteriser are implementation-defined; the
 * canonical software form used here: float edge functions at pixel centres, top-left fill rule,
 * screen-space barycentric depth, fragments outside 0 <= z <= 1 dropped.  Triangles are clipped in
 * clip space against the near plane z <= w (Sutherland-Hodgman, the intersection at
 * t = d_p / (d_p - d_q) with d = w - z, each coordinate p + t*(q - p)); what remains has w > 0 for a
 * projection whose near plane lies in front of the eye, anything else is dropped.  Every expression
 * fixes its association order. */
typedef struct { float x, y, z; int ok; } screen_vert;
typedef struct { float x, y, z, w; } clip_vert;

static clip_vert clip_vertex(const float *mvp, const float *pos) {
    clip_vert r;
    r.x = ((M(mvp, 0, 0) * pos[0] + M(mvp, 0, 1) * pos[1]) + M(mvp, 0, 2) * pos[2]) + M(mvp, 0, 3);
    r.y = ((M(mvp, 1, 0) * pos[0] + M(mvp, 1, 1) * pos[1]) + M(mvp, 1, 2) * pos[2]) + M(mvp, 1, 3);
    r.z = ((M(mvp, 2, 0) * pos[0] + M(mvp, 2, 1) * pos[1]) + M(mvp, 2, 2) * pos[2]) + M(mvp, 2, 3);
    r.w = ((M(mvp, 3, 0) * pos[0] + M(mvp, 3, 1) * pos[1]) + M(mvp, 3, 2) * pos[2]) + M(mvp, 3, 3);
    return r;
}

static screen_vert to_screen(clip_vert c, float wf, float hf) {
    screen_vert r;
    r.ok = c.w > 0.0f;
    float nx = c.x / c.w, ny = c.y / c.w;
    r.z = c.z / c.w;
    r.x = (nx * 0.5f + 0.5f) * wf;   /* viewport transform, Engine.cpp:2413-2420 */
    r.y = (ny * 0.5f + 0.5f) * hf;
    return r;
}

static clip_vert clip_intersect(clip_vert p, clip_vert q) {
    float dp = p.w - p.z, dq = q.w - q.z;
    float t = dp / (dp - dq);
    clip_vert r;
    r.x = p.x + t * (q.x - p.x);
    r.y = p.y + t * (q.y - p.y);
    r.z = p.z + t * (q.z - p.z);
    r.w = p.w + t * (q.w - p.w);
    return r;
}

/* edge a->b of a triangle that is counter-clockwise on a y-down screen: interior is e < 0 */
static inline float edge_fn(float ax, float ay, float bx, float by, float px, float py) {
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
static inline int edge_top_left(float ax, float ay, float bx, float by) {
    float dx = bx - ax, dy = by - ay;
    return dy > 0.0f || (dy == 0.0f && dx < 0.0f);
}

static void raster_triangle(screen_vert a, screen_vert b, screen_vert c, uint32_t w, uint32_t h, float *depth) {
    if (!(a.ok && b.ok && c.ok)) return;
    float area = (b.x - a.x) * (c.y - a.y) - (c.x - a.x) * (b.y - a.y);
    if (!(area < 0.0f)) return;  /* back-facing, degenerate or NaN */
    float minx = fminf(fminf(a.x, b.x), c.x), maxx = fmaxf(fmaxf(a.x, b.x), c.x);
    float miny = fminf(fminf(a.y, b.y), c.y), maxy = fmaxf(fmaxf(a.y, b.y), c.y);
    float x_lo = fmaxf(ceilf(minx - 0.5f), 0.0f), x_hi = fminf(floorf(maxx - 0.5f), (float)(w - 1));
    float y_lo = fmaxf(ceilf(miny - 0.5f), 0.0f), y_hi = fminf(floorf(maxy - 0.5f), (float)(h - 1));
    if (!(x_lo <= x_hi && y_lo <= y_hi)) return;
    int tl0 = edge_top_left(b.x, b.y, c.x, c.y), tl1 = edge_top_left(c.x, c.y, a.x, a.y), tl2 = edge_top_left(a.x, a.y, b.x, b.y);
    for (uint32_t py = (uint32_t)y_lo; py <= (uint32_t)y_hi; py++)
        for (uint32_t px = (uint32_t)x_lo; px <= (uint32_t)x_hi; px++) {
            float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
            float w0 = edge_fn(b.x, b.y, c.x, c.y, cx, cy);
            float w1 = edge_fn(c.x, c.y, a.x, a.y, cx, cy);
            float w2 = edge_fn(a.x, a.y, b.x, b.y, cx, cy);
            if (!((w0 < 0.0f || (w0 == 0.0f && tl0)) && (w1 < 0.0f || (w1 == 0.0f && tl1)) && (w2 < 0.0f || (w2 == 0.0f && tl2))))
                continue;
            float z = ((w0 * a.z + w1 * b.z) + w2 * c.z) / area;
            if (!(z >= 0.0f && z <= 1.0f)) continue;
            uint32_t bits; memcpy(&bits, &z, 4);
            bits &= 0x7fffffffu;  /* -0 -> +0 */
            uint32_t cur; memcpy(&cur, &depth[(size_t)py * w + px], 4);
            if (bits > cur) memcpy(&depth[(size_t)py * w + px], &bits, 4);  /* GREATER_OR_EQUAL on non-negative floats */
        }
}

void orc_rasterize_depth(const mcull_vertex *vertices, const uint32_t *meshlet_data, const mcull_meshlet *meshlets,
                         const mcull_transform *transforms, const mcull_visible *list, uint64_t count,
                         const mcull_globals *g, uint32_t w, uint32_t h, float *depth) {
    float vp[16];
    m4_mul(g->proj, g->view, vp);
    for (uint64_t e = 0; e < count; e++) {
        const mcull_meshlet *m = &meshlets[list[e].meshletID];
        float mvp[16];
        m4_mul(vp, transforms[list[e].drawID].model, mvp);
        const uint32_t *globals = meshlet_data + m->dataOffset;
        const uint8_t *locals = (const uint8_t *)(globals + m->vertexCount);
        clip_vert cv[64];
        for (uint32_t i = 0; i < m->vertexCount; i++) cv[i] = clip_vertex(mvp, &vertices[globals[i]].vx);
        for (uint32_t t = 0; t < m->triangleCount; t++) {
            clip_vert tri[3] = {cv[locals[3 * t]], cv[locals[3 * t + 1]], cv[locals[3 * t + 2]]};
            int in[3], n_in = 0;
            for (int k = 0; k < 3; k++) { in[k] = (tri[k].w - tri[k].z) >= 0.0f; n_in += in[k]; }
            if (n_in == 0) continue;
            clip_vert poly[4]; int np = 0;
            if (n_in == 3) {
                poly[0] = tri[0]; poly[1] = tri[1]; poly[2] = tri[2]; np = 3;
            } else {
                for (int k = 0; k < 3; k++) {           /* edges 0->1, 1->2, 2->0 in winding order */
                    int q = (k + 1) % 3;
                    if (in[k]) poly[np++] = tri[k];
                    if (in[k] != in[q]) poly[np++] = in[k] ? clip_intersect(tri[k], tri[q]) : clip_intersect(tri[q], tri[k]);
                }
            }
            screen_vert sv[4];
            for (int k = 0; k < np; k++) sv[k] = to_screen(poly[k], (float)w, (float)h);
            raster_triangle(sv[0], sv[1], sv[2], w, h, depth);
            if (np == 4) raster_triangle(sv[0], sv[2], sv[3], w, h, depth);
        }
    }
}

uint64_t orc_fnv1a64(const void *data, uint64_t bytes) {
    const uint8_t *p = (const uint8_t *)data;
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < bytes; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ====================================================================== geometry */

void orc_mesh_free(orc_mesh *m) {
    if (!m) return;
    free(m->vertices); free(m->indices); free(m->meshlets); free(m->meshlet_data);
    memset(m, 0, sizeof *m);
}

/* utils.hpp:94-99: (_Float16)f is an IEEE round-to-nearest-even conversion. */
uint16_t orc_float_to_half(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? (0x0200u | ((absx >> 13) & 0x3ffu)) : 0));
    if (absx >= 0x477ff000u) /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    if (absx < 0x33000001u) /* <= 2^-25 rounds to zero */
        return (uint16_t)sign;
    int32_t e = (int32_t)(absx >> 23) - 127;
    uint32_t mant = (absx & 0x7fffffu) | 0x800000u;
    uint32_t shift, half;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        half = 0;
    } else {
        shift = 13;
        half = (uint32_t)(e + 15) << 10;
        mant &= 0x7fffffu;
    }
    uint32_t q = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    return (uint16_t)(sign | (half + q)); /* carry into the exponent is the correct rounding */
}

/* Engine.cpp:265-268: normalize, then uint8_t((n*0.5f+0.5f)*255.0f), truncating.  A NaN
 * component (normalize of the zero vector when the OBJ has no normals) converts to 0, which
 * is what the x86 cvttss2si + byte truncation of the reference build produces. */
static uint8_t quantise_normal(float n) {
    float s = (n * 0.5f + 0.5f) * 255.0f;
    if (!(s == s)) return 0;
    if (s <= -1.0f || s >= 2147483648.0f) return 0;
    return (uint8_t)(int32_t)s;
}


static uint64_t vertex_hash(const mcull_vertex *v) {
    uint32_t w[5];
    float p[3] = {v->vx, v->vy, v->vz};
    for (int i = 0; i < 3; i++) {
        float f = p[i] == 0.0f ? 0.0f : p[i]; /* -0 == +0 under operator== (Engine.hpp:57-61) */
        memcpy(&w[i], &f, 4);
    }
    w[3] = (uint32_t)v->nx | ((uint32_t)v->ny << 8) | ((uint32_t)v->nz << 16);
    w[4] = (uint32_t)v->tu | ((uint32_t)v->tv << 16);
    return orc_fnv1a64(w, sizeof w);
}
static int vertex_equal(const mcull_vertex *a, const mcull_vertex *b) { /* Engine.hpp:57-61 */
    return a->vx == b->vx && a->vy == b->vy && a->vz == b->vz && a->nx == b->nx && a->ny == b->ny &&
           a->nz == b->nz && a->tu == b->tu && a->tv == b->tv;
}

/* Engine.cpp:252-281: one Vertex per face corner, dedup in first-seen order. */
int orc_build_vertices(const float *pos, uint64_t npos, const float *nrm, uint64_t nnrm,
                       const float *uv, uint64_t nuv, const int32_t *corners, uint64_t ncorners,
                       orc_mesh *out) {
    memset(out, 0, sizeof *out);
    uint64_t cap = 16;
    while (cap < ncorners * 2 + 16) cap <<= 1;
    uint32_t *slots = (uint32_t *)malloc(cap * sizeof(uint32_t));
    mcull_vertex *verts = (mcull_vertex *)calloc(ncorners ? ncorners : 1, sizeof(mcull_vertex));
    uint32_t *idx = (uint32_t *)malloc((ncorners ? ncorners : 1) * sizeof(uint32_t));
    if (!slots || !verts || !idx) { free(slots); free(verts); free(idx); return -1; }
    memset(slots, 0xff, cap * sizeof(uint32_t));
    uint64_t nv = 0;
    for (uint64_t c = 0; c < ncorners; c++) {
        int32_t vi = corners[3 * c + 0], ti = corners[3 * c + 1], ni = corners[3 * c + 2];
        if (vi < 0 || (uint64_t)vi >= npos) { free(slots); free(verts); free(idx); return -1; }
        mcull_vertex v; memset(&v, 0, sizeof v);             /* Vertex vertex{}; */
        v.vx = pos[3 * vi + 0]; v.vy = pos[3 * vi + 1]; v.vz = pos[3 * vi + 2];
        v3 n = v3_make(0.0f, 0.0f, 0.0f);
        if (nnrm != 0) {                                      /* !attrib.normals.empty() */
            if (ni < 0 || (uint64_t)ni >= nnrm) { free(slots); free(verts); free(idx); return -1; }
            n = v3_make(nrm[3 * ni + 0], nrm[3 * ni + 1], nrm[3 * ni + 2]);
        }
        n = v3_normalize(n);
        v.nx = quantise_normal(n.x); v.ny = quantise_normal(n.y); v.nz = quantise_normal(n.z);
        if (nuv != 0) {                                       /* !attrib.texcoords.empty() */
            if (ti < 0 || (uint64_t)ti >= nuv) { free(slots); free(verts); free(idx); return -1; }
            v.tu = orc_float_to_half(uv[2 * ti + 0]);
            v.tv = orc_float_to_half(1.0f - uv[2 * ti + 1]);
        }
        uint64_t h = vertex_hash(&v) & (cap - 1);
        uint32_t found = 0xffffffffu;
        int has_nan = !(v.vx == v.vx && v.vy == v.vy && v.vz == v.vz);
        while (!has_nan && slots[h] != 0xffffffffu) {
            if (vertex_equal(&verts[slots[h]], &v)) { found = slots[h]; break; }
            h = (h + 1) & (cap - 1);
        }
        if (found == 0xffffffffu) {
            found = (uint32_t)nv;
            verts[nv++] = v;
            if (!has_nan) slots[h] = found;
        }
        idx[c] = found;
    }
    free(slots);
    out->vertices = verts; out->vertex_count = nv;
    out->indices = idx; out->index_count = ncorners;
    return 0;
}

/* Minimal OBJ reader standing in for tinyobj::LoadObj (Engine.cpp:250): v / vn / vt / f
 * records, 1-based and negative indices, fan triangulation of polygons, faces taken in
 * file order.  Numbers are parsed with strtod and narrowed to float. */
int orc_load_obj(const char *path, orc_mesh *out) {
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    uint64_t np = 0, nn = 0, nt = 0, nc = 0, cp = 1024, cn = 1024, ct = 1024, cc = 4096;
    float *pos = (float *)malloc(cp * 12), *nrm = (float *)malloc(cn * 12), *uv = (float *)malloc(ct * 8);
    int32_t *corners = (int32_t *)malloc(cc * 12);
    char *line = NULL; size_t linecap = 0;
    int rc = 0;
    while (getline(&line, &linecap, f) > 0) {
        char *p = line;
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            if (np == cp) { cp *= 2; pos = (float *)realloc(pos, cp * 12); }
            char *e = p + 1;
            for (int k = 0; k < 3; k++) pos[3 * np + k] = (float)strtod(e, &e);
            np++;
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            if (nn == cn) { cn *= 2; nrm = (float *)realloc(nrm, cn * 12); }
            char *e = p + 2;
            for (int k = 0; k < 3; k++) nrm[3 * nn + k] = (float)strtod(e, &e);
            nn++;
        } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
            if (nt == ct) { ct *= 2; uv = (float *)realloc(uv, ct * 8); }
            char *e = p + 2;
            for (int k = 0; k < 2; k++) uv[2 * nt + k] = (float)strtod(e, &e);
            nt++;
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            int32_t face[64][3]; int nf = 0;
            char *e = p + 1;
            for (;;) {
                while (*e == ' ' || *e == '\t') e++;
                if (*e == '\0' || *e == '\n' || *e == '\r' || nf == 64) break;
                long a = 0, b = 0, c = 0; int hb = 0, hc = 0;
                a = strtol(e, &e, 10);
                if (*e == '/') {
                    e++;
                    if (*e != '/') { b = strtol(e, &e, 10); hb = 1; }
                    if (*e == '/') { e++; c = strtol(e, &e, 10); hc = 1; }
                }
                face[nf][0] = (int32_t)(a > 0 ? a - 1 : (long)np + a);
                face[nf][1] = hb ? (int32_t)(b > 0 ? b - 1 : (long)nt + b) : -1;
                face[nf][2] = hc ? (int32_t)(c > 0 ? c - 1 : (long)nn + c) : -1;
                nf++;
            }
            for (int k = 2; k < nf; k++) {
                if (nc + 3 > cc) { cc *= 2; corners = (int32_t *)realloc(corners, cc * 12); }
                memcpy(&corners[3 * nc], face[0], 12); nc++;
                memcpy(&corners[3 * nc], face[k - 1], 12); nc++;
                memcpy(&corners[3 * nc], face[k], 12); nc++;
            }
        }
    }
    free(line); fclose(f);
    rc = orc_build_vertices(pos, np, nrm, nn, uv, nt, corners, nc, out);
    free(pos); free(nrm); free(uv); free(corners);
    return rc;
}

/* computeVertexScore: Engine.cpp:308-335.  std::pow(float,float) is powf. */
float orc_vertex_score(int cache_position, int valence) {
    if (valence == 0) return -1.0f;
    float cache_score = 0.0f;
    if (cache_position < 0) {
        cache_score = 0.0f;
    } else if (cache_position < 3) {
        cache_score = 0.75f;
    } else {
        const int CACHE_SIZE = 32;
        float normalized = (cache_position - 3) * (1.0f / (CACHE_SIZE - 3));
        cache_score = powf(1.0f - normalized, 1.5f);
    }
    float valence_score = 2.0f * powf((float)valence, -0.5f);
    return cache_score + valence_score;
}

/* optimizeGeometry: Engine.cpp:338-471, statement for statement. */
void orc_optimize_geometry(orc_mesh *m) {
    const uint32_t tri_count = (uint32_t)(m->index_count / 3);
    const uint32_t vert_count = (uint32_t)m->vertex_count;
    uint32_t *ind = m->indices;
    if (tri_count == 0) return;
    /* vertexToTriangles as CSR, lists in push_back order (:345-355) */
    uint32_t *adj_start = (uint32_t *)calloc((size_t)vert_count + 1, 4);
    for (uint64_t i = 0; i < (uint64_t)tri_count * 3; i++) adj_start[ind[i] + 1]++;
    for (uint32_t v = 0; v < vert_count; v++) adj_start[v + 1] += adj_start[v];
    uint32_t *adj = (uint32_t *)malloc((size_t)tri_count * 3 * 4);
    uint32_t *fill = (uint32_t *)malloc(((size_t)vert_count + 1) * 4);
    memcpy(fill, adj_start, ((size_t)vert_count + 1) * 4);
    for (uint32_t t = 0; t < tri_count; t++)
        for (int k = 0; k < 3; k++) adj[fill[ind[3 * t + k]]++] = t;
    free(fill);

    int *unprocessed = (int *)malloc((size_t)vert_count * sizeof(int));
    int *cache_pos = (int *)malloc((size_t)vert_count * sizeof(int));
    float *vscore = (float *)malloc((size_t)vert_count * sizeof(float));
    float *tscore = (float *)malloc((size_t)tri_count * sizeof(float));
    uint8_t *finished = (uint8_t *)calloc(tri_count, 1);
    uint32_t *reordered = (uint32_t *)malloc((size_t)tri_count * 3 * 4);
    for (uint32_t v = 0; v < vert_count; v++) {
        unprocessed[v] = (int)(adj_start[v + 1] - adj_start[v]);
        cache_pos[v] = -1;
    }
    for (uint32_t v = 0; v < vert_count; v++) vscore[v] = orc_vertex_score(cache_pos[v], unprocessed[v]);
    for (uint32_t t = 0; t < tri_count; t++)
        tscore[t] = vscore[ind[3 * t]] + vscore[ind[3 * t + 1]] + vscore[ind[3 * t + 2]];

    int cache[40]; int cache_size = 0;
    uint64_t out_n = 0;
    int next_best = -1;
    for (uint32_t i = 0; i < tri_count; i++) {
        int best = next_best;
        float best_score;
        if (best < 0 || finished[best]) {
            best = -1;
            best_score = FLT_MIN; /* std::numeric_limits<float>::min() */
            for (uint32_t j = 0; j < tri_count; j++) {
                if (tscore[j] > best_score && !finished[j]) { best_score = tscore[j]; best = (int)j; }
            }
        }
        uint32_t tri_v[3] = {ind[best * 3 + 0], ind[best * 3 + 1], ind[best * 3 + 2]};
        reordered[out_n++] = tri_v[0]; reordered[out_n++] = tri_v[1]; reordered[out_n++] = tri_v[2];
        finished[best] = 1;

        uint32_t affected[8]; int n_affected = 0;
        for (int k = 0; k < 3; k++) {
            uint32_t v = tri_v[k];
            int at = -1;
            for (int c = 0; c < cache_size; c++) if (cache[c] == (int)v) { at = c; break; }
            if (at >= 0) { memmove(&cache[at], &cache[at + 1], (size_t)(cache_size - at - 1) * sizeof(int)); cache_size--; }
            memmove(&cache[1], &cache[0], (size_t)cache_size * sizeof(int));
            cache[0] = (int)v; cache_size++;
            if (cache_size > 32) {
                int evicted = cache[cache_size - 1];
                cache_size--;
                cache_pos[evicted] = -1;
                affected[n_affected++] = (uint32_t)evicted;
            }
            affected[n_affected++] = v;
            unprocessed[v]--;
        }
        for (int c = 0; c < cache_size; c++) cache_pos[cache[c]] = c;
        for (int a = 0; a < n_affected; a++) {
            uint32_t v = affected[a];
            vscore[v] = orc_vertex_score(cache_pos[v], unprocessed[v]);
            for (uint32_t e = adj_start[v]; e < adj_start[v + 1]; e++) {
                uint32_t t = adj[e];
                if (!finished[t]) tscore[t] = vscore[ind[3 * t]] + vscore[ind[3 * t + 1]] + vscore[ind[3 * t + 2]];
            }
        }
        next_best = -1;
        best_score = FLT_MIN;
        for (int c = 0; c < cache_size; c++) {
            uint32_t v = (uint32_t)cache[c];
            for (uint32_t e = adj_start[v]; e < adj_start[v + 1]; e++) {
                uint32_t t = adj[e];
                if (!finished[t] && tscore[t] > best_score) { best_score = tscore[t]; next_best = (int)t; }
            }
        }
    }
    memcpy(ind, reordered, (size_t)tri_count * 3 * 4);
    free(adj_start); free(adj); free(unprocessed); free(cache_pos); free(vscore); free(tscore);
    free(finished); free(reordered);
}

/* buildMeshletCon: Engine.cpp:472-547.  Reads the meshlet's packed block; writes cone,
 * coneApex (accumulated from the zero the caller's `meshlet = {}` left) and radius. */
void orc_build_meshlet_cone(const mcull_vertex *vertices, const uint32_t *meshlet_data,
                            mcull_meshlet *meshlet) {
    const uint32_t *globals = meshlet_data + meshlet->dataOffset;
    const uint8_t *locals = (const uint8_t *)(globals + meshlet->vertexCount);
    float normals[124][3];
    for (uint32_t i = 0; i < meshlet->triangleCount; i++) {
        const mcull_vertex *v0 = &vertices[globals[locals[i * 3 + 0]]];
        const mcull_vertex *v1 = &vertices[globals[locals[i * 3 + 1]]];
        const mcull_vertex *v2 = &vertices[globals[locals[i * 3 + 2]]];
        v3 p0 = v3_make(v0->vx, v0->vy, v0->vz), p1 = v3_make(v1->vx, v1->vy, v1->vz),
           p2 = v3_make(v2->vx, v2->vy, v2->vz);
        v3 n = v3_normalize(v3_cross(v3_sub(p1, p0), v3_sub(p2, p0)));
        normals[i][0] = n.x; normals[i][1] = n.y; normals[i][2] = n.z;
    }
    v3 avg = v3_make(0.0f, 0.0f, 0.0f);
    for (uint32_t i = 0; i < meshlet->triangleCount; i++) {
        avg.x += normals[i][0]; avg.y += normals[i][1]; avg.z += normals[i][2];
    }
    avg = v3_normalize(avg);
    float half_angle = 1.0f;
    for (uint32_t i = 0; i < meshlet->triangleCount; i++) {
        /* written as a*b + c*d + e*f: left to right */
        float dp = (normals[i][0] * avg.x + normals[i][1] * avg.y) + normals[i][2] * avg.z;
        half_angle = (dp < half_angle) ? dp : half_angle; /* std::min(halfAngle, dp) */
    }
    float cone_w = half_angle < 0.0f ? 1.0f : sqrtf(1 - half_angle * half_angle);
    meshlet->cone[0] = avg.x; meshlet->cone[1] = avg.y; meshlet->cone[2] = avg.z; meshlet->cone[3] = cone_w;
    meshlet->coneApex[0] = meshlet->coneApex[1] = meshlet->coneApex[2] = meshlet->coneApex[3] = 0.0f;
    for (uint32_t i = 0; i < meshlet->vertexCount; i++) {
        meshlet->coneApex[0] += vertices[globals[i]].vx / (float)meshlet->vertexCount;
        meshlet->coneApex[1] += vertices[globals[i]].vy / (float)meshlet->vertexCount;
        meshlet->coneApex[2] += vertices[globals[i]].vz / (float)meshlet->vertexCount;
    }
    float max_d2 = 0.0f;
    for (uint32_t i = 0; i < meshlet->vertexCount; i++) {
        float dx = vertices[globals[i]].vx - meshlet->coneApex[0];
        float dy = vertices[globals[i]].vy - meshlet->coneApex[1];
        float dz = vertices[globals[i]].vz - meshlet->coneApex[2];
        float d2 = (dx * dx + dy * dy) + dz * dz;
        max_d2 = (max_d2 < d2) ? d2 : max_d2; /* std::max(maxDistanceSqr, d2) */
    }
    meshlet->radius = sqrtf(max_d2);
}

/* createMeshlets: Engine.cpp:548-628. */
void orc_build_meshlet_cones(const mcull_vertex *vertices, const uint32_t *meshlet_data,
                             mcull_meshlet *meshlets, uint64_t count) {
    for (uint64_t i = 0; i < count; i++) orc_build_meshlet_cone(vertices, meshlet_data, &meshlets[i]);
}

void orc_create_meshlets(orc_mesh *m) {
    free(m->meshlets); free(m->meshlet_data);
    uint64_t tri_count = m->index_count / 3;
    uint64_t mcap = tri_count / 16 + 16, dcap = m->index_count + 1024, nm = 0, nd = 0;
    mcull_meshlet *meshlets = (mcull_meshlet *)malloc(mcap * sizeof(mcull_meshlet));
    uint32_t *data = (uint32_t *)malloc(dcap * 4);
    uint8_t *table = (uint8_t *)malloc(m->vertex_count ? m->vertex_count : 1);
    memset(table, 0xff, m->vertex_count ? m->vertex_count : 1);
    uint32_t cur_verts[64 + 3]; uint32_t n_verts = 0;
    uint8_t cur_idx[124 * 3 + 8]; uint32_t n_idx = 0;

#define FLUSH_MESHLET()                                                                         \
    do {                                                                                        \
        mcull_meshlet ml; memset(&ml, 0, sizeof ml);                                            \
        ml.dataOffset = (uint32_t)nd;                                                           \
        ml.triangleCount = (uint8_t)(n_idx / 3);                                                \
        ml.vertexCount = (uint8_t)n_verts;                                                      \
        while (n_idx % 4 != 0) cur_idx[n_idx++] = 0;                                            \
        while (nd + n_verts + n_idx / 4 > dcap) { dcap *= 2; data = (uint32_t *)realloc(data, dcap * 4); } \
        memcpy(&data[nd], cur_verts, (size_t)n_verts * 4); nd += n_verts;                       \
        memcpy(&data[nd], cur_idx, n_idx); nd += n_idx / 4;                                     \
        orc_build_meshlet_cone(m->vertices, data, &ml);                                         \
        if (nm == mcap) { mcap *= 2; meshlets = (mcull_meshlet *)realloc(meshlets, mcap * sizeof(mcull_meshlet)); } \
        meshlets[nm++] = ml;                                                                    \
    } while (0)

    for (uint64_t i = 0; i < m->index_count; i += 3) {
        uint32_t g0 = m->indices[i], g1 = m->indices[i + 1], g2 = m->indices[i + 2];
        int new_count = (table[g0] == 0xff) + (table[g1] == 0xff) + (table[g2] == 0xff);
        if (n_verts + (uint32_t)new_count > 64 || n_idx / 3 >= 124) {
            FLUSH_MESHLET();
            for (uint32_t j = 0; j < n_verts; j++) table[cur_verts[j]] = 0xff;
            n_verts = 0; n_idx = 0;
        }
        if (table[g0] == 0xff) { table[g0] = (uint8_t)n_verts; cur_verts[n_verts++] = g0; }
        if (table[g1] == 0xff) { table[g1] = (uint8_t)n_verts; cur_verts[n_verts++] = g1; }
        if (table[g2] == 0xff) { table[g2] = (uint8_t)n_verts; cur_verts[n_verts++] = g2; }
        cur_idx[n_idx++] = table[g0]; cur_idx[n_idx++] = table[g1]; cur_idx[n_idx++] = table[g2];
    }
    if (n_idx != 0) FLUSH_MESHLET();
#undef FLUSH_MESHLET
    free(table);
    m->meshlets = meshlets; m->meshlet_count = nm;
    m->meshlet_data = data; m->meshlet_data_words = nd;
}

/* ====================================================================== scene inputs */

/* Engine.cpp:2337-2349: Gribb-Hartmann rows of vp = proj*view, each plane divided by |xyz|. */
void orc_extract_frustum(mcull_globals *g) {
    float vp[16];
    m4_mul(g->proj, g->view, vp);
    /* vp[c][r] in glm notation is M(vp, r, c) */
    for (int c = 0; c < 4; c++) {
        g->frustum[0][c] = M(vp, 3, c) + M(vp, 0, c);
        g->frustum[1][c] = M(vp, 3, c) - M(vp, 0, c);
        g->frustum[2][c] = M(vp, 3, c) + M(vp, 1, c);
        g->frustum[3][c] = M(vp, 3, c) - M(vp, 1, c);
        g->frustum[4][c] = M(vp, 3, c) + M(vp, 2, c);
    }
    for (int i = 0; i < 5; i++) {
        float len = v3_length(v3_make(g->frustum[i][0], g->frustum[i][1], g->frustum[i][2]));
        for (int c = 0; c < 4; c++) g->frustum[i][c] /= len;
    }
}

/* createGlobals: Engine.cpp:2323-2352; glm::lookAt is lookAtRH. */
void orc_create_globals(const float eye[3], const float centre[3], const float up[3], float fov_deg,
                        float z_near, uint32_t width, uint32_t height, int flip_y,
                        uint32_t meshlet_count, mcull_globals *out) {
    memset(out, 0, sizeof *out);
    v3 e = v3_make(eye[0], eye[1], eye[2]);
    v3 f = v3_normalize(v3_sub(v3_make(centre[0], centre[1], centre[2]), e));
    v3 s = v3_normalize(v3_cross(f, v3_make(up[0], up[1], up[2])));
    v3 u = v3_cross(s, f);
    float *V = out->view;
    m4_identity(V);
    M(V, 0, 0) = s.x; M(V, 0, 1) = s.y; M(V, 0, 2) = s.z;
    M(V, 1, 0) = u.x; M(V, 1, 1) = u.y; M(V, 1, 2) = u.z;
    M(V, 2, 0) = -f.x; M(V, 2, 1) = -f.y; M(V, 2, 2) = -f.z;
    M(V, 0, 3) = -v3_dot(s, e); M(V, 1, 3) = -v3_dot(u, e); M(V, 2, 3) = v3_dot(f, e);
    /* glm::radians(deg) = deg * 0.01745329251994329576923690768489f; tan(float) is tanf here */
    float fl = 1.0f / tanf((fov_deg * 0.01745329251994329576923690768489f) / 2.0f);
    float *P = out->proj; /* proj[c][r] in glm notation = M(P, r, c) */
    M(P, 0, 0) = fl / ((float)width / (float)height);
    M(P, 1, 1) = fl;
    M(P, 2, 2) = 0.0f;
    M(P, 3, 2) = -1.0f; /* proj[2][3] */
    M(P, 2, 3) = z_near; /* proj[3][2] */
    if (flip_y) M(P, 1, 1) *= -1;
    out->meshletCount = meshlet_count;
    orc_extract_frustum(out);
}

/* glm::rotate(mat4(1), angle, axis) for a unit coordinate axis (gtc/matrix_transform) */
static void m4_rotate(float angle, v3 axis_in, float *out) {
    float c = cosf(angle), s = sinf(angle);
    v3 axis = v3_normalize(axis_in);
    v3 temp = v3_scale(axis, 1.0f - c);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x; R[0][1] = temp.x * axis.y + s * axis.z; R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z; R[1][1] = c + temp.y * axis.y; R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y; R[2][1] = temp.z * axis.y - s * axis.x; R[2][2] = c + temp.z * axis.z;
    /* Result[k] = m[0]*R[k][0] + m[1]*R[k][1] + m[2]*R[k][2] with m = identity */
    float id[16]; m4_identity(id);
    m4_identity(out);
    for (int k = 0; k < 3; k++)
        for (int r = 0; r < 4; r++)
            M(out, r, k) = (M(id, r, 0) * R[k][0] + M(id, r, 1) * R[k][1]) + M(id, r, 2) * R[k][2];
}

/* updateTransforms: Engine.cpp:2353-2379. */
void orc_update_transforms(uint32_t draw_count, mcull_transform *out) {
    srand(42);
    const float rad = 0.01745329251994329576923690768489f;
    for (uint32_t i = 0; i < draw_count; i++) {
        float ox = ((float)rand() / RAND_MAX) * 15.0f - 7.0f;
        float oy = ((float)rand() / RAND_MAX) * 15.0f - 7.0f;
        float oz = ((float)rand() / RAND_MAX) * 15.0f - 7.0f;
        float scale = 1.0f;
        float rx = ((float)rand() / RAND_MAX) * (30.0f * rad);
        float ry = ((float)rand() / RAND_MAX) * (45.0f * rad);
        float rz = ((float)rand() / RAND_MAX) * (180.0f * rad);
        float model[16], r[16];
        /* glm::translate(mat4(1), v): Result[3] = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3] */
        m4_identity(model);
        {
            float id[16]; m4_identity(id);
            for (int row = 0; row < 4; row++)
                M(model, row, 3) = ((M(id, row, 0) * ox + M(id, row, 1) * oy) + M(id, row, 2) * oz) + M(id, row, 3);
        }
        m4_rotate(rx, v3_make(1, 0, 0), r); m4_mul(model, r, model);
        m4_rotate(ry, v3_make(0, 1, 0), r); m4_mul(model, r, model);
        m4_rotate(rz, v3_make(0, 0, 1), r); m4_mul(model, r, model);
        m4_rotate(90.0f * rad, v3_make(1, 0, 0), r); m4_mul(model, r, model);
        m4_identity(r); r[0] = r[5] = r[10] = scale; m4_mul(model, r, model);
        memcpy(out[i].model, model, 64);
    }
}

/* createIndirectBuffer: Engine.cpp:1466-1484. */
void orc_create_indirect_buffer(uint32_t draw_count, uint32_t meshlet_count, uint32_t index_count,
                                mcull_draw_indirect *out) {
    uint32_t task_workgroups = (meshlet_count + 31) / 32;
    for (uint32_t i = 0; i < draw_count; i++) {
        out[i].indexCount = index_count; out[i].instanceCount = 1; out[i].vertexOffset = 0;
        out[i].firstInstance = 0; out[i].firstIndex = 0;
        out[i].commandMeshIndirect.firstTask = 0;
        out[i].commandMeshIndirect.taskCount = task_workgroups;
    }
}

/* ====================================================================== Hi-Z pyramid */

/* Engine.cpp:1651: floor(log2(max(w,h))) + 1, evaluated exactly on the integer. */
uint32_t orc_hiz_level_count(uint32_t w, uint32_t h) {
    uint32_t m = w > h ? w : h, l = 0;
    while (m > 1) { m >>= 1; l++; }
    return l + 1;
}

uint64_t orc_hiz_layout(uint32_t w, uint32_t h, uint32_t *levels, uint32_t lw[16], uint32_t lh[16],
                        uint64_t off[16]) {
    uint32_t n = orc_hiz_level_count(w, h);
    if (n > 16) n = 16;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        lw[i] = (w >> i) ? (w >> i) : 1; /* std::max(1u, width>>i), Engine.cpp:1845-1846 */
        lh[i] = (h >> i) ? (h >> i) : 1;
        off[i] = total;
        total += (uint64_t)lw[i] * lh[i];
    }
    if (levels) *levels = n;
    return total;
}

/* buildHiZPyramid: one pass per level in order (Engine.cpp:1840-1865); per texel
 * shader.comp:12-31.  Source reads beyond the source extent (possible only when one
 * dimension collapsed to 1 before the other) clamp to the edge. */
void orc_hiz_build(const float *depth, uint32_t w, uint32_t h, float *pyramid, int threads) {
    uint32_t levels, lw[16], lh[16]; uint64_t off[16];
    orc_hiz_layout(w, h, &levels, lw, lh, off);
    (void)threads;
    for (uint32_t i = 0; i < levels; i++) {
        float *dst = pyramid + off[i];
        if (i == 0) { /* downsampling == 0: copy (shader.comp:19-21) */
#pragma omp parallel for num_threads(threads > 1 ? threads : 1) if (threads > 1)
            for (int64_t y = 0; y < (int64_t)lh[0]; y++)
                memcpy(dst + (uint64_t)y * lw[0], depth + (uint64_t)y * w, (size_t)lw[0] * 4);
            continue;
        }
        const float *src = pyramid + off[i - 1];
        uint32_t sw = lw[i - 1], sh = lh[i - 1];
#pragma omp parallel for num_threads(threads > 1 ? threads : 1) if (threads > 1)
        for (int64_t y = 0; y < (int64_t)lh[i]; y++) {
            uint32_t y0 = (uint32_t)(2 * y), y1 = y0 + 1;
            if (y0 >= sh) y0 = sh - 1;
            if (y1 >= sh) y1 = sh - 1;
            for (uint32_t x = 0; x < lw[i]; x++) {
                uint32_t x0 = 2 * x, x1 = x0 + 1;
                if (x0 >= sw) x0 = sw - 1;
                if (x1 >= sw) x1 = sw - 1;
                float d0 = src[(uint64_t)y0 * sw + x0], d1 = src[(uint64_t)y0 * sw + x1];
                float d2 = src[(uint64_t)y1 * sw + x0], d3 = src[(uint64_t)y1 * sw + x1];
                float a = d1 < d0 ? d1 : d0; /* GLSL min(x,y) = y < x ? y : x */
                float b = d3 < d2 ? d3 : d2;
                dst[(uint64_t)y * lw[i] + x] = b < a ? b : a;
            }
        }
    }
}

/* ====================================================================== per-pair tests */

void orc_view_init(orc_view *v, const mcull_globals *g, const float *pyramid, uint32_t w0, uint32_t h0) {
    memset(v, 0, sizeof *v);
    v->g = *g;
    float inv[16];
    m4_inverse(g->view, inv);                       /* shader.task:88 */
    v->cam_pos[0] = M(inv, 0, 3); v->cam_pos[1] = M(inv, 1, 3); v->cam_pos[2] = M(inv, 2, 3); /* :89 */
    m4_mul(g->proj, g->view, v->vp);                /* shader.task:60, (proj*view) first */
    v->pyramid = pyramid; v->w0 = w0; v->h0 = h0;
    if (pyramid) orc_hiz_layout(w0, h0, &v->levels, v->lw, v->lh, v->off);
}

/* shader.task:71-72: clamp(ceil(log2(max(px,1))), 0, levels-1), exact: for finite x > 1,
 * x = 1.f * 2^E so ceil(log2 x) = E when the mantissa field is zero, else E + 1. */
uint32_t orc_mip_level(float px, uint32_t levels) {
    if (!(px > 1.0f)) return 0; /* includes NaN: max(NaN,1) is taken as 1 */
    uint32_t bits; memcpy(&bits, &px, 4);
    uint32_t e = (bits >> 23) & 0xffu, mant = bits & 0x7fffffu;
    uint32_t l = (e == 0xffu) ? 0xffu : (e - 127u) + (mant ? 1u : 0u);
    return l > levels - 1 ? levels - 1 : l;
}

/* nearest + REPEAT (Engine.cpp:1623-1627): i = floor(u * size) mod size.  Coordinates whose
 * scaled value is not a finite float below 1e9 in magnitude address texel 0. */
static inline uint32_t wrap_texel(float scaled_floor, uint32_t size) {
    if (!(fabsf(scaled_floor) < 1.0e9f)) return 0;
    int32_t i = (int32_t)scaled_floor;
    int32_t mod = i % (int32_t)size;
    if (mod < 0) mod += (int32_t)size;
    return (uint32_t)mod;
}

void orc_cull_pair(const orc_view *v, const float model[16], const float cone[4], const float centre[3],
                   float radius, uint32_t tests, orc_pair_result *out) {
    memset(out, 0, sizeof *out);
    /* shader.task:91 */
    v3 c = m4_transform_point(model, v3_make(centre[0], centre[1], centre[2]));
    /* :95 */
    v3 axis = v3_normalize(m4_transform_dir(model, v3_make(cone[0], cone[1], cone[2])));
    float cone_w = cone[3];                                                  /* :97 */
    float r = radius * v3_length(v3_make(model[0], model[1], model[2]));     /* :99 */

    /* frustumCull, shader.task:43-56 */
    float fmin = INFINITY; int fc = 0;
    for (int i = 0; i < 5; i++) {
        const float *p = v->g.frustum[i];
        float d = (v3_dot(v3_make(p[0], p[1], p[2]), c) + p[3]) + r;
        if (d < 0.0f) fc = 1;
        if (d < fmin || !(fmin == fmin)) fmin = d;
    }
    out->frustum_culled = fc; out->frustum_margin = fmin;

    /* occlusionCull, shader.task:58-77 */
    int oc = 0;
    if (v->pyramid) {
        const float *vp = v->vp;
        float cx = ((M(vp,0,0) * c.x + M(vp,0,1) * c.y) + M(vp,0,2) * c.z) + M(vp,0,3);
        float cy = ((M(vp,1,0) * c.x + M(vp,1,1) * c.y) + M(vp,1,2) * c.z) + M(vp,1,3);
        float cz = ((M(vp,2,0) * c.x + M(vp,2,1) * c.y) + M(vp,2,2) * c.z) + M(vp,2,3);
        float cw = ((M(vp,3,0) * c.x + M(vp,3,1) * c.y) + M(vp,3,2) * c.z) + M(vp,3,3);
        float nx = cx / cw, ny = cy / cw, nz = cz / cw;                      /* :62 */
        float ndc_radius = (M(v->g.proj, 1, 1) * r) / cw;                    /* :63 */
        float u = nx * 0.5f + 0.5f;                                          /* :65 */
        float w = 1.0f - (ny * 0.5f + 0.5f);                                 /* :66 */
        float tex_max = (float)(v->w0 > v->h0 ? v->w0 : v->h0);              /* :68 */
        float px = (ndc_radius * 2.0f) * tex_max;                            /* :69 */
        uint32_t mip = orc_mip_level(px, v->levels);                         /* :71-72 */
        uint32_t tx = wrap_texel(floorf(u * (float)v->lw[mip]), v->lw[mip]);
        uint32_t ty = wrap_texel(floorf(w * (float)v->lh[mip]), v->lh[mip]);
        float d = v->pyramid[v->off[mip] + (uint64_t)ty * v->lw[mip] + tx];  /* :74 */
        float thr = d - 0.05f;
        oc = nz < thr;                                                       /* :76 */
        out->occlusion_margin = nz - thr;
        out->mip = mip; out->tx = tx; out->ty = ty;
    }
    out->occlusion_culled = oc;

    /* coneCull, shader.task:37-41 */
    v3 cam = v3_make(v->cam_pos[0], v->cam_pos[1], v->cam_pos[2]);
    v3 view_dir = v3_normalize(v3_sub(cam, c));
    float dp = v3_dot(axis, view_dir);
    out->cone_culled = dp < -cone_w;
    out->cone_margin = dp + cone_w;

    int culled = 0;
    if ((tests & MCULL_TEST_FRUSTUM) && out->frustum_culled) culled = 1;
    if ((tests & MCULL_TEST_HIZ) && out->occlusion_culled) culled = 1;
    if ((tests & MCULL_TEST_CONE) && out->cone_culled) culled = 1;
    out->visible = !culled;                                                  /* :101 */
}

/* ====================================================================== scene cull */

static inline void scene_bounds(const orc_scene *s, uint64_t m, float cone[4], float centre[3], float *radius) {
    if (s->meshlets) {
        memcpy(cone, s->meshlets[m].cone, 16);
        memcpy(centre, s->meshlets[m].coneApex, 12);
        *radius = s->meshlets[m].radius;
    } else {
        memcpy(cone, s->cone_axis_w + 4 * m, 16);
        memcpy(centre, s->centre_radius + 4 * m, 12);
        *radius = s->centre_radius[4 * m + 3];
    }
}

int orc_cull_scene(const orc_scene *s, const orc_view *v, uint32_t tests, int threads, orc_cull_output *out) {
    uint64_t I = s->instance_count;
    uint64_t *group_base = (uint64_t *)malloc((I + 1) * 8);
    uint64_t *inst_count = (uint64_t *)calloc(I + 1, 8);
    uint64_t groups = 0, pairs = 0;
    for (uint64_t i = 0; i < I; i++) {
        uint32_t cnt = s->range_count ? s->range_count[i] : v->g.meshletCount;
        group_base[i] = groups;
        groups += (cnt + 63) / 64;
        pairs += cnt;
    }
    group_base[I] = groups;
    out->n_groups = groups; out->n_pairs = pairs;
    uint64_t *mask = out->mask ? out->mask : (uint64_t *)malloc((groups ? groups : 1) * 8);
    uint64_t cf = 0, co = 0, cc = 0;
    int nt = threads > 1 ? threads : 1;
    if (out->margins) nt = 1; /* pair-ordered margin output is filled by the scalar loop */
    uint64_t *pair_base = NULL;
    if (out->margins) {
        pair_base = (uint64_t *)malloc((I + 1) * 8);
        uint64_t p = 0;
        for (uint64_t i = 0; i < I; i++) { pair_base[i] = p; p += s->range_count ? s->range_count[i] : v->g.meshletCount; }
    }
    /* pass 1: shader.task:85-101 per pair -> one ballot word per 64 meshlets */
#pragma omp parallel for num_threads(nt) schedule(static) reduction(+ : cf, co, cc) if (nt > 1)
    for (int64_t i = 0; i < (int64_t)I; i++) {
        uint32_t off = s->range_offset ? s->range_offset[i] : 0;
        uint32_t cnt = s->range_count ? s->range_count[i] : v->g.meshletCount;
        const float *model = s->transforms[i].model;
        uint64_t vis = 0;
        for (uint32_t g = 0; g * 64 < cnt; g++) {
            uint64_t word = 0;
            uint32_t lim = cnt - g * 64 < 64 ? cnt - g * 64 : 64;
            for (uint32_t l = 0; l < lim; l++) {
                float cone[4], centre[3], radius;
                scene_bounds(s, (uint64_t)off + g * 64 + l, cone, centre, &radius);
                orc_pair_result r;
                orc_cull_pair(v, model, cone, centre, radius, tests, &r);
                if (r.visible) word |= 1ull << l;
                cf += (uint64_t)r.frustum_culled; co += (uint64_t)r.occlusion_culled; cc += (uint64_t)r.cone_culled;
                if (out->margins) {
                    float *mg = out->margins + 3 * (pair_base[i] + g * 64 + l);
                    mg[0] = r.frustum_margin; mg[1] = r.occlusion_margin; mg[2] = r.cone_margin;
                }
            }
            mask[group_base[i] + g] = word;
            vis += (uint64_t)__builtin_popcountll(word);
        }
        inst_count[i] = vis;
    }
    /* exclusive scan in instance order */
    uint64_t total = 0;
    for (uint64_t i = 0; i < I; i++) { uint64_t c = inst_count[i]; inst_count[i] = total; total += c; }
    inst_count[I] = total;
    out->total = total;
    out->culled_frustum = cf; out->culled_occlusion = co; out->culled_cone = cc;
    int overflow = 0;
    if (out->commands) {
        for (uint64_t i = 0; i < I; i++) {
            out->commands[i].taskCount = (uint32_t)(inst_count[i + 1] - inst_count[i]);
            out->commands[i].firstTask = (uint32_t)inst_count[i];
        }
    }
    /* pass 2: compaction, shader.task:103-118 made global */
    if (out->list) {
        if (total > out->list_capacity) overflow = 1;
        else {
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
            for (int64_t i = 0; i < (int64_t)I; i++) {
                uint32_t off = s->range_offset ? s->range_offset[i] : 0;
                uint64_t o = inst_count[i];
                for (uint64_t g = group_base[i]; g < group_base[i + 1]; g++) {
                    uint64_t word = mask[g];
                    while (word) {
                        uint32_t l = (uint32_t)__builtin_ctzll(word);
                        word &= word - 1;
                        out->list[o].drawID = (uint32_t)i;
                        out->list[o].meshletID = off + (uint32_t)(g - group_base[i]) * 64 + l;
                        o++;
                    }
                }
            }
        }
    }
    if (!out->mask) free(mask);
    free(group_base); free(inst_count); free(pair_base);
    return overflow ? -1 : 0;
}

/* shader.task:79-119 literally, one payload per 32-thread task workgroup. */
uint64_t orc_task_payloads(const orc_scene *s, const orc_view *v, uint32_t tests, mcull_task_payload *out) {
    uint32_t mc = v->g.meshletCount;
    uint32_t groups = (mc + 31) / 32; /* Engine.cpp:1475 */
    uint64_t n = 0;
    for (uint64_t draw = 0; draw < s->instance_count; draw++) {
        for (uint32_t wg = 0; wg < groups; wg++) {
            mcull_task_payload p; memset(&p, 0, sizeof p);
            uint32_t count = 0;
            for (uint32_t tid = 0; tid < 32; tid++) {
                uint32_t meshlet_id = wg * 32 + tid;
                int accept = 0;
                if (meshlet_id < mc) {
                    float cone[4], centre[3], radius;
                    scene_bounds(s, meshlet_id, cone, centre, &radius);
                    orc_pair_result r;
                    orc_cull_pair(v, s->transforms[draw].model, cone, centre, radius, tests, &r);
                    accept = r.visible;
                }
                if (accept) p.meshletIndices[count++] = meshlet_id; /* ballot + exclusive bit count */
            }
            p.drawID = (uint32_t)draw;
            p.taskCount = count;
            out[n++] = p;
        }
    }
    return n;
}
