// host_scene.hip — the per-frame host inputs of the culling path, built the way the reference's
// Engine builds them: createGlobals (Engine.cpp:2323-2352), updateTransforms (Engine.cpp:2353-2379)
// and createIndirectBuffer (Engine.cpp:1466-1484).  Host code only; no GPU call is made here.
//
// Arithmetic is IEEE binary32 without contraction (the file is compiled with -ffp-contract=off) in
// the association order glm's column-major templates evaluate: a dot product is (x + y) + z, a
// matrix product sums its four column terms left to right, normalize(v) is v * (1 / sqrt(dot(v,v))).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "../../include/mcull.h"

namespace {

struct V3 {
    float x, y, z;
};

// column-major 4x4: c[j][i] = column j, row i (glm::mat4 indexing)
struct M4 {
    float c[4][4];
};

inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V3 normalize(V3 a) {
    const float inv = 1.0f / sqrtf(dot(a, a));
    return {a.x * inv, a.y * inv, a.z * inv};
}
// glm::radians: degrees * (pi / 180) with the constant rounded to float first
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

M4 identity() {
    M4 m;
    memset(&m, 0, sizeof(m));
    m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.0f;
    return m;
}

M4 mul(const M4 &a, const M4 &b) {
    M4 o;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            o.c[j][i] = ((a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1]) + a.c[2][i] * b.c[j][2]) + a.c[3][i] * b.c[j][3];
    return o;
}

// glm::translate(m, v): only the last column changes
M4 translate(const M4 &m, V3 v) {
    M4 o = m;
    for (int i = 0; i < 4; i++) o.c[3][i] = ((m.c[0][i] * v.x + m.c[1][i] * v.y) + m.c[2][i] * v.z) + m.c[3][i];
    return o;
}

// glm::rotate(m, angle, axis)
M4 rotate(const M4 &m, float angle, V3 axis_in) {
    const float c = cosf(angle), s = sinf(angle);
    const V3 a = normalize(axis_in);
    const float k = 1.0f - c;
    const V3 t = {k * a.x, k * a.y, k * a.z};
    float r[3][3];
    r[0][0] = c + t.x * a.x;
    r[0][1] = t.x * a.y + s * a.z;
    r[0][2] = t.x * a.z - s * a.y;
    r[1][0] = t.y * a.x - s * a.z;
    r[1][1] = c + t.y * a.y;
    r[1][2] = t.y * a.z + s * a.x;
    r[2][0] = t.z * a.x + s * a.y;
    r[2][1] = t.z * a.y - s * a.x;
    r[2][2] = c + t.z * a.z;
    M4 o = m;
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 4; i++) o.c[j][i] = (m.c[0][i] * r[j][0] + m.c[1][i] * r[j][1]) + m.c[2][i] * r[j][2];
    return o;
}

// glm::scale(m, v)
M4 scale(const M4 &m, V3 v) {
    M4 o = m;
    for (int i = 0; i < 4; i++) {
        o.c[0][i] = m.c[0][i] * v.x;
        o.c[1][i] = m.c[1][i] * v.y;
        o.c[2][i] = m.c[2][i] * v.z;
    }
    return o;
}

M4 load(const float *p) {
    M4 m;
    memcpy(&m, p, sizeof(m));
    return m;
}

}  // namespace

extern "C" {

int mcull_extract_frustum(mcull_globals *g) {
    if (!g) return MCULL_E_INVALID;
    const M4 vp = mul(load(g->proj), load(g->view));
    for (int j = 0; j < 4; j++) {
        g->frustum[0][j] = vp.c[j][3] + vp.c[j][0];  // left
        g->frustum[1][j] = vp.c[j][3] - vp.c[j][0];  // right
        g->frustum[2][j] = vp.c[j][3] + vp.c[j][1];  // bottom
        g->frustum[3][j] = vp.c[j][3] - vp.c[j][1];  // top
        g->frustum[4][j] = vp.c[j][3] + vp.c[j][2];  // near
    }
    for (int i = 0; i < 5; i++) {
        const V3 n = {g->frustum[i][0], g->frustum[i][1], g->frustum[i][2]};
        const float len = sqrtf(dot(n, n));
        for (int j = 0; j < 4; j++) g->frustum[i][j] = g->frustum[i][j] / len;
    }
    return MCULL_OK;
}

int mcull_create_globals(const float eye[3], const float centre[3], const float up[3], float fov_deg, float z_near,
                         uint32_t width, uint32_t height, int flip_y, uint32_t meshlet_count, mcull_globals *out) {
    if (!out || width == 0 || height == 0) return MCULL_E_INVALID;
    // the reference's camera: eye (2,2,2) looking at the origin, +Z up (Engine.cpp:2325, Engine.hpp cameraPos)
    const V3 e = eye ? V3{eye[0], eye[1], eye[2]} : V3{2.0f, 2.0f, 2.0f};
    const V3 ctr = centre ? V3{centre[0], centre[1], centre[2]} : V3{0.0f, 0.0f, 0.0f};
    const V3 upv = up ? V3{up[0], up[1], up[2]} : V3{0.0f, 0.0f, 1.0f};
    memset(out, 0, sizeof(*out));
    // glm::lookAt, right-handed
    const V3 f = normalize(sub(ctr, e));
    const V3 s = normalize(cross(f, upv));
    const V3 u = cross(s, f);
    M4 view = identity();
    view.c[0][0] = s.x; view.c[1][0] = s.y; view.c[2][0] = s.z;
    view.c[0][1] = u.x; view.c[1][1] = u.y; view.c[2][1] = u.z;
    view.c[0][2] = -f.x; view.c[1][2] = -f.y; view.c[2][2] = -f.z;
    view.c[3][0] = -dot(s, e);
    view.c[3][1] = -dot(u, e);
    view.c[3][2] = dot(f, e);
    // infinite far plane, reverse Z (Engine.cpp:2327-2335)
    const float fl = 1.0f / tanf(radians(fov_deg) / 2.0f);
    M4 proj;
    memset(&proj, 0, sizeof(proj));
    proj.c[0][0] = fl / ((float)width / (float)height);
    proj.c[1][1] = flip_y ? -fl : fl;
    proj.c[2][3] = -1.0f;
    proj.c[3][2] = z_near;
    memcpy(out->view, &view, sizeof(view));
    memcpy(out->proj, &proj, sizeof(proj));
    out->meshletCount = meshlet_count;
    return mcull_extract_frustum(out);
}

int mcull_reference_transforms(uint32_t draw_count, uint32_t seed, mcull_transform *out) {
    if (!out && draw_count) return MCULL_E_INVALID;
    // srand(seed) / rand() of glibc without touching the process-wide generator: rand() is random()
    // on the 128-byte (TYPE_3) state, which initstate_r reproduces on a private buffer.
    struct random_data rd;
    char state[128];
    memset(&rd, 0, sizeof(rd));
    memset(state, 0, sizeof(state));
    if (initstate_r(seed, state, sizeof(state), &rd) != 0) return MCULL_E_INVALID;
    auto next01 = [&]() {
        int32_t r = 0;
        random_r(&rd, &r);
        return (float)r / (float)RAND_MAX;
    };
    const V3 ax = {1.0f, 0.0f, 0.0f}, ay = {0.0f, 1.0f, 0.0f}, az = {0.0f, 0.0f, 1.0f};
    for (uint32_t i = 0; i < draw_count; i++) {
        const float ox = next01() * 15.0f - 7.0f;
        const float oy = next01() * 15.0f - 7.0f;
        const float oz = next01() * 15.0f - 7.0f;
        const float rx = next01() * radians(30.0f);
        const float ry = next01() * radians(45.0f);
        const float rz = next01() * radians(180.0f);
        M4 m = translate(identity(), V3{ox, oy, oz});
        m = mul(m, rotate(identity(), rx, ax));
        m = mul(m, rotate(identity(), ry, ay));
        m = mul(m, rotate(identity(), rz, az));
        m = mul(m, rotate(identity(), radians(90.0f), ax));  // stands the model upright (Engine.cpp:2374)
        m = mul(m, scale(identity(), V3{1.0f, 1.0f, 1.0f}));
        memcpy(out[i].model, &m, sizeof(m));
    }
    return MCULL_OK;
}

int mcull_create_indirect_buffer(uint32_t draw_count, uint32_t meshlet_count, uint32_t index_count, mcull_draw_indirect *out) {
    if (!out && draw_count) return MCULL_E_INVALID;
    const uint32_t task_groups = (meshlet_count + 31u) / 32u;  // one task group per 32 meshlets
    for (uint32_t i = 0; i < draw_count; i++) {
        memset(&out[i], 0, sizeof(out[i]));
        out[i].indexCount = index_count;
        out[i].instanceCount = 1;
        out[i].commandMeshIndirect.taskCount = task_groups;
        out[i].commandMeshIndirect.firstTask = 0;
    }
    return MCULL_OK;
}

}  // extern "C"
