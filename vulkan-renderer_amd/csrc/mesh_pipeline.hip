// Load-time mesh pipeline of the reference, host side (it is host code in the reference too):
//   loadModel          Engine.cpp:243-282   OBJ -> one Vertex per face corner, deduplicated in first-seen order
//   optimizeGeometry   Engine.cpp:338-471   the reference's variant of Forsyth's vertex-cache reordering
//   createMeshlets     Engine.cpp:548-628   greedy split into meshlets of <= 64 vertices / <= 124 triangles
// The per-meshlet cone/sphere (buildMeshletCon, Engine.cpp:472-547) is NOT computed here: meshlets
// leave this file with zero bounds and mcull_mesh_upload runs the GPU bounds kernel on them.
//
// Results are the reference's exactly.  Two things are done differently for speed, neither of which
// can change a result: vertices are looked up in a hash table keyed on the same equality as
// Vertex::operator== (Engine.hpp:57-61; the ids depend only on first-seen order, never on the hash),
// and the reorderer's fallback "scan every triangle for the best score" (Engine.cpp:401-408) reads
// the answer from a lazily maintained max-heap ordered by (score, lower triangle id first), which is
// the triangle that scan's strict `>` comparison selects.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

#include "mcull_internal.h"

struct mcull_mesh {
    std::vector<mcull_vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<mcull_meshlet> meshlets;
    std::vector<uint32_t> meshlet_data;
};

namespace {

// utils.hpp:94-99
uint16_t float_to_half(float f) {
    const _Float16 h = (_Float16)f;
    uint16_t bits;
    std::memcpy(&bits, &h, sizeof bits);
    return bits;
}

// uint8_t((n*0.5f + 0.5f)*255.0f), Engine.cpp:266-268.  The conversion of a float outside [0, 256)
// is undefined in C++; the reference's x86 build converts through a 32-bit signed integer and keeps
// the low byte, and gives 0 for NaN and for magnitudes beyond the integer range (the bunny asset has
// no normals, so normalize(0) = NaN reaches this line).
uint8_t quantise(float n) {
    const float s = (n * 0.5f + 0.5f) * 255.0f;
    if (std::isnan(s) || s <= -1.0f || s >= 2147483648.0f) return 0;
    return (uint8_t)(int32_t)s;
}

struct VertexKey {
    uint32_t w[5];
    bool operator==(const VertexKey &o) const { return std::memcmp(w, o.w, sizeof w) == 0; }
};
struct VertexKeyHash {
    size_t operator()(const VertexKey &k) const {
        uint64_t h = 0x9e3779b97f4a7c15ull;
        for (uint32_t x : k.w) h = (h ^ x) * 0xff51afd7ed558ccdull, h ^= h >> 29;
        return (size_t)h;
    }
};

// The key reproduces Vertex::operator==: float compare (so -0 equals +0; a NaN position never equals
// anything, such corners always create a new vertex), bytes and half bits compared as integers.
bool make_key(const mcull_vertex &v, VertexKey *k) {
    const float p[3] = {v.vx, v.vy, v.vz};
    for (int i = 0; i < 3; i++) {
        if (std::isnan(p[i])) return false;
        const float f = p[i] == 0.0f ? 0.0f : p[i];
        std::memcpy(&k->w[i], &f, 4);
    }
    k->w[3] = (uint32_t)v.nx | ((uint32_t)v.ny << 8) | ((uint32_t)v.nz << 16);
    k->w[4] = (uint32_t)v.tu | ((uint32_t)v.tv << 16);
    return true;
}

int build_vertices(const float *pos, uint64_t npos, const float *nrm, uint64_t nnrm, const float *uv, uint64_t nuv,
                   const int32_t *corners, uint64_t ncorners, mcull_mesh *m) {
    std::unordered_map<VertexKey, uint32_t, VertexKeyHash> unique;
    unique.reserve(ncorners / 2 + 16);
    m->indices.reserve(ncorners);
    for (uint64_t c = 0; c < ncorners; c++) {
        const int32_t vi = corners[3 * c], ti = corners[3 * c + 1], ni = corners[3 * c + 2];
        if (vi < 0 || (uint64_t)vi >= npos) return MCULL_E_INVALID;
        mcull_vertex v;
        std::memset(&v, 0, sizeof v);  // Vertex vertex{};
        v.vx = pos[3 * vi];
        v.vy = pos[3 * vi + 1];
        v.vz = pos[3 * vi + 2];
        float n[3] = {0.0f, 0.0f, 0.0f};
        if (nnrm) {  // !attrib.normals.empty()
            if (ni < 0 || (uint64_t)ni >= nnrm) return MCULL_E_INVALID;
            n[0] = nrm[3 * ni]; n[1] = nrm[3 * ni + 1]; n[2] = nrm[3 * ni + 2];
        }
        // glm::normalize = v * inversesqrt(dot(v, v)); dot summed left to right
        const float inv = 1.0f / std::sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
        v.nx = quantise(n[0] * inv);
        v.ny = quantise(n[1] * inv);
        v.nz = quantise(n[2] * inv);
        if (nuv) {  // !attrib.texcoords.empty()
            if (ti < 0 || (uint64_t)ti >= nuv) return MCULL_E_INVALID;
            v.tu = float_to_half(uv[2 * ti]);
            v.tv = float_to_half(1.0f - uv[2 * ti + 1]);
        }
        VertexKey key;
        uint32_t id;
        if (make_key(v, &key)) {
            auto it = unique.find(key);
            if (it == unique.end()) {
                id = (uint32_t)m->vertices.size();
                unique.emplace(key, id);
                m->vertices.push_back(v);
            } else {
                id = it->second;
            }
        } else {
            id = (uint32_t)m->vertices.size();
            m->vertices.push_back(v);
        }
        m->indices.push_back(id);
    }
    return MCULL_OK;
}

// computeVertexScore, Engine.cpp:308-335
float vertex_score(int cache_position, int valence) {
    if (valence == 0) return -1.0f;
    float cache_score = 0.0f;
    if (cache_position >= 3) {
        const float normalized = (cache_position - 3) * (1.0f / (32 - 3));
        cache_score = std::pow(1.0f - normalized, 1.5f);
    } else if (cache_position >= 0) {
        cache_score = 0.75f;
    }
    return cache_score + 2.0f * std::pow(float(valence), -0.5f);
}

struct HeapEntry {
    float score;
    uint32_t tri;
    // max-heap: higher score first, among equal scores the lower triangle id
    bool operator<(const HeapEntry &o) const { return score < o.score || (score == o.score && tri > o.tri); }
};

void optimize(mcull_mesh *m) {
    std::vector<uint32_t> &ind = m->indices;
    const uint32_t n_tri = (uint32_t)(ind.size() / 3), n_vert = (uint32_t)m->vertices.size();
    if (n_tri == 0) return;
    // vertex -> triangles, ascending triangle id (CSR form of vertexToTriangles)
    std::vector<uint32_t> start(n_vert + 1, 0), adj(ind.size());
    for (uint32_t x : ind) start[x + 1]++;
    for (uint32_t v = 0; v < n_vert; v++) start[v + 1] += start[v];
    {
        std::vector<uint32_t> fill(start.begin(), start.end() - 1);
        for (uint32_t t = 0; t < n_tri; t++)
            for (int k = 0; k < 3; k++) adj[fill[ind[3 * t + k]]++] = t;
    }
    std::vector<int> valence(n_vert), cache_pos(n_vert, -1);
    std::vector<float> vscore(n_vert), tscore(n_tri);
    std::vector<uint8_t> done(n_tri, 0);
    for (uint32_t v = 0; v < n_vert; v++) valence[v] = (int)(start[v + 1] - start[v]);
    for (uint32_t v = 0; v < n_vert; v++) vscore[v] = vertex_score(-1, valence[v]);
    auto tri_score = [&](uint32_t t) { return vscore[ind[3 * t]] + vscore[ind[3 * t + 1]] + vscore[ind[3 * t + 2]]; };
    std::priority_queue<HeapEntry> heap;
    {
        std::vector<HeapEntry> init(n_tri);
        for (uint32_t t = 0; t < n_tri; t++) {
            tscore[t] = tri_score(t);
            init[t] = {tscore[t], t};
        }
        heap = std::priority_queue<HeapEntry>(std::less<HeapEntry>(), std::move(init));
    }
    const float kFloor = std::numeric_limits<float>::min();  // the reference's initial "best" (smallest positive float)
    std::vector<uint32_t> cache, out;
    cache.reserve(36);
    out.reserve(ind.size());
    int64_t next_best = -1;
    for (uint32_t step = 0; step < n_tri; step++) {
        int64_t best = next_best;
        if (best < 0 || done[best]) {
            // Engine.cpp:401-408: first triangle, in id order, with the highest score above kFloor
            best = -1;
            while (!heap.empty()) {
                const HeapEntry e = heap.top();
                if (done[e.tri] || e.score != tscore[e.tri]) {
                    heap.pop();  // stale entry
                    continue;
                }
                if (e.score > kFloor) best = e.tri;
                break;
            }
            if (best < 0) {
                // no unfinished triangle scores above the floor; the reference would index with -1 here.
                // Cannot happen: an unfinished triangle's vertices all have valence >= 1, hence score > 0.
                break;
            }
        }
        const uint32_t tv[3] = {ind[3 * best], ind[3 * best + 1], ind[3 * best + 2]};
        out.insert(out.end(), tv, tv + 3);
        done[best] = 1;
        uint32_t affected[6];
        int n_aff = 0;
        for (uint32_t v : tv) {
            auto it = std::find(cache.begin(), cache.end(), v);
            if (it != cache.end()) cache.erase(it);
            cache.insert(cache.begin(), v);
            if (cache.size() > 32) {
                const uint32_t evicted = cache.back();
                cache.pop_back();
                cache_pos[evicted] = -1;
                affected[n_aff++] = evicted;
            }
            affected[n_aff++] = v;
            valence[v]--;
        }
        for (size_t i = 0; i < cache.size(); i++) cache_pos[cache[i]] = (int)i;
        // only the affected vertices are rescored (Engine.cpp:447-456); vertices that merely moved
        // inside the cache keep their previous score
        for (int a = 0; a < n_aff; a++) {
            const uint32_t v = affected[a];
            vscore[v] = vertex_score(cache_pos[v], valence[v]);
            for (uint32_t i = start[v]; i < start[v + 1]; i++) {
                const uint32_t t = adj[i];
                if (done[t]) continue;
                const float s = tri_score(t);
                if (s != tscore[t]) {
                    tscore[t] = s;
                    heap.push({s, t});
                }
            }
        }
        // a heap that has grown far beyond the live triangles is rebuilt from the current scores
        if (heap.size() > 8ull * n_tri + 1024) {
            std::vector<HeapEntry> live;
            for (uint32_t t = 0; t < n_tri; t++)
                if (!done[t]) live.push_back({tscore[t], t});
            heap = std::priority_queue<HeapEntry>(std::less<HeapEntry>(), std::move(live));
        }
        next_best = -1;
        float best_score = kFloor;
        for (uint32_t v : cache)
            for (uint32_t i = start[v]; i < start[v + 1]; i++) {
                const uint32_t t = adj[i];
                if (!done[t] && tscore[t] > best_score) {
                    best_score = tscore[t];
                    next_best = t;
                }
            }
    }
    ind.swap(out);
}

void flush_meshlet(mcull_mesh *m, std::vector<uint32_t> &verts, std::vector<uint8_t> &local) {
    mcull_meshlet ml;
    std::memset(&ml, 0, sizeof ml);
    ml.dataOffset = (uint32_t)m->meshlet_data.size();
    ml.triangleCount = (uint8_t)(local.size() / 3);
    ml.vertexCount = (uint8_t)verts.size();
    while (local.size() % 4) local.push_back(0);
    m->meshlet_data.insert(m->meshlet_data.end(), verts.begin(), verts.end());
    for (size_t j = 0; j < local.size(); j += 4) {
        uint32_t packed;
        std::memcpy(&packed, &local[j], 4);
        m->meshlet_data.push_back(packed);
    }
    m->meshlets.push_back(ml);
}

void build_meshlets(mcull_mesh *m) {
    m->meshlets.clear();
    m->meshlet_data.clear();
    std::vector<uint32_t> verts;
    std::vector<uint8_t> local;
    std::vector<uint8_t> slot(m->vertices.size(), 0xff);
    for (size_t i = 0; i + 2 < m->indices.size(); i += 3) {
        const uint32_t g[3] = {m->indices[i], m->indices[i + 1], m->indices[i + 2]};
        // counted per corner, as the reference does: a vertex repeated inside one triangle counts twice
        const int fresh = (slot[g[0]] == 0xff) + (slot[g[1]] == 0xff) + (slot[g[2]] == 0xff);
        if (verts.size() + fresh > 64 || local.size() / 3 >= 124) {
            flush_meshlet(m, verts, local);
            for (uint32_t v : verts) slot[v] = 0xff;
            verts.clear();
            local.clear();
        }
        for (uint32_t v : g)
            if (slot[v] == 0xff) {
                slot[v] = (uint8_t)verts.size();
                verts.push_back(v);
            }
        for (uint32_t v : g) local.push_back(slot[v]);
    }
    if (!local.empty()) flush_meshlet(m, verts, local);
}

}  // namespace

extern "C" {

int mcull_mesh_from_arrays(const float *positions, uint64_t n_positions, const float *normals, uint64_t n_normals,
                           const float *texcoords, uint64_t n_texcoords, const int32_t *corners, uint64_t n_corners,
                           mcull_mesh **out) {
    if (!out) return MCULL_E_INVALID;
    *out = nullptr;
    if ((!positions && n_positions) || (!normals && n_normals) || (!texcoords && n_texcoords) || (!corners && n_corners) ||
        n_corners % 3 != 0 || n_corners > 0xfffffff0ull)
        return MCULL_E_INVALID;
    mcull_mesh *m = new (std::nothrow) mcull_mesh;
    if (!m) return MCULL_E_NOMEM;
    const int rc = build_vertices(positions, n_positions, normals, n_normals, texcoords, n_texcoords, corners, n_corners, m);
    if (rc) {
        delete m;
        return rc;
    }
    *out = m;
    return MCULL_OK;
}

// Stands in for tinyobj::LoadObj with default options (Engine.cpp:250): v / vn / vt / f records,
// 1-based and negative (relative) indices, polygons triangulated as fans, faces in file order.
int mcull_mesh_load_obj(const char *path, mcull_mesh **out) {
    if (!path || !out) return MCULL_E_INVALID;
    *out = nullptr;
    FILE *f = std::fopen(path, "rb");
    if (!f) return MCULL_E_IO;
    std::vector<float> pos, nrm, uv;
    std::vector<int32_t> corners;
    std::string line;
    char buf[4096];
    auto is_blank = [](char c) { return c == ' ' || c == '\t'; };
    for (;;) {
        line.clear();
        bool got = false;
        while (std::fgets(buf, sizeof buf, f)) {
            got = true;
            line += buf;
            if (!line.empty() && line.back() == '\n') break;
        }
        if (!got) break;
        const char *p = line.c_str();
        while (is_blank(*p)) p++;
        auto floats = [&](const char *s, int n, std::vector<float> &dst) {
            char *e = const_cast<char *>(s);
            for (int k = 0; k < n; k++) dst.push_back((float)std::strtod(e, &e));
        };
        if (p[0] == 'v' && is_blank(p[1])) floats(p + 1, 3, pos);
        else if (p[0] == 'v' && p[1] == 'n' && is_blank(p[2])) floats(p + 2, 3, nrm);
        else if (p[0] == 'v' && p[1] == 't' && is_blank(p[2])) floats(p + 2, 2, uv);
        else if (p[0] == 'f' && is_blank(p[1])) {
            int32_t face[64][3];
            int nf = 0;
            char *e = const_cast<char *>(p + 1);
            const long np = (long)(pos.size() / 3), nt = (long)(uv.size() / 2), nn = (long)(nrm.size() / 3);
            while (nf < 64) {
                while (is_blank(*e)) e++;
                if (*e == '\0' || *e == '\n' || *e == '\r') break;
                long a = std::strtol(e, &e, 10), b = 0, c = 0;
                bool has_b = false, has_c = false;
                if (*e == '/') {
                    e++;
                    if (*e != '/') { b = std::strtol(e, &e, 10); has_b = true; }
                    if (*e == '/') { e++; c = std::strtol(e, &e, 10); has_c = true; }
                }
                face[nf][0] = (int32_t)(a > 0 ? a - 1 : np + a);
                face[nf][1] = has_b ? (int32_t)(b > 0 ? b - 1 : nt + b) : -1;
                face[nf][2] = has_c ? (int32_t)(c > 0 ? c - 1 : nn + c) : -1;
                nf++;
            }
            for (int k = 2; k < nf; k++)
                for (int corner : {0, k - 1, k}) corners.insert(corners.end(), face[corner], face[corner] + 3);
        }
    }
    std::fclose(f);
    return mcull_mesh_from_arrays(pos.data(), pos.size() / 3, nrm.data(), nrm.size() / 3, uv.data(), uv.size() / 2,
                                  corners.data(), corners.size() / 3, out);
}

int mcull_mesh_optimize(mcull_mesh *mesh) {
    if (!mesh) return MCULL_E_INVALID;
    optimize(mesh);
    return MCULL_OK;
}

int mcull_mesh_build_meshlets(mcull_mesh *mesh) {
    if (!mesh) return MCULL_E_INVALID;
    build_meshlets(mesh);
    return MCULL_OK;
}

int mcull_mesh_counts(const mcull_mesh *mesh, mcull_scene_counts *out) {
    if (!mesh || !out) return MCULL_E_INVALID;
    std::memset(out, 0, sizeof *out);
    out->vertex_count = mesh->vertices.size();
    out->index_count = mesh->indices.size();
    out->meshlet_data_words = mesh->meshlet_data.size();
    out->meshlet_count = mesh->meshlets.size();
    return MCULL_OK;
}

const mcull_vertex *mcull_mesh_vertices(const mcull_mesh *mesh) { return mesh ? mesh->vertices.data() : nullptr; }
const uint32_t *mcull_mesh_indices(const mcull_mesh *mesh) { return mesh ? mesh->indices.data() : nullptr; }
const uint32_t *mcull_mesh_meshlet_data(const mcull_mesh *mesh) { return mesh ? mesh->meshlet_data.data() : nullptr; }
const mcull_meshlet *mcull_mesh_meshlets(const mcull_mesh *mesh) { return mesh ? mesh->meshlets.data() : nullptr; }

int mcull_mesh_upload(mcull_ctx *ctx, mcull_mesh *mesh) {
    if (!ctx || !mesh) return MCULL_E_INVALID;
    int rc;
    if ((rc = mcull_upload_meshlets(ctx, mesh->meshlets.data(), mesh->meshlets.size(), MCULL_MEM_HOST))) return rc;
    if ((rc = mcull_upload_geometry(ctx, mesh->vertices.data(), mesh->vertices.size(), mesh->meshlet_data.data(),
                                    mesh->meshlet_data.size(), MCULL_MEM_HOST)))
        return rc;
    if (mesh->meshlets.empty()) return MCULL_OK;
    if ((rc = mcull_compute_bounds(ctx))) return rc;  // buildMeshletCon on the GPU
    return mcull_download_meshlets(ctx, mesh->meshlets.data(), mesh->meshlets.size());
}

// The same dedup loop with the per-corner work, the hash table and the first-seen numbering on the GPU.
int mcull_mesh_from_arrays_gpu(mcull_ctx *ctx, const float *positions, uint64_t n_positions, const float *normals,
                               uint64_t n_normals, const float *texcoords, uint64_t n_texcoords, const int32_t *corners,
                               uint64_t n_corners, mcull_mesh **out) {
    if (!ctx || !out) return MCULL_E_INVALID;
    *out = nullptr;
    if ((!positions && n_positions) || (!normals && n_normals) || (!texcoords && n_texcoords) || (!corners && n_corners) ||
        n_corners % 3 != 0 || n_corners > 0x7ffffff0ull)
        return mcull_ctx_fail(ctx, MCULL_E_INVALID, "mesh arrays: null pointer or corner count not a multiple of 3");
    // index ranges are checked on the host before anything is launched (the kernels index unchecked)
    for (uint64_t c = 0; c < n_corners; c++) {
        const int32_t vi = corners[3 * c], ti = corners[3 * c + 1], ni = corners[3 * c + 2];
        if (vi < 0 || (uint64_t)vi >= n_positions || (n_normals && (ni < 0 || (uint64_t)ni >= n_normals)) ||
            (n_texcoords && (ti < 0 || (uint64_t)ti >= n_texcoords)))
            return mcull_ctx_fail(ctx, MCULL_E_INVALID, "mesh arrays: a corner index lies outside its array");
    }
    mcull_mesh *m = new (std::nothrow) mcull_mesh;
    if (!m) return MCULL_E_NOMEM;
    if (hipSetDevice(mcull_ctx_device(ctx)) != hipSuccess) {
        delete m;
        return mcull_ctx_fail(ctx, MCULL_E_HIP, "hipSetDevice failed");
    }
    const hipError_t e = mcull::gpu_dedup_vertices(positions, n_positions, normals, n_normals, texcoords, n_texcoords, corners,
                                                   (uint32_t)n_corners, &m->vertices, &m->indices, mcull_ctx_stream(ctx));
    if (e != hipSuccess) {
        delete m;
        return mcull_ctx_fail(ctx, e == hipErrorOutOfMemory ? MCULL_E_NOMEM : MCULL_E_HIP, hipGetErrorString(e));
    }
    *out = m;
    return MCULL_OK;
}

// createMeshlets on the GPU followed by mcull_mesh_upload (bounds on the GPU as well).
int mcull_mesh_build_meshlets_gpu(mcull_ctx *ctx, mcull_mesh *mesh) {
    if (!ctx || !mesh) return MCULL_E_INVALID;
    if (mesh->indices.size() / 3 > 0xfffffff0ull) return mcull_ctx_fail(ctx, MCULL_E_INVALID, "too many triangles");
    for (uint32_t i : mesh->indices)
        if (i >= mesh->vertices.size()) return mcull_ctx_fail(ctx, MCULL_E_INVALID, "index outside the vertex array");
    if (hipSetDevice(mcull_ctx_device(ctx)) != hipSuccess) return mcull_ctx_fail(ctx, MCULL_E_HIP, "hipSetDevice failed");
    const hipError_t e = mcull::gpu_build_meshlets(mesh->indices.data(), (uint32_t)(mesh->indices.size() / 3), &mesh->meshlets,
                                                   &mesh->meshlet_data, mcull_ctx_stream(ctx));
    if (e != hipSuccess) return mcull_ctx_fail(ctx, e == hipErrorOutOfMemory ? MCULL_E_NOMEM : MCULL_E_HIP, hipGetErrorString(e));
    return mcull_mesh_upload(ctx, mesh);
}

int mcull_mesh_save(const mcull_mesh *mesh, const char *path) {
    if (!mesh || !path) return MCULL_E_INVALID;
    return mcull_scene_save(path, mesh->vertices.data(), mesh->vertices.size(), mesh->indices.data(), mesh->indices.size(),
                            mesh->meshlet_data.data(), mesh->meshlet_data.size(), mesh->meshlets.data(), mesh->meshlets.size(),
                            nullptr, nullptr, 0);
}

void mcull_mesh_destroy(mcull_mesh *mesh) { delete mesh; }

}  // extern "C"
