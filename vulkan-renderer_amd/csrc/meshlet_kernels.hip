// GPU front half of the load-time mesh pipeline: vertex deduplication and meshlet building, with
// results identical to the sequential host code in mesh_pipeline.hip (and therefore to the
// reference's loadModel loop, Engine.cpp:252-281, and createMeshlets, Engine.cpp:548-628).
//
// Deduplication.  The reference numbers vertices in first-seen order of the face corners.  Here every
// corner builds its Vertex record in parallel, the records go into an open-addressing hash table whose
// slots hold the LOWEST corner index of each distinct vertex (compare-and-swap to claim a slot,
// atomicMin among equal vertices), a flag marks the corners that are the first of their vertex, and an
// exclusive scan of the flags in corner order is the first-seen numbering.  Equality is
// Vertex::operator== (Engine.hpp:57-61): float compare on the position (-0 == +0, NaN equals nothing).
//
// Meshlets.  createMeshlets is a greedy scan, but the meshlet that starts at a given triangle is a
// function of that start alone.  One thread per triangle computes where a meshlet starting there
// would end (at most 124 triangles ahead, its <= 64 vertex ids kept in LDS); the real boundaries are
// the chain 0 -> end(0) -> end(end(0)) ..., followed on the host over the downloaded `end` array
// (one dependent step per meshlet); one thread per real meshlet then writes its vertex list in
// first-seen order and its packed local indices.  Cone/apex/radius come from the bounds kernel.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "mcull_internal.h"

namespace mcull {

constexpr int kMeshBlock = 256;
constexpr uint32_t kEmpty = 0xffffffffu;

// ---------------------------------------------------------------------------------------------
// corner -> Vertex record (Engine.cpp:254-273)

__device__ __forceinline__ uint8_t quantise_normal(float n) {
    const float s = (n * 0.5f + 0.5f) * 255.0f;
    if (!(s == s) || s <= -1.0f || s >= 2147483648.0f) return 0;  // NaN / outside int32: what the x86 build yields
    return (uint8_t)(int32_t)s;
}

__device__ __forceinline__ uint16_t float_to_half_bits(float f) {
    const _Float16 h = (_Float16)f;  // round to nearest even, utils.hpp:94-99
    return __builtin_bit_cast(uint16_t, h);
}

__global__ __launch_bounds__(kMeshBlock) void corner_vertex_kernel(const float *__restrict__ pos, const float *__restrict__ nrm,
                                                                   uint32_t n_nrm, const float *__restrict__ uv, uint32_t n_uv,
                                                                   const int32_t *__restrict__ corners, uint32_t n_corners,
                                                                   mcull_vertex *__restrict__ cv) {
    const uint32_t c = blockIdx.x * kMeshBlock + threadIdx.x;
    if (c >= n_corners) return;
    const int32_t vi = corners[3 * c], ti = corners[3 * c + 1], ni = corners[3 * c + 2];
    mcull_vertex v;
    v.vx = pos[3 * vi]; v.vy = pos[3 * vi + 1]; v.vz = pos[3 * vi + 2]; v.vw = 0.0f;
    float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
    if (n_nrm) { n0 = nrm[3 * ni]; n1 = nrm[3 * ni + 1]; n2 = nrm[3 * ni + 2]; }
    const float inv = 1.0f / sqrtf((n0 * n0 + n1 * n1) + n2 * n2);
    v.nx = quantise_normal(n0 * inv); v.ny = quantise_normal(n1 * inv); v.nz = quantise_normal(n2 * inv); v.nw = 0;
    v.tu = 0; v.tv = 0;
    if (n_uv) {
        v.tu = float_to_half_bits(uv[2 * ti]);
        v.tv = float_to_half_bits(1.0f - uv[2 * ti + 1]);
    }
    cv[c] = v;
}

__device__ __forceinline__ bool vertex_equal(const mcull_vertex &a, const mcull_vertex &b) {
    return a.vx == b.vx && a.vy == b.vy && a.vz == b.vz && a.nx == b.nx && a.ny == b.ny && a.nz == b.nz && a.tu == b.tu &&
           a.tv == b.tv;
}
__device__ __forceinline__ bool has_nan_position(const mcull_vertex &v) { return !(v.vx == v.vx && v.vy == v.vy && v.vz == v.vz); }
__device__ __forceinline__ uint32_t vertex_hash(const mcull_vertex &v) {
    uint32_t w[5] = {__float_as_uint(v.vx == 0.0f ? 0.0f : v.vx), __float_as_uint(v.vy == 0.0f ? 0.0f : v.vy),
                     __float_as_uint(v.vz == 0.0f ? 0.0f : v.vz), (uint32_t)v.nx | ((uint32_t)v.ny << 8) | ((uint32_t)v.nz << 16),
                     (uint32_t)v.tu | ((uint32_t)v.tv << 16)};
    uint32_t h = 0x811c9dc5u;
    for (int i = 0; i < 5; i++) {
        h = (h ^ w[i]) * 0x01000193u;
        h ^= h >> 15;
    }
    return h;
}

// slot = lowest corner index among the corners whose vertex equals the slot's vertex
__global__ __launch_bounds__(kMeshBlock) void dedup_insert_kernel(const mcull_vertex *__restrict__ cv, uint32_t n_corners,
                                                                  uint32_t *__restrict__ table, uint32_t mask) {
    const uint32_t c = blockIdx.x * kMeshBlock + threadIdx.x;
    if (c >= n_corners) return;
    const mcull_vertex v = cv[c];
    if (has_nan_position(v)) return;
    uint32_t h = vertex_hash(v) & mask;
    for (uint32_t probe = 0; probe <= mask; probe++) {  // the table holds at least 2x the corners: always terminates early
        uint32_t cur = __hip_atomic_load(table + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmpty) {
            cur = atomicCAS(table + h, kEmpty, c);
            if (cur == kEmpty) return;  // claimed
        }
        if (vertex_equal(cv[cur], v)) {
            atomicMin(table + h, c);
            return;
        }
        h = (h + 1u) & mask;
    }
}

__global__ __launch_bounds__(kMeshBlock) void dedup_lookup_kernel(const mcull_vertex *__restrict__ cv, uint32_t n_corners,
                                                                  const uint32_t *__restrict__ table, uint32_t mask,
                                                                  uint32_t *__restrict__ first, uint32_t *__restrict__ flag) {
    const uint32_t c = blockIdx.x * kMeshBlock + threadIdx.x;
    if (c >= n_corners) return;
    const mcull_vertex v = cv[c];
    uint32_t f = c;
    if (!has_nan_position(v)) {
        uint32_t h = vertex_hash(v) & mask;
        for (uint32_t probe = 0; probe <= mask; probe++) {
            const uint32_t cur = table[h];
            if (cur == kEmpty) break;  // cannot happen after the insert pass
            if (vertex_equal(cv[cur], v)) {
                f = cur;
                break;
            }
            h = (h + 1u) & mask;
        }
    }
    first[c] = f;
    flag[c] = (f == c) ? 1u : 0u;
}

// exclusive scan of 32-bit values, 1024 per workgroup: (1) per-workgroup scan + total, (2) one
// workgroup scans the totals, (3) add
__device__ __forceinline__ uint32_t wave_inclusive(uint32_t x, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(x, d, 64);
        if (lane >= (uint32_t)d) x += up;
    }
    return x;
}
__global__ __launch_bounds__(kMeshBlock) void scan_blocks_kernel(const uint32_t *__restrict__ in, uint32_t n,
                                                                 uint32_t *__restrict__ out, uint32_t *__restrict__ block_total) {
    __shared__ uint32_t s_wave[kMeshBlock / 64];
    const uint32_t base = (blockIdx.x * kMeshBlock + threadIdx.x) * 4u, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        x[k] = base + k < n ? in[base + k] : 0u;
        sum += x[k];
    }
    const uint32_t incl = wave_inclusive(sum, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
    for (uint32_t w = 0; w < kMeshBlock / 64; w++) {
        if (w < wave) wave_base += s_wave[w];
        total += s_wave[w];
    }
    uint32_t run = wave_base + incl - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + k < n) out[base + k] = run;
        run += x[k];
    }
    if (threadIdx.x == 0) block_total[blockIdx.x] = total;
}
__global__ __launch_bounds__(kMeshBlock) void scan_totals_kernel(uint32_t *__restrict__ block_total, uint32_t n_blocks,
                                                                 uint32_t *__restrict__ grand_total) {
    __shared__ uint32_t s_wave[kMeshBlock / 64];
    __shared__ uint32_t s_carry;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += kMeshBlock) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t x = i < n_blocks ? block_total[i] : 0u;
        const uint32_t incl = wave_inclusive(x, lane);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_base = 0, total = 0;
        for (uint32_t w = 0; w < kMeshBlock / 64; w++) {
            if (w < wave) wave_base += s_wave[w];
            total += s_wave[w];
        }
        const uint32_t carry = s_carry;
        if (i < n_blocks) block_total[i] = carry + wave_base + incl - x;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand_total = s_carry;
}
__global__ __launch_bounds__(kMeshBlock) void scan_add_kernel(uint32_t *__restrict__ out, uint32_t n,
                                                              const uint32_t *__restrict__ block_total) {
    const uint32_t base = (blockIdx.x * kMeshBlock + threadIdx.x) * 4u, add = block_total[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (base + k < n) out[base + k] += add;
}

__global__ __launch_bounds__(kMeshBlock) void dedup_emit_kernel(const mcull_vertex *__restrict__ cv, uint32_t n_corners,
                                                                const uint32_t *__restrict__ first,
                                                                const uint32_t *__restrict__ flag,
                                                                const uint32_t *__restrict__ rank,
                                                                uint32_t *__restrict__ indices, mcull_vertex *__restrict__ vertices) {
    const uint32_t c = blockIdx.x * kMeshBlock + threadIdx.x;
    if (c >= n_corners) return;
    indices[c] = rank[first[c]];
    if (flag[c]) vertices[rank[c]] = cv[c];
}

// ---------------------------------------------------------------------------------------------
// meshlets

// The greedy scan of Engine.cpp:560-612 started at triangle `s`: returns where it flushes.  The vertex
// set of thread t is column t of `set` (64 rows).  EMIT additionally writes the meshlet's data block.
template <bool EMIT>
__device__ __forceinline__ uint32_t greedy_meshlet(const uint32_t *__restrict__ indices, uint32_t n_tri, uint32_t s,
                                                   uint32_t (*set)[kMeshBlock], uint32_t tid, uint32_t *n_vertices_out,
                                                   uint32_t *__restrict__ data_out, uint32_t limit_tri) {
    uint32_t nv = 0, nt = 0, t = s;
    uint8_t *local_out = nullptr;
    for (; t < n_tri; t++) {
        const uint32_t g[3] = {indices[3 * t], indices[3 * t + 1], indices[3 * t + 2]};
        uint32_t slot[3];
        uint32_t fresh = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            slot[k] = kEmpty;
            for (uint32_t j = 0; j < nv; j++)
                if (set[j][tid] == g[k]) {
                    slot[k] = j;
                    break;
                }
            fresh += slot[k] == kEmpty;  // per corner: a new vertex repeated inside the triangle counts twice (Engine.cpp:571)
        }
        if (nv + fresh > 64u || nt >= 124u) break;
        if (EMIT && nt >= limit_tri) break;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (slot[k] == kEmpty) {
                // may have been added by an earlier corner of this same triangle
                for (int q = 0; q < k; q++)
                    if (g[q] == g[k]) slot[k] = slot[q];
                if (slot[k] == kEmpty) {
                    slot[k] = nv;
                    set[nv][tid] = g[k];
                    nv++;
                }
            }
        }
        if (EMIT) {
            // local indices are written after the vertex list, whose length is known to the caller
            if (!local_out) local_out = reinterpret_cast<uint8_t *>(data_out + *n_vertices_out);
            local_out[3 * nt] = (uint8_t)slot[0];
            local_out[3 * nt + 1] = (uint8_t)slot[1];
            local_out[3 * nt + 2] = (uint8_t)slot[2];
        }
        nt++;
    }
    if (EMIT) {
        for (uint32_t j = 0; j < nv; j++) data_out[j] = set[j][tid];
        if (local_out)
            for (uint32_t b = 3 * nt; b % 4u; b++) local_out[b] = 0;  // pad to a whole word (Engine.cpp:577-579)
    } else {
        *n_vertices_out = nv;
    }
    return t;
}

__global__ __launch_bounds__(kMeshBlock) void meshlet_span_kernel(const uint32_t *__restrict__ indices, uint32_t n_tri,
                                                                  uint32_t *__restrict__ end, uint8_t *__restrict__ n_vertices) {
    __shared__ uint32_t set[64][kMeshBlock];
    const uint32_t s = blockIdx.x * kMeshBlock + threadIdx.x;
    if (s >= n_tri) return;
    uint32_t nv = 0;
    end[s] = greedy_meshlet<false>(indices, n_tri, s, set, threadIdx.x, &nv, nullptr, 0u);
    n_vertices[s] = (uint8_t)nv;
}

__global__ __launch_bounds__(kMeshBlock) void meshlet_emit_kernel(const uint32_t *__restrict__ indices, uint32_t n_tri,
                                                                  const mcull_meshlet *__restrict__ meshlets,
                                                                  const uint32_t *__restrict__ first_tri, uint32_t n_meshlets,
                                                                  uint32_t *__restrict__ meshlet_data) {
    __shared__ uint32_t set[64][kMeshBlock];
    const uint32_t m = blockIdx.x * kMeshBlock + threadIdx.x;
    if (m >= n_meshlets) return;
    uint32_t nv = meshlets[m].vertexCount;
    greedy_meshlet<true>(indices, n_tri, first_tri[m], set, threadIdx.x, &nv, meshlet_data + meshlets[m].dataOffset,
                         meshlets[m].triangleCount);
}

// ---------------------------------------------------------------------------------------------
// host orchestration

namespace {
struct Scratch {  // frees whatever was allocated when it goes out of scope
    std::vector<void *> ptrs;
    ~Scratch() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    hipError_t alloc(T **out, size_t count) {
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 256));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T *>(p);
        return e;
    }
};
#define MESH_HIP(call)                          \
    do {                                        \
        const hipError_t e_ = (call);           \
        if (e_ != hipSuccess) return e_;        \
    } while (0)

hipError_t exclusive_scan(const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *total_dev, Scratch &sc, hipStream_t st) {
    const uint32_t blocks = (n + 1023u) / 1024u;
    uint32_t *block_total;
    MESH_HIP(sc.alloc(&block_total, blocks));
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(blocks), dim3(kMeshBlock), 0, st, in, n, out, block_total);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(kMeshBlock), 0, st, block_total, blocks, total_dev);
    hipLaunchKernelGGL(scan_add_kernel, dim3(blocks), dim3(kMeshBlock), 0, st, out, n, block_total);
    return hipGetLastError();
}
}  // namespace

hipError_t gpu_dedup_vertices(const float *pos, uint64_t npos, const float *nrm, uint64_t nnrm, const float *uv, uint64_t nuv,
                              const int32_t *corners, uint32_t n_corners, std::vector<mcull_vertex> *vertices,
                              std::vector<uint32_t> *indices, hipStream_t st) {
    vertices->clear();
    indices->clear();
    if (n_corners == 0) return hipSuccess;
    Scratch sc;
    float *d_pos, *d_nrm, *d_uv;
    int32_t *d_corners;
    mcull_vertex *d_cv, *d_vertices;
    uint32_t *d_table, *d_first, *d_flag, *d_rank, *d_indices, *d_total;
    MESH_HIP(sc.alloc(&d_pos, npos * 3)); MESH_HIP(sc.alloc(&d_nrm, nnrm * 3)); MESH_HIP(sc.alloc(&d_uv, nuv * 2));
    MESH_HIP(sc.alloc(&d_corners, (size_t)n_corners * 3));
    MESH_HIP(sc.alloc(&d_cv, n_corners)); MESH_HIP(sc.alloc(&d_vertices, n_corners));
    uint32_t table_size = 1024;
    while (table_size < 2ull * n_corners) table_size <<= 1;
    MESH_HIP(sc.alloc(&d_table, table_size));
    MESH_HIP(sc.alloc(&d_first, n_corners)); MESH_HIP(sc.alloc(&d_flag, n_corners)); MESH_HIP(sc.alloc(&d_rank, n_corners));
    MESH_HIP(sc.alloc(&d_indices, n_corners)); MESH_HIP(sc.alloc(&d_total, 1));
    MESH_HIP(hipMemcpyAsync(d_pos, pos, npos * 12, hipMemcpyHostToDevice, st));
    if (nnrm) MESH_HIP(hipMemcpyAsync(d_nrm, nrm, nnrm * 12, hipMemcpyHostToDevice, st));
    if (nuv) MESH_HIP(hipMemcpyAsync(d_uv, uv, nuv * 8, hipMemcpyHostToDevice, st));
    MESH_HIP(hipMemcpyAsync(d_corners, corners, (size_t)n_corners * 12, hipMemcpyHostToDevice, st));
    MESH_HIP(hipMemsetAsync(d_table, 0xff, (size_t)table_size * 4, st));
    const dim3 grid((n_corners + kMeshBlock - 1) / kMeshBlock), block(kMeshBlock);
    hipLaunchKernelGGL(corner_vertex_kernel, grid, block, 0, st, d_pos, d_nrm, (uint32_t)nnrm, d_uv, (uint32_t)nuv, d_corners,
                       n_corners, d_cv);
    hipLaunchKernelGGL(dedup_insert_kernel, grid, block, 0, st, d_cv, n_corners, d_table, table_size - 1u);
    hipLaunchKernelGGL(dedup_lookup_kernel, grid, block, 0, st, d_cv, n_corners, d_table, table_size - 1u, d_first, d_flag);
    MESH_HIP(hipGetLastError());
    MESH_HIP(exclusive_scan(d_flag, n_corners, d_rank, d_total, sc, st));
    hipLaunchKernelGGL(dedup_emit_kernel, grid, block, 0, st, d_cv, n_corners, d_first, d_flag, d_rank, d_indices, d_vertices);
    MESH_HIP(hipGetLastError());
    uint32_t n_vertices = 0;
    MESH_HIP(hipMemcpyAsync(&n_vertices, d_total, 4, hipMemcpyDeviceToHost, st));
    MESH_HIP(hipStreamSynchronize(st));
    vertices->resize(n_vertices);
    indices->resize(n_corners);
    MESH_HIP(hipMemcpyAsync(vertices->data(), d_vertices, (size_t)n_vertices * sizeof(mcull_vertex), hipMemcpyDeviceToHost, st));
    MESH_HIP(hipMemcpyAsync(indices->data(), d_indices, (size_t)n_corners * 4, hipMemcpyDeviceToHost, st));
    return hipStreamSynchronize(st);
}

hipError_t gpu_build_meshlets(const uint32_t *indices, uint32_t n_tri, std::vector<mcull_meshlet> *meshlets,
                              std::vector<uint32_t> *meshlet_data, hipStream_t st) {
    meshlets->clear();
    meshlet_data->clear();
    if (n_tri == 0) return hipSuccess;
    Scratch sc;
    uint32_t *d_indices, *d_end, *d_first, *d_data;
    uint8_t *d_nv;
    mcull_meshlet *d_meshlets;
    MESH_HIP(sc.alloc(&d_indices, (size_t)n_tri * 3)); MESH_HIP(sc.alloc(&d_end, n_tri)); MESH_HIP(sc.alloc(&d_nv, n_tri));
    MESH_HIP(hipMemcpyAsync(d_indices, indices, (size_t)n_tri * 12, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(meshlet_span_kernel, dim3((n_tri + kMeshBlock - 1) / kMeshBlock), dim3(kMeshBlock), 0, st, d_indices, n_tri,
                       d_end, d_nv);
    MESH_HIP(hipGetLastError());
    std::vector<uint32_t> end(n_tri), first;
    std::vector<uint8_t> nv(n_tri);
    MESH_HIP(hipMemcpyAsync(end.data(), d_end, (size_t)n_tri * 4, hipMemcpyDeviceToHost, st));
    MESH_HIP(hipMemcpyAsync(nv.data(), d_nv, n_tri, hipMemcpyDeviceToHost, st));
    MESH_HIP(hipStreamSynchronize(st));
    // the real boundaries: 0 -> end(0) -> end(end(0)) ...
    uint64_t words = 0;
    for (uint32_t s = 0; s < n_tri;) {
        const uint32_t e = end[s];
        if (e <= s || e > n_tri) return hipErrorUnknown;  // a meshlet always takes at least one triangle
        mcull_meshlet m;
        std::memset(&m, 0, sizeof m);
        m.dataOffset = (uint32_t)words;
        m.triangleCount = (uint8_t)(e - s);
        m.vertexCount = nv[s];
        words += m.vertexCount + ((uint32_t)m.triangleCount * 3u + 3u) / 4u;
        if (words > 0xffffffffull) return hipErrorUnknown;
        meshlets->push_back(m);
        first.push_back(s);
        s = e;
    }
    const uint32_t n_meshlets = (uint32_t)meshlets->size();
    MESH_HIP(sc.alloc(&d_first, n_meshlets)); MESH_HIP(sc.alloc(&d_meshlets, n_meshlets)); MESH_HIP(sc.alloc(&d_data, words));
    MESH_HIP(hipMemcpyAsync(d_first, first.data(), (size_t)n_meshlets * 4, hipMemcpyHostToDevice, st));
    MESH_HIP(hipMemcpyAsync(d_meshlets, meshlets->data(), (size_t)n_meshlets * sizeof(mcull_meshlet), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(meshlet_emit_kernel, dim3((n_meshlets + kMeshBlock - 1) / kMeshBlock), dim3(kMeshBlock), 0, st, d_indices,
                       n_tri, d_meshlets, d_first, n_meshlets, d_data);
    MESH_HIP(hipGetLastError());
    meshlet_data->resize(words);
    MESH_HIP(hipMemcpyAsync(meshlet_data->data(), d_data, words * 4, hipMemcpyDeviceToHost, st));
    return hipStreamSynchronize(st);
}

}  // namespace mcull
