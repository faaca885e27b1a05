// Per-meshlet bounding sphere + normal cone on the GPU, and the AoS -> SoA split.
//
// Reference: Engine::buildMeshletCon (Engine.cpp:472-547), CPU-only there.  One wavefront per
// meshlet: lanes own vertices (<= 64) and triangles (<= 124, two per lane).  Everything that is
// order-independent (min, max) is reduced across lanes; the two floating-point SUMS (normal
// average, centroid) are accumulated by one lane per component in the reference's element order,
// so cone, apex and radius come out bit-identical to the scalar evaluation (no fused
// multiply-add: this file is compiled with -ffp-contract=off).
#include "mcull_internal.h"

namespace mcull {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ float dot3b(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;
}

// {cone.xyz, coneW} and {coneApex.xyz, radius} streams from the reference's 48-byte records
// (the fields shader.task:91-99 reads).
__global__ __launch_bounds__(kBlock) void split_meshlets_kernel(const mcull_meshlet *__restrict__ aos, uint64_t n,
                                                                float4 *__restrict__ cone_axis_w,
                                                                float4 *__restrict__ centre_radius) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 *rec = reinterpret_cast<const float4 *>(aos + i);  // 48-byte records, 16-byte aligned
    const float4 cone = rec[0], apex = rec[1], tail = rec[2];
    cone_axis_w[i] = cone;
    centre_radius[i] = make_float4(apex.x, apex.y, apex.z, tail.x);
}

__global__ __launch_bounds__(kBlock) void bounds_kernel(const mcull_vertex *__restrict__ vertices,
                                                        const uint32_t *__restrict__ meshlet_data,
                                                        mcull_meshlet *__restrict__ meshlets, uint64_t n,
                                                        float4 *__restrict__ cone_axis_w,
                                                        float4 *__restrict__ centre_radius) {
    __shared__ float s_pos[kWaves][64][3];
    __shared__ float s_nrm[kWaves][124][3];
    __shared__ float s_div[kWaves][64][3];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t mi = (uint64_t)blockIdx.x * kWaves + wave;
    if (mi >= n) return;  // whole wave exits together; no workgroup barrier is used below

    mcull_meshlet *ml = meshlets + mi;
    const uint32_t data_offset = ml->dataOffset;
    const uint32_t tri_count = ml->triangleCount;
    const uint32_t vert_count = ml->vertexCount;
    const uint32_t *globals = meshlet_data + data_offset;
    const uint8_t *locals = reinterpret_cast<const uint8_t *>(globals + vert_count);

    // vertex positions (Engine.cpp:480-487)
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    if (lane < vert_count) {
        const mcull_vertex *v = vertices + globals[lane];
        px = v->vx; py = v->vy; pz = v->vz;
        s_pos[wave][lane][0] = px; s_pos[wave][lane][1] = py; s_pos[wave][lane][2] = pz;
        // centroid terms: each coordinate divided by the vertex count first (Engine.cpp:532-536)
        const float fc = (float)vert_count;
        s_div[wave][lane][0] = px / fc; s_div[wave][lane][1] = py / fc; s_div[wave][lane][2] = pz / fc;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // triangle normals (Engine.cpp:475-497)
    for (uint32_t t = lane; t < tri_count; t += 64) {
        const uint32_t i0 = locals[3 * t + 0], i1 = locals[3 * t + 1], i2 = locals[3 * t + 2];
        const float p0x = s_pos[wave][i0][0], p0y = s_pos[wave][i0][1], p0z = s_pos[wave][i0][2];
        const float ax = s_pos[wave][i1][0] - p0x, ay = s_pos[wave][i1][1] - p0y, az = s_pos[wave][i1][2] - p0z;
        const float bx = s_pos[wave][i2][0] - p0x, by = s_pos[wave][i2][1] - p0y, bz = s_pos[wave][i2][2] - p0z;
        // glm::cross(a, b) = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
        const float cx = ay * bz - by * az, cy = az * bx - bz * ax, cz = ax * by - bx * ay;
        const float inv = 1.0f / sqrtf(dot3b(cx, cy, cz, cx, cy, cz));
        s_nrm[wave][t][0] = cx * inv; s_nrm[wave][t][1] = cy * inv; s_nrm[wave][t][2] = cz * inv;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // lanes 0..2: serial sums in element order, one component each (Engine.cpp:500-503, :532-536)
    float sum_n = 0.0f, sum_p = 0.0f;
    if (lane < 3) {
        for (uint32_t t = 0; t < tri_count; t++) sum_n += s_nrm[wave][t][lane];
        for (uint32_t i = 0; i < vert_count; i++) sum_p += s_div[wave][i][lane];
    }
    const float nx = __shfl(sum_n, 0, 64), ny = __shfl(sum_n, 1, 64), nz = __shfl(sum_n, 2, 64);
    const float cx = __shfl(sum_p, 0, 64), cy = __shfl(sum_p, 1, 64), cz = __shfl(sum_p, 2, 64);
    const float inv_n = 1.0f / sqrtf(dot3b(nx, ny, nz, nx, ny, nz));  // glm::normalize, Engine.cpp:504
    const float ax = nx * inv_n, ay = ny * inv_n, az = nz * inv_n;

    // minimum of dot(normal, axis) starting from 1.0; std::min(h, dp) ignores a NaN dp (Engine.cpp:509-513)
    float h = 1.0f;
    for (uint32_t t = lane; t < tri_count; t += 64) {
        const float dp = (s_nrm[wave][t][0] * ax + s_nrm[wave][t][1] * ay) + s_nrm[wave][t][2] * az;
        h = (dp < h) ? dp : h;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float o = __shfl_xor(h, d, 64);
        h = (o < h) ? o : h;
    }
    const float cone_w = h < 0.0f ? 1.0f : sqrtf(1 - h * h);  // Engine.cpp:526

    // radius = sqrt(max squared distance to the centroid) (Engine.cpp:539-546)
    float d2 = 0.0f;
    if (lane < vert_count) {
        const float dx = px - cx, dy = py - cy, dz = pz - cz;
        d2 = (dx * dx + dy * dy) + dz * dz;
        if (!(d2 == d2)) d2 = 0.0f;  // std::max(m, NaN) keeps m
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float o = __shfl_xor(d2, d, 64);
        d2 = (d2 < o) ? o : d2;
    }
    const float radius = sqrtf(d2);

    if (lane == 0) {
        float4 *rec = reinterpret_cast<float4 *>(ml);
        rec[0] = make_float4(ax, ay, az, cone_w);
        rec[1] = make_float4(cx, cy, cz, 0.0f);
        ml->radius = radius;
        cone_axis_w[mi] = make_float4(ax, ay, az, cone_w);
        centre_radius[mi] = make_float4(cx, cy, cz, radius);
    }
}

hipError_t launch_split_meshlets(const mcull_meshlet *aos, uint64_t n, float4 *cone_axis_w, float4 *centre_radius,
                                 hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(split_meshlets_kernel, dim3((uint32_t)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, aos, n,
                       cone_axis_w, centre_radius);
    return hipGetLastError();
}

hipError_t launch_bounds(const mcull_vertex *vertices, const uint32_t *meshlet_data, mcull_meshlet *meshlets,
                         uint64_t n, float4 *cone_axis_w, float4 *centre_radius, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bounds_kernel, dim3((uint32_t)((n + kWaves - 1) / kWaves)), dim3(kBlock), 0, st, vertices,
                       meshlet_data, meshlets, n, cone_axis_w, centre_radius);
    return hipGetLastError();
}

}  // namespace mcull
