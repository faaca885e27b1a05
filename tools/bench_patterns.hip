// Design-evidence benchmark for the cull + compact kernels (not part of the product): the product
// kernels against variants that make the other choice (AoS records, LDS-staged transforms, plain
// loads/stores, atomic compaction) on the 100 M meshlet-instance scene.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I. tools/bench_patterns.hip -o tools/bench_patterns
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../vulkan-renderer_amd/csrc/cull_kernels.hip"

using namespace mcull;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// Variant: bounds read from the reference's 48-byte AoS records instead of the two SoA streams.
template <uint32_t TESTS, int NG>
__global__ __launch_bounds__(kBlock) void cull_aos_kernel(const mcull_meshlet *__restrict__ aos,
                                                          const float4 *__restrict__ transforms,
                                                          const float *__restrict__ xf_scale,
                                                          const uint4 *__restrict__ group_rec, uint32_t n_groups, ViewConst v,
                                                          uint64_t *__restrict__ mask) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g0 = blockIdx.x * (kWaves * NG) + wave * NG;
    GroupInfo gi[NG]; Xform m[NG]; float scale[NG]; float4 a[NG], c[NG];
#pragma unroll
    for (int k = 0; k < NG; k++) {
        gi[k] = group_lookup(min(g0 + k, n_groups - 1u), group_rec, 0u, 0u);
        if (g0 + k >= n_groups) gi[k].count = 0;
        m[k] = load_xform(transforms, gi[k].inst);
        scale[k] = xf_scale[gi[k].inst];
        const float4 *rec = reinterpret_cast<const float4 *>(aos + gi[k].first_meshlet + lane);
        const float4 cone = rec[0], apex = rec[1], tail = rec[2];
        a[k] = cone; c[k] = make_float4(apex.x, apex.y, apex.z, tail.x);
    }
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const bool vis = pair_visible<TESTS, true>(v, m[k], scale[k], a[k], c[k]) && (lane < gi[k].count);
        const uint64_t ballot = __ballot(vis);
        if (lane == 0 && gi[k].count) mask[g0 + k] = ballot;
    }
}

// Variant: instance transforms fetched per wave with wave-uniform scalar loads into SGPRs (the
// product stages them in LDS with one coalesced load per workgroup).
template <uint32_t TESTS, int NG, bool NT>
__global__ __launch_bounds__(kBlock) void cull_sgpr_xform_kernel(const float4 *__restrict__ cone_axis_w,
                                                                 const float4 *__restrict__ centre_radius,
                                                                 const float4 *__restrict__ transforms,
                                                                 const float *__restrict__ xf_scale,
                                                                 const uint4 *__restrict__ group_rec, uint32_t n_groups,
                                                                 ViewConst v, uint64_t *__restrict__ mask) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g0 = blockIdx.x * (kWaves * NG) + wave * NG;
    GroupInfo gi[NG]; Xform m[NG]; float scale[NG]; float4 a[NG], c[NG];
#pragma unroll
    for (int k = 0; k < NG; k++) {
        gi[k] = group_lookup(min(g0 + k, n_groups - 1u), group_rec, 0u, 0u);
        if (g0 + k >= n_groups) gi[k].count = 0;
        m[k] = load_xform(transforms, gi[k].inst);
        scale[k] = xf_scale[gi[k].inst];
        const uint32_t last = gi[k].count ? gi[k].count - 1u : 0u;
        const uint32_t id = gi[k].first_meshlet + (lane < last ? lane : last);
        a[k] = load_bounds<NT>(cone_axis_w + id);
        c[k] = load_bounds<NT>(centre_radius + id);
    }
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const bool vis = pair_visible<TESTS, true>(v, m[k], scale[k], a[k], c[k]) && (lane < gi[k].count);
        const uint64_t ballot = __ballot(vis);
        if (lane == 0 && gi[k].count) mask[g0 + k] = ballot;
    }
}

// Probe for a fused single-pass design: the cull pass that also writes the visible entries, with the
// list offset of every group given (here precomputed from an earlier run's mask; a real single pass
// would obtain it by a decoupled look-back over the workgroups).  Measures what fusing could gain.
template <uint32_t TESTS, int NG>
__global__ __launch_bounds__(kBlock) void cull_fused_probe(const float4 *__restrict__ cone_axis_w,
                                                           const float4 *__restrict__ centre_radius,
                                                           const float4 *__restrict__ transforms, const float *__restrict__ xf_scale,
                                                           const uint4 *__restrict__ group_rec, uint32_t n_groups, ViewConst v,
                                                           const uint32_t *__restrict__ group_off, mcull_visible *__restrict__ list) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g0 = (blockIdx.x * kWaves + wave) * NG;
    if (g0 >= n_groups) return;
    GroupInfo gi[NG];
    Xform m[NG];
    float scale[NG];
    float4 a[NG], c[NG];
    uint32_t off[NG];
#pragma unroll
    for (int k = 0; k < NG; k++) {
        gi[k] = group_lookup(min(g0 + k, n_groups - 1u), group_rec, 0u, 0u);
        if (g0 + k >= n_groups) gi[k].count = 0;
        m[k] = load_xform(transforms, gi[k].inst);
        scale[k] = xf_scale[gi[k].inst];
        off[k] = group_off[min(g0 + k, n_groups - 1u)];
        const uint32_t last = gi[k].count ? gi[k].count - 1u : 0u;
        const uint32_t id = gi[k].first_meshlet + (lane < last ? lane : last);
        a[k] = load_bounds<true>(cone_axis_w + id);
        c[k] = load_bounds<true>(centre_radius + id);
    }
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const bool vis = pair_visible<TESTS, true>(v, m[k], scale[k], a[k], c[k]) && (lane < gi[k].count);
        const uint64_t ballot = __ballot(vis);
        if (vis) store_entry<true>(list, (unsigned long long)off[k] + __popcll(ballot & ((1ull << lane) - 1ull)), gi[k].inst,
                                   gi[k].first_meshlet + lane);
    }
}

template <bool STRIDED, int UNROLL, bool NT>
__global__ __launch_bounds__(kBlock) void read_floor_kernel(const float4 *__restrict__ cone_axis_w,
                                                            const float4 *__restrict__ centre_radius, uint32_t n_groups,
                                                            uint64_t *__restrict__ mask) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t g, g_end, step;
    if (STRIDED) { g = blockIdx.x * kWaves + wave; g_end = n_groups; step = gridDim.x * kWaves; }
    else { g = blockIdx.x * kTileGroups + wave; g_end = min(blockIdx.x * kTileGroups + kTileGroups, n_groups); step = kWaves; }
    for (; g < g_end; g += step * UNROLL) {
        float4 a[UNROLL], c[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t gg = min(g + u * step, g_end - 1);
            a[u] = load_bounds<NT>(&cone_axis_w[(size_t)gg * 64 + lane]);
            c[u] = load_bounds<NT>(&centre_radius[(size_t)gg * 64 + lane]);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint64_t ballot = __ballot(a[u].w + c[u].w > 0.6f);
            if (lane == 0 && g + u * step < g_end) mask[g + u * step] = ballot;
        }
    }
}

// stand-in for the compaction pass's output traffic: n8 8-byte entries
template <bool NT>
__global__ __launch_bounds__(256) void write_kernel(unsigned long long *out, size_t n8) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n8; i += stride) {
        if (NT) __builtin_nontemporal_store((unsigned long long)i, out + i);
        else out[i] = i;
    }
}

__global__ void fill_aos_kernel(const float4 *cone, const float4 *sphere, mcull_meshlet *aos, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 a = cone[i], c = sphere[i];
    aos[i].cone[0] = a.x; aos[i].cone[1] = a.y; aos[i].cone[2] = a.z; aos[i].cone[3] = a.w;
    aos[i].coneApex[0] = c.x; aos[i].coneApex[1] = c.y; aos[i].coneApex[2] = c.z; aos[i].radius = c.w;
}

int main(int argc, char **argv) {
    const uint32_t n_inst = 781250, per = 128;
    const uint64_t n = (uint64_t)n_inst * per;
    const uint32_t n_groups = n_inst * 2;
    float4 *cone, *sphere, *xf; uint4 *rec; uint64_t *mask; uint32_t *tc;
    CK(hipMalloc(&cone, n * 16)); CK(hipMalloc(&sphere, n * 16)); CK(hipMalloc(&xf, (size_t)n_inst * 64));
    CK(hipMalloc(&rec, (size_t)n_groups * 16)); CK(hipMalloc(&mask, (size_t)n_groups * 8)); CK(hipMalloc(&tc, 1 << 20));
    float *scale; CK(hipMalloc(&scale, (size_t)n_inst * 4));
    {
        std::vector<float4> h(1 << 22), h2(1 << 22);
        srand(1);
        auto U = [] { return rand() / (float)RAND_MAX; };
        for (auto &q : h) {  // unit axis + coneW
            float x = U() - 0.5f, y = U() - 0.5f, z = U() - 0.5f, l = sqrtf(x * x + y * y + z * z) + 1e-9f;
            q = make_float4(x / l, y / l, z / l, U() < 0.1f ? 1.0f : U());
        }
        for (auto &q : h2) q = make_float4(U(), U(), U(), 0.01f + 0.04f * U());
        for (uint64_t o = 0; o < n; o += h.size()) {
            uint64_t k = std::min<uint64_t>(h.size(), n - o);
            CK(hipMemcpy(cone + o, h.data(), k * 16, hipMemcpyHostToDevice));
            CK(hipMemcpy(sphere + o, h2.data(), k * 16, hipMemcpyHostToDevice));
        }
        std::vector<float> t((size_t)n_inst * 16, 0.f), sc(n_inst, 1.0f);
        for (uint32_t i = 0; i < n_inst; i++) {
            float qw = U() - 0.5f, qx = U() - 0.5f, qy = U() - 0.5f, qz = U() - 0.5f, l = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
            qw /= l; qx /= l; qy /= l; qz /= l;
            float *m = &t[16 * i];
            m[0] = 1 - 2 * (qy * qy + qz * qz); m[4] = 2 * (qx * qy - qz * qw); m[8] = 2 * (qx * qz + qy * qw);
            m[1] = 2 * (qx * qy + qz * qw); m[5] = 1 - 2 * (qx * qx + qz * qz); m[9] = 2 * (qy * qz - qx * qw);
            m[2] = 2 * (qx * qz - qy * qw); m[6] = 2 * (qy * qz + qx * qw); m[10] = 1 - 2 * (qx * qx + qy * qy);
            for (int k = 0; k < 3; k++) m[12 + k] = (U() - 0.5f) * 100.f;
            m[15] = 1.f;
            sc[i] = sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        }
        CK(hipMemcpy(xf, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(scale, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint4> r(n_groups);
        for (uint32_t g = 0; g < n_groups; g++) r[g] = make_uint4(g / 2, (g / 2) * per + (g & 1) * 64, 64, g & 1);
        CK(hipMemcpy(rec, r.data(), r.size() * 16, hipMemcpyHostToDevice));
    }
    ViewConst v{};
    {   // the benchmark camera: eye (51, 54.4, 40.8) looking at the origin, fov 45, aspect 16:9
        const float ex = 51.f, ey = 54.4f, ez = 40.8f, el = sqrtf(ex * ex + ey * ey + ez * ez);
        const float f[3] = {-ex / el, -ey / el, -ez / el};
        float sx = f[1], sy = -f[0], sl = sqrtf(sx * sx + sy * sy); sx /= sl; sy /= sl;      // s = f x (0,0,1)
        const float s3[3] = {sx, sy, 0.f};
        const float u[3] = {s3[1] * f[2] - s3[2] * f[1], s3[2] * f[0] - s3[0] * f[2], s3[0] * f[1] - s3[1] * f[0]};
        const float th = tanf(0.3926991f), ta = th * 16.f / 9.f;
        auto plane = [&](int i, float nx, float ny, float nz) {
            const float l = sqrtf(nx * nx + ny * ny + nz * nz); nx /= l; ny /= l; nz /= l;
            v.frustum[i][0] = nx; v.frustum[i][1] = ny; v.frustum[i][2] = nz; v.frustum[i][3] = -(nx * ex + ny * ey + nz * ez);
        };
        plane(0, f[0] * ta + s3[0], f[1] * ta + s3[1], f[2] * ta + s3[2]);
        plane(1, f[0] * ta - s3[0], f[1] * ta - s3[1], f[2] * ta - s3[2]);
        plane(2, f[0] * th + u[0], f[1] * th + u[1], f[2] * th + u[2]);
        plane(3, f[0] * th - u[0], f[1] * th - u[1], f[2] * th - u[2]);
        plane(4, f[0], f[1], f[2]);
        v.cam[0] = ex; v.cam[1] = ey; v.cam[2] = ez; v.tests = 3;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto &&launch) {
        for (int i = 0; i < 3; i++) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 10; r++) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
        }
        CK(hipGetLastError());
        printf("%-44s avg %.1f us  best %.1f us  (%.2f TB/s of bounds)\n", name, sum * 100, best * 1000, n * 32 / (sum / 10 * 1e-3) / 1e12);
    };
    const uint32_t tiles = (n_groups + 255) / 256;
    (void)tiles;
    CK(hipMemset(tc, 0, 1 << 20));
    printf("-- cull pass (mask only): group assignment, streaming hint, requests in flight\n");
#define PRODUCT(NG, NT) timeit("product cull_mask_kernel<3," #NG "," #NT ">", [&] { hipLaunchKernelGGL((cull_mask_kernel<3, NG, NT>), dim3((n_groups + 4 * NG - 1) / (4 * NG)), dim3(256), 0, 0, cone, sphere, xf, scale, rec, 0u, n_groups, v, mask, tc, 0u, (const ViewConst *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr); });
    PRODUCT(1, false) PRODUCT(1, true) PRODUCT(2, false) PRODUCT(2, true) PRODUCT(4, false) PRODUCT(4, true) PRODUCT(8, true)
    {
        std::vector<uint64_t> hm(n_groups); CK(hipMemcpy(hm.data(), mask, (size_t)n_groups * 8, hipMemcpyDeviceToHost));
        uint64_t vis = 0; for (auto w : hm) vis += __builtin_popcountll(w);
        printf("visible fraction %.3f\n", (double)vis / n);
    }
    printf("-- workgroup -> XCD mapping: round-robin (default) vs one contiguous eighth of the scene per XCD\n");
    PRODUCT(4, true)
    timeit("product cull_mask_kernel<3,4,true>, XCD-contiguous", [&] { hipLaunchKernelGGL((cull_mask_kernel<3, 4, true>), dim3((((n_groups + 15) / 16) + 7) / 8 * 8), dim3(256), 0, 0, cone, sphere, xf, scale, rec, 0u, n_groups, v, mask, tc, 1u, (const ViewConst *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr); });
    PRODUCT(4, true)
    printf("-- layout: the reference's 48-byte AoS records vs two SoA float4 streams\n");
    {
        mcull_meshlet *aos; CK(hipMalloc(&aos, n * sizeof(mcull_meshlet)));
        CK(hipMemset(aos, 0, n * sizeof(mcull_meshlet)));
        // fill the AoS records from the SoA streams on the device
        hipLaunchKernelGGL(fill_aos_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, cone, sphere, aos, n);
        CK(hipDeviceSynchronize());
        timeit("cull_aos_kernel<3,4> (48-byte AoS records)", [&] { hipLaunchKernelGGL((cull_aos_kernel<3, 4>), dim3((n_groups + 15) / 16), dim3(256), 0, 0, aos, xf, scale, rec, n_groups, v, mask); });
        CK(hipFree(aos));
    }
    printf("-- instance transforms: staged in LDS (product) vs scalar loads into SGPRs\n");
    timeit("cull_sgpr_xform_kernel<3,4,true> (SGPR transforms)", [&] { hipLaunchKernelGGL((cull_sgpr_xform_kernel<3, 4, true>), dim3((n_groups + 15) / 16), dim3(256), 0, 0, cone, sphere, xf, scale, rec, n_groups, v, mask); });
    PRODUCT(4, true)
    printf("-- read floor of the two bounds streams (trivial arithmetic)\n");
    timeit("read floor, plain loads", [&] { hipLaunchKernelGGL((read_floor_kernel<false, 4, false>), dim3(tiles), dim3(256), 0, 0, cone, sphere, n_groups, mask); });
    timeit("read floor, non-temporal loads", [&] { hipLaunchKernelGGL((read_floor_kernel<false, 4, true>), dim3(tiles), dim3(256), 0, 0, cone, sphere, n_groups, mask); });
    printf("-- compaction: ordered (mask + tile counters + scatter) vs one atomic per wavefront / per workgroup\n");
    {
        mcull_visible *list; const uint64_t cap = n; CK(hipMalloc(&list, cap * 8));
        uint32_t *inst_first, *inst_count; unsigned long long *total;
        CK(hipMalloc(&inst_first, (size_t)(n_inst + 1) * 4)); CK(hipMalloc(&inst_count, (size_t)(n_inst + 1) * 4)); CK(hipMalloc(&total, 256));
        CK(hipMemset(tc, 0, 1 << 20));
        auto zero_tc = [&] { hipLaunchKernelGGL(emit_commands_kernel, dim3((tiles + 255) / 256), dim3(256), 0, 0, (const uint2 *)nullptr, (const uint32_t *)nullptr, 0u, 0u, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint64_t *)nullptr, (const uint64_t *)nullptr, 0u, 8u, 0u, 0u, (uint8_t *)nullptr, tc, tiles); };
        timeit("ordered: cull_mask<3,4,nt> + compact<nt> + counter reset", [&] {
            hipLaunchKernelGGL((cull_mask_kernel<3, 4, true>), dim3((n_groups + 15) / 16), dim3(256), 0, 0, cone, sphere, xf, scale, rec, 0u, n_groups, v, mask, tc, 0u, (const ViewConst *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr);
            hipLaunchKernelGGL(compact_kernel<true>, dim3(tiles), dim3(256), 0, 0, mask, tc, (const unsigned long long *)nullptr, rec, 0u, n_groups, list, cap, inst_first, total, 1u);
            zero_tc();
        });
        {   // the same three launches, 20 steps back to back without a host synchronisation in between
            auto one = [&] {
                hipLaunchKernelGGL((cull_mask_kernel<3, 4, true>), dim3((n_groups + 15) / 16), dim3(256), 0, 0, cone, sphere, xf, scale, rec, 0u, n_groups, v, mask, tc, 0u, (const ViewConst *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr);
                hipLaunchKernelGGL(compact_kernel<true>, dim3(tiles), dim3(256), 0, 0, mask, tc, (const unsigned long long *)nullptr, rec, 0u, n_groups, list, cap, inst_first, total, 1u);
                zero_tc();
            };
            for (int rep = 0; rep < 3; rep++) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < 20; i++) one();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("ordered, 20 steps back to back: %.1f us per step\n", ms * 50);
            }
        }
        // the scatter pass alone on the counters of one cull pass: workgroups per 256-group tile
        hipLaunchKernelGGL((cull_mask_kernel<3, 4, true>), dim3((n_groups + 15) / 16), dim3(256), 0, 0, cone, sphere, xf, scale, rec, 0u, n_groups, v, mask, tc, 0u, (const ViewConst *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr);
        timeit("compact<nt> alone, 1 workgroup per tile", [&] { hipLaunchKernelGGL(compact_kernel<true>, dim3(tiles), dim3(256), 0, 0, mask, tc, (const unsigned long long *)nullptr, rec, 0u, n_groups, list, cap, inst_first, total, 1u); });
        timeit("compact<nt> alone, 2 workgroups per tile", [&] { hipLaunchKernelGGL(compact_kernel<true>, dim3(tiles * 2), dim3(256), 0, 0, mask, tc, (const unsigned long long *)nullptr, rec, 0u, n_groups, list, cap, inst_first, total, 2u); });
        timeit("compact<nt> alone, 4 workgroups per tile", [&] { hipLaunchKernelGGL(compact_kernel<true>, dim3(tiles * 4), dim3(256), 0, 0, mask, tc, (const unsigned long long *)nullptr, rec, 0u, n_groups, list, cap, inst_first, total, 4u); });
        zero_tc();
        timeit("single pass: one atomic per wavefront", [&] {
            CK(hipMemsetAsync(total, 0, 8)); CK(hipMemsetAsync(inst_count, 0, (size_t)(n_inst + 1) * 4));
            hipLaunchKernelGGL((cull_atomic_kernel<3, false>), dim3(tiles), dim3(256), 0, 0, cone, sphere, xf, scale, rec, 0u, n_groups, v, list, cap, inst_count, total, (uint64_t *)nullptr);
        });
        timeit("single pass: one atomic per workgroup step", [&] {
            CK(hipMemsetAsync(total, 0, 8)); CK(hipMemsetAsync(inst_count, 0, (size_t)(n_inst + 1) * 4));
            hipLaunchKernelGGL((cull_atomic_kernel<3, true>), dim3(tiles), dim3(256), 0, 0, cone, sphere, xf, scale, rec, 0u, n_groups, v, list, cap, inst_count, total, (uint64_t *)nullptr);
        });
        // what a fused single pass could reach: per-group list offsets given, entries written by the cull pass itself
        {
            hipLaunchKernelGGL((cull_mask_kernel<3, 4, true>), dim3((n_groups + 15) / 16), dim3(256), 0, 0, cone, sphere, xf, scale, rec, 0u, n_groups, v, mask, tc, 0u, (const ViewConst *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr);
            std::vector<uint64_t> hm(n_groups); CK(hipMemcpy(hm.data(), mask, (size_t)n_groups * 8, hipMemcpyDeviceToHost));
            std::vector<uint32_t> ho(n_groups); uint32_t run = 0;
            for (uint32_t g = 0; g < n_groups; g++) { ho[g] = run; run += (uint32_t)__builtin_popcountll(hm[g]); }
            uint32_t *goff; CK(hipMalloc(&goff, (size_t)n_groups * 4)); CK(hipMemcpy(goff, ho.data(), (size_t)n_groups * 4, hipMemcpyHostToDevice));
            timeit("fused probe: cull pass writing the list, offsets given", [&] { hipLaunchKernelGGL((cull_fused_probe<3, 4>), dim3((n_groups + 15) / 16), dim3(256), 0, 0, cone, sphere, xf, scale, rec, n_groups, v, goff, list); });
            CK(hipFree(goff));
            zero_tc();
        }
    }
    printf("-- list stores: plain vs non-temporal, effect on the following cull pass\n");
    {
        unsigned long long *wbuf; const size_t n8 = 51400000; CK(hipMalloc(&wbuf, n8 * 8));
        hipEvent_t ev[64]; for (auto &e : ev) CK(hipEventCreate(&e));
        for (int nt = 0; nt < 2; nt++) {
            for (int rep = 0; rep < 2; rep++) {
                for (int i = 0; i < 10; i++) {
                    CK(hipEventRecord(ev[3 * i]));
                    hipLaunchKernelGGL((cull_mask_kernel<3, 4, true>), dim3((n_groups + 15) / 16), dim3(256), 0, 0, cone, sphere, xf, scale, rec, 0u, n_groups, v, mask, tc, 0u, (const ViewConst *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr);
                    CK(hipEventRecord(ev[3 * i + 1]));
                    if (nt) hipLaunchKernelGGL(write_kernel<true>, dim3(2048), dim3(256), 0, 0, wbuf, n8);
                    else hipLaunchKernelGGL(write_kernel<false>, dim3(2048), dim3(256), 0, 0, wbuf, n8);
                    CK(hipEventRecord(ev[3 * i + 2]));
                }
                CK(hipEventRecord(ev[30]));
                CK(hipDeviceSynchronize());
            }
            float k1 = 0, wr = 0, tot = 0, ms;
            for (int i = 0; i < 10; i++) {
                CK(hipEventElapsedTime(&ms, ev[3 * i], ev[3 * i + 1])); k1 += ms;
                CK(hipEventElapsedTime(&ms, ev[3 * i + 1], ev[3 * i + 2])); wr += ms;
            }
            CK(hipEventElapsedTime(&tot, ev[0], ev[30]));
            printf("alternating cull / 411 MB %s write: cull %.1f us, write %.1f us, pair %.1f us\n", nt ? "non-temporal" : "plain", k1 * 100, wr * 100, tot * 100);
        }
    }
    return 0;
}
