// Probe (not part of the product): what would fusing the two launches of the large-image pyramid cost?  The
// streaming pass alone, and the streaming pass with what a fused version adds to EVERY workgroup - draining its
// stores and taking a ticket in the counter of its 512x512 region - without the tile work the last arriver would do.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 tools/exp_hiz_fused.hip -o tools/exp_hiz_fused
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../vulkan-renderer_amd/csrc/hiz_kernels.hip"

using namespace mcull;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(kBlock) void rows_ticket_kernel(const float *__restrict__ depth, uint32_t w, uint32_t h,
                                                             float *__restrict__ pyr, const uint32_t *__restrict__ tbl,
                                                             uint32_t *__restrict__ counter, uint32_t regions_x) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t x0 = blockIdx.x * 256u + 4u * lane;
    const uint32_t y0 = (blockIdx.y * (kBlock / 64u) + (threadIdx.x >> 6)) * kRowsPerWave;
    if (y0 < h) hiz_rows8<true>(depth, w, h, pyr, tbl, x0, y0, lane, counter + 500, nullptr, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t region = (blockIdx.y / 16u) * regions_x + blockIdx.x / 2u;  // 512x512 texels = 2 x 16 workgroups
        const uint32_t old = __hip_atomic_fetch_add(counter + 1u + region, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0xffffffffu) pyr[0] = 0.0f;  // keep the returning atomic
    }
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t sizes[][2] = {{3840, 2160}, {4096, 4096}, {7680, 4320}, {8192, 8192}};
    for (auto &sz : sizes) {
        const uint32_t w = sz[0], h = sz[1];
        uint32_t levels = 1; for (uint32_t m = std::max(w, h); m > 1; m >>= 1) levels++;
        uint32_t tbl[48] = {0}; uint64_t total = 0;
        for (uint32_t i = 0; i < levels; i++) { tbl[i] = std::max(1u, w >> i); tbl[16 + i] = std::max(1u, h >> i); tbl[32 + i] = (uint32_t)total; total += (uint64_t)tbl[i] * tbl[16 + i]; }
        float *depth, *pyr; uint32_t *dtbl, *counter;
        CK(hipMalloc(&depth, (size_t)w * h * 4)); CK(hipMalloc(&pyr, total * 4));
        CK(hipMalloc(&dtbl, sizeof tbl)); CK(hipMalloc(&counter, 1 << 16)); CK(hipMemset(counter, 0, 1 << 16));
        CK(hipMemcpy(dtbl, tbl, sizeof tbl, hipMemcpyHostToDevice));
        CK(hipMemset(depth, 0x3c, (size_t)w * h * 4));
        auto timeit = [&](const char *name, auto &&launch) {
            for (int i = 0; i < 3; i++) launch();
            CK(hipDeviceSynchronize());
            float sum = 0, best = 1e9f;
            for (int r = 0; r < 10; r++) {
                CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; best = std::min(best, ms);
            }
            CK(hipGetLastError());
            printf("%4ux%-4u %-60s avg %8.1f us  best %8.1f us\n", w, h, name, sum * 100, best * 1000);
        };
        const dim3 grid((w + 255u) / 256u, (h + 31u) / 32u);
        timeit("both launches (the product)", [&] { CK(launch_hiz(depth, w, h, pyr, levels, dtbl, counter, counter + 500, nullptr, 0, 0)); });
        timeit("streaming pass alone (levels 0-3)", [&] {
            hipLaunchKernelGGL((hiz_rows_kernel<true>), grid, dim3(kBlock), 0, 0, depth, w, h, pyr, dtbl, counter + 500, nullptr, 0u);
        });
        timeit("streaming pass + drain and region ticket per workgroup", [&] {
            hipLaunchKernelGGL(rows_ticket_kernel, grid, dim3(kBlock), 0, 0, depth, w, h, pyr, dtbl, counter, (w + 511u) / 512u);
        });
        CK(hipFree(depth)); CK(hipFree(pyr)); CK(hipFree(dtbl)); CK(hipFree(counter));
    }
    return 0;
}
