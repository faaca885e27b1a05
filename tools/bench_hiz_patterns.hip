// Design-evidence benchmark for the Hi-Z pyramid (not part of the product): the single-launch tiled
// kernel against (a) the same tiles followed by a second launch for the top levels and (b) the
// reference's scheme, one 8x8-workgroup dispatch per level with a dependency in between
// (shaders/shader.comp:12-31 driven by Engine.cpp:1840-1865).  Build + run:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 tools/bench_hiz_patterns.hip -o tools/bench_hiz_patterns
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../vulkan-renderer_amd/csrc/hiz_kernels.hip"

using namespace mcull;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// shader.comp as written: one thread per destination texel, 8x8 workgroups, one level per launch
__global__ __launch_bounds__(64) void per_level_kernel(const float *__restrict__ src, uint32_t sw, uint32_t sh,
                                                       float *__restrict__ dst, uint32_t dw, uint32_t dh, int downsampling) {
    const uint32_t x = blockIdx.x * 8 + (threadIdx.x & 7), y = blockIdx.y * 8 + (threadIdx.x >> 3);
    if (x >= dw || y >= dh) return;
    if (downsampling == 0) {
        dst[(size_t)y * dw + x] = src[(size_t)y * sw + x];
    } else {
        const uint32_t x0 = min(2 * x, sw - 1), x1 = min(2 * x + 1, sw - 1), y0 = min(2 * y, sh - 1), y1 = min(2 * y + 1, sh - 1);
        dst[(size_t)y * dw + x] = min4(src[(size_t)y0 * sw + x0], src[(size_t)y0 * sw + x1], src[(size_t)y1 * sw + x0],
                                       src[(size_t)y1 * sw + x1]);
    }
}

// streaming floor: level 0 alone (a non-temporal copy of the depth image)
__global__ __launch_bounds__(256) void copy_floor_kernel(const f4v *__restrict__ src, f4v *__restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t sizes[][2] = {{800, 600}, {1280, 720}, {1920, 1080}, {2560, 1440}, {3840, 2160}, {4096, 4096}, {7680, 4320}, {8192, 8192}};
    for (auto &sz : sizes) {
        const uint32_t w = sz[0], h = sz[1];
        uint32_t levels = 1; for (uint32_t m = std::max(w, h); m > 1; m >>= 1) levels++;
        uint32_t tbl[48] = {0}; uint64_t total = 0;
        for (uint32_t i = 0; i < levels; i++) { tbl[i] = std::max(1u, w >> i); tbl[16 + i] = std::max(1u, h >> i); tbl[32 + i] = (uint32_t)total; total += (uint64_t)tbl[i] * tbl[16 + i]; }
        float *depth, *pyr, *pyr_ref; uint32_t *dtbl, *counter;
        CK(hipMalloc(&depth, (size_t)w * h * 4)); CK(hipMalloc(&pyr, total * 4)); CK(hipMalloc(&pyr_ref, total * 4));
        CK(hipMalloc(&dtbl, sizeof tbl)); CK(hipMalloc(&counter, kHizCounterBytes)); CK(hipMemset(counter, 0, kHizCounterBytes));
        CK(hipMemcpy(dtbl, tbl, sizeof tbl, hipMemcpyHostToDevice));
        {
            std::vector<float> hd((size_t)w * h); srand(3);
            for (auto &x : hd) x = rand() / (float)RAND_MAX;
            CK(hipMemcpy(depth, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
        }
        const double algo = 4.0 * w * h + 4.0 * total;
        auto timeit = [&](const char *name, auto &&launch) {
            for (int i = 0; i < 3; i++) launch();
            CK(hipDeviceSynchronize());
            float sum = 0, best = 1e9f;
            for (int r = 0; r < 10; r++) {
                CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; best = std::min(best, ms);
            }
            CK(hipGetLastError());
            printf("%4ux%-4u %-52s avg %8.1f us  best %8.1f us  %6.2f TB/s algorithmic\n", w, h, name, sum * 100, best * 1000,
                   algo / (sum / 10 * 1e-3) / 1e12);
        };
        timeit("streaming pass (levels 0-3) + tiles on level 3", [&] { CK(launch_hiz(depth, w, h, pyr, levels, dtbl, counter, counter + 500, nullptr, 0, 0)); });
        timeit("64x64 tiles, single launch (last-arriver tail)", [&] { CK(launch_hiz(depth, w, h, pyr, levels, dtbl, counter, counter + 500, nullptr, 2, 0)); });
        timeit("64x64 tiles + one-workgroup launch for the top", [&] { CK(launch_hiz(depth, w, h, pyr, levels, dtbl, counter, counter + 500, nullptr, 1, 0)); });
        if (w % 4 == 0)
            timeit("floor: non-temporal copy of level 0 only", [&] {
                const size_t n4 = (size_t)w * h / 4;
                hipLaunchKernelGGL(copy_floor_kernel, dim3((uint32_t)((n4 + 255) / 256)), dim3(256), 0, 0, (const f4v *)depth, (f4v *)pyr, n4);
            });
        timeit("reference scheme: one dispatch per level", [&] {
            for (uint32_t i = 0; i < levels; i++) {
                const float *src = i == 0 ? depth : pyr_ref + tbl[32 + i - 1];
                const uint32_t sw = i == 0 ? w : tbl[i - 1], sh = i == 0 ? h : tbl[16 + i - 1];
                hipLaunchKernelGGL(per_level_kernel, dim3((tbl[i] + 7) / 8, (tbl[16 + i] + 7) / 8), dim3(64), 0, 0, src, sw, sh,
                                   pyr_ref + tbl[32 + i], tbl[i], tbl[16 + i], (int)i);
            }
        });
        // the three schemes must agree bit for bit
        std::vector<uint32_t> a(total), b(total);
        CK(launch_hiz(depth, w, h, pyr, levels, dtbl, counter, counter + 500, nullptr, 0, 0));
        CK(hipMemcpy(a.data(), pyr, total * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), pyr_ref, total * 4, hipMemcpyDeviceToHost));
        printf("%4ux%-4u streaming pass + tiles == per-level result: %s\n", w, h, a == b ? "yes" : "NO");
        CK(hipFree(depth)); CK(hipFree(pyr)); CK(hipFree(pyr_ref)); CK(hipFree(dtbl)); CK(hipFree(counter));
    }
    return 0;
}
