// Design-evidence benchmark (not part of the product): where the 64x64-tile pyramid kernel loses time
// against a plain copy - access pattern (tile-shaped against row-shaped waves, level 0 alone and levels
// 0-2) and the two stages of the streaming scheme timed separately.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 tools/exp_hiz.hip -o tools/exp_hiz
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../vulkan-renderer_amd/csrc/hiz_kernels.hip"
using namespace mcull;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_linear(const f4v *__restrict__ src, f4v *__restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// V1: the tile kernel's thread -> texel mapping, level 0 only
__global__ __launch_bounds__(256) void tile_copy(const float *__restrict__ depth, uint32_t w, uint32_t h, float *__restrict__ pyr) {
    const uint32_t t = threadIdx.x, tx = t & 15u, ty = t >> 4;
    const uint32_t x0 = blockIdx.x * 64 + 4 * tx, y0 = blockIdx.y * 64 + 4 * ty;
    f4v q[4];
#pragma unroll
    for (int r = 0; r < 4; r++) q[r] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(depth + (size_t)min(y0 + r, h - 1) * w + x0));
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (y0 + r < h) __builtin_nontemporal_store(q[r], reinterpret_cast<f4v *>(pyr + (size_t)(y0 + r) * w + x0));
}

// V2: + levels 1 and 2 from registers
__global__ __launch_bounds__(256) void tile_l2(const float *__restrict__ depth, uint32_t w, uint32_t h, float *__restrict__ pyr,
                                               const uint32_t *__restrict__ tbl) {
    const uint32_t t = threadIdx.x, tx = t & 15u, ty = t >> 4;
    const uint32_t x0 = blockIdx.x * 64 + 4 * tx, y0 = blockIdx.y * 64 + 4 * ty;
    float p[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(depth + (size_t)min(y0 + r, h - 1) * w + x0));
        p[r][0] = q.x; p[r][1] = q.y; p[r][2] = q.z; p[r][3] = q.w;
        if (y0 + r < h) __builtin_nontemporal_store(q, reinterpret_cast<f4v *>(pyr + (size_t)(y0 + r) * w + x0));
    }
    const uint32_t lw = tbl[1], lh = tbl[17];
    float *dst = pyr + tbl[33];
    float q1[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int i = 0; i < 2; i++) q1[j][i] = min4(p[2 * j][2 * i], p[2 * j][2 * i + 1], p[2 * j + 1][2 * i], p[2 * j + 1][2 * i + 1]);
        if ((y0 >> 1) + j < lh) *reinterpret_cast<float2 *>(dst + (size_t)((y0 >> 1) + j) * lw + (x0 >> 1)) = make_float2(q1[j][0], q1[j][1]);
    }
    const float q2 = min4(q1[0][0], q1[0][1], q1[1][0], q1[1][1]);
    if ((x0 >> 2) < tbl[2] && (y0 >> 2) < tbl[18]) pyr[tbl[34] + (size_t)(y0 >> 2) * tbl[2] + (x0 >> 2)] = q2;
}

// V3: wide mapping, level 0 only: workgroup = 256 x 16 texels, wave w = rows 4w..4w+3, lane = 4 texels of a row
__global__ __launch_bounds__(256) void wide_copy(const float *__restrict__ depth, uint32_t w, uint32_t h, float *__restrict__ pyr) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x0 = blockIdx.x * 256 + 4 * lane, y0 = blockIdx.y * 16 + 4 * wave;
    if (x0 >= w) return;
    f4v q[4];
#pragma unroll
    for (int r = 0; r < 4; r++) q[r] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(depth + (size_t)min(y0 + r, h - 1) * w + x0));
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (y0 + r < h) __builtin_nontemporal_store(q[r], reinterpret_cast<f4v *>(pyr + (size_t)(y0 + r) * w + x0));
}

// V4: wide mapping + levels 1 and 2
__global__ __launch_bounds__(256) void wide_l2(const float *__restrict__ depth, uint32_t w, uint32_t h, float *__restrict__ pyr,
                                               const uint32_t *__restrict__ tbl) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x0 = blockIdx.x * 256 + 4 * lane, y0 = blockIdx.y * 16 + 4 * wave;
    if (x0 >= w) return;
    float p[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(depth + (size_t)min(y0 + r, h - 1) * w + x0));
        p[r][0] = q.x; p[r][1] = q.y; p[r][2] = q.z; p[r][3] = q.w;
        if (y0 + r < h) __builtin_nontemporal_store(q, reinterpret_cast<f4v *>(pyr + (size_t)(y0 + r) * w + x0));
    }
    const uint32_t lw = tbl[1], lh = tbl[17];
    float *dst = pyr + tbl[33];
    float q1[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int i = 0; i < 2; i++) q1[j][i] = min4(p[2 * j][2 * i], p[2 * j][2 * i + 1], p[2 * j + 1][2 * i], p[2 * j + 1][2 * i + 1]);
        if ((y0 >> 1) + j < lh) *reinterpret_cast<float2 *>(dst + (size_t)((y0 >> 1) + j) * lw + (x0 >> 1)) = make_float2(q1[j][0], q1[j][1]);
    }
    const float q2 = min4(q1[0][0], q1[0][1], q1[1][0], q1[1][1]);
    if ((x0 >> 2) < tbl[2] && (y0 >> 2) < tbl[18]) pyr[tbl[34] + (size_t)(y0 >> 2) * tbl[2] + (x0 >> 2)] = q2;
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
        CK(hipMalloc(&dtbl, sizeof tbl)); CK(hipMalloc(&counter, kHizCounterBytes)); CK(hipMemset(counter, 0, kHizCounterBytes));
        CK(hipMemcpy(dtbl, tbl, sizeof tbl, hipMemcpyHostToDevice));
        CK(hipMemset(depth, 0x3c, (size_t)w * h * 4));
        auto timeit = [&](const char *name, auto &&launch) {
            for (int i = 0; i < 5; i++) launch();
            CK(hipDeviceSynchronize());
            float sum = 0, best = 1e9f;
            for (int r = 0; r < 20; r++) {
                CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; best = std::min(best, ms);
            }
            CK(hipGetLastError());
            printf("%4ux%-4u %-40s avg %8.1f us  best %8.1f us\n", w, h, name, sum * 50, best * 1000);
        };
        const dim3 tg((w + 63) / 64, (h + 63) / 64), wg((w + 255) / 256, (h + 15) / 16);
        const size_t n4 = (size_t)w * h / 4;
        timeit("linear copy", [&] { hipLaunchKernelGGL(copy_linear, dim3((uint32_t)((n4 + 255) / 256)), dim3(256), 0, 0, (const f4v *)depth, (f4v *)pyr, n4); });
        timeit("tile mapping, L0", [&] { hipLaunchKernelGGL(tile_copy, tg, dim3(256), 0, 0, depth, w, h, pyr); });
        timeit("tile mapping, L0-2", [&] { hipLaunchKernelGGL(tile_l2, tg, dim3(256), 0, 0, depth, w, h, pyr, dtbl); });
        timeit("wide mapping, L0", [&] { hipLaunchKernelGGL(wide_copy, wg, dim3(256), 0, 0, depth, w, h, pyr); });
        timeit("wide mapping, L0-2", [&] { hipLaunchKernelGGL(wide_l2, wg, dim3(256), 0, 0, depth, w, h, pyr, dtbl); });
        timeit("product, tiles + top launch", [&] { CK(launch_hiz(depth, w, h, pyr, levels, dtbl, counter, counter + 500, nullptr, 1, 0)); });
        timeit("product, rows + tiles on level 3", [&] { CK(launch_hiz(depth, w, h, pyr, levels, dtbl, counter, counter + 500, nullptr, 0, 0)); });
        CK(hipFree(depth)); CK(hipFree(pyr)); CK(hipFree(dtbl)); CK(hipFree(counter));
    }
    return 0;
}
