// Single-launch Hi-Z min-depth pyramid for gfx950.
//
// Reference: shaders/shader.comp:12-31 run once per level with a barrier in between
// (Engine.cpp:1834-1866): level 0 copies the depth image, level k takes the 2x2 minimum of
// level k-1; level extents are max(1, w>>k) x max(1, h>>k).  Because floor division composes,
// every texel of level k depends only on its own aligned 2^k x 2^k block of level 0, so one
// workgroup can produce levels 0..6 of a 64x64 tile from a single read of the depth image:
// levels 1-2 in registers (4x4 texels per thread), 3-6 through LDS.  Each workgroup contributes
// exactly one level-6 texel; the workgroup that finishes last (atomic ticket, agent-scope
// release/acquire) reduces the remaining levels.  min() is written as (y < x ? y : x), GLSL's
// definition, so the result is bit-identical to the level-by-level evaluation for every input
// including NaN and signed zero.
#include "mcull_internal.h"

namespace mcull {

constexpr int kBlock = 256;
constexpr int kTile = 64;
constexpr uint32_t kTicketGroups = 255;  // + 1 top counter = 1 KiB of ticket words
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float glsl_min(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float min4(float d0, float d1, float d2, float d3) {
    return glsl_min(glsl_min(d0, d1), glsl_min(d2, d3));  // shader.comp:28
}

// Levels [k_begin, levels) computed one after another from global memory by ONE workgroup, with
// the reference's per-level semantics; child coordinates beyond the source extent (possible only
// when one dimension has collapsed to 1 before the other) clamp to the edge.
__device__ void reduce_levels_from_global(float *__restrict__ pyr, const uint32_t *__restrict__ tbl, uint32_t k_begin,
                                          uint32_t levels) {
    for (uint32_t k = k_begin; k < levels; k++) {
        const uint32_t dw = tbl[k], dh = tbl[16 + k], sw = tbl[k - 1], sh = tbl[16 + k - 1];
        const float *src = pyr + tbl[32 + k - 1];
        float *dst = pyr + tbl[32 + k];
        const uint32_t n = dw * dh;
        for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
            const uint32_t y = i / dw, x = i - y * dw;
            const uint32_t x0 = min(2 * x, sw - 1), x1 = min(2 * x + 1, sw - 1);
            const uint32_t y0 = min(2 * y, sh - 1), y1 = min(2 * y + 1, sh - 1);
            dst[i] = min4(src[(size_t)y0 * sw + x0], src[(size_t)y0 * sw + x1], src[(size_t)y1 * sw + x0],
                          src[(size_t)y1 * sw + x1]);
        }
        __syncthreads();  // this workgroup's stores of level k are read by its own next iteration
    }
}

// Whole pyramid by one workgroup (images with a side below 4, where levels 1-2 can already
// collapse) — also the reference-order evaluation used to cross-check the tiled kernel.
__global__ __launch_bounds__(kBlock) void hiz_generic_kernel(const float *__restrict__ depth, uint32_t w, uint32_t h,
                                                             float *__restrict__ pyr,
                                                             const uint32_t *__restrict__ tbl, uint32_t levels,
                                                             uint32_t k_begin) {
    if (k_begin == 0) {
        const size_t n = (size_t)w * h;
        for (size_t i = threadIdx.x; i < n; i += kBlock) pyr[i] = depth[i];  // shader.comp:19-21
        __syncthreads();
        reduce_levels_from_global(pyr, tbl, 1, levels);
    } else {
        reduce_levels_from_global(pyr, tbl, k_begin, levels);
    }
}

// Levels base+1 .. base+4 of one 16x16 block of level `base` held in s2, through LDS (8x8, 4x4, 2x2, 1).
// origin_x/y = level-0 coordinates of the block.  Child indices clamp to the last valid texel of the
// source level inside the block (a no-op unless one image dimension has collapsed to 1).  With
// WT_LAST the single texel of level base+4 is stored write-through at agent scope, because another
// workgroup will read it in the same launch.
template <bool WT_LAST>
__device__ __forceinline__ void lds_levels(float (*s2)[17], float (*s3)[9], float (*s4)[5], float (*s5)[3],
                                           float *__restrict__ pyr, const uint32_t *__restrict__ tbl, uint32_t levels,
                                           uint32_t base, uint32_t origin_x, uint32_t origin_y, uint32_t t) {
#define HIZ_LDS_LEVEL(STEP, SRC, DST, N, WT)                                                                    \
    if (levels > base + (STEP)) {                                                                               \
        const uint32_t K = base + (STEP);                                                                       \
        if (t < (N) * (N)) {                                                                                    \
            const uint32_t x = t % (N), y = t / (N);                                                            \
            const uint32_t sw = tbl[K - 1], sh = tbl[16 + K - 1];                                               \
            const uint32_t ox = origin_x >> (K - 1), oy = origin_y >> (K - 1);                                  \
            const uint32_t mx = sw > ox ? min(sw - 1 - ox, 2u * (N)-1u) : 0u;                                   \
            const uint32_t my = sh > oy ? min(sh - 1 - oy, 2u * (N)-1u) : 0u;                                   \
            const uint32_t cx0 = min(2 * x, mx), cx1 = min(2 * x + 1, mx);                                      \
            const uint32_t cy0 = min(2 * y, my), cy1 = min(2 * y + 1, my);                                      \
            const float v = min4(SRC[cy0][cx0], SRC[cy0][cx1], SRC[cy1][cx0], SRC[cy1][cx1]);                   \
            DST;                                                                                                \
            const uint32_t lw = tbl[K], lh = tbl[16 + K];                                                       \
            const uint32_t gx = (origin_x >> K) + x, gy = (origin_y >> K) + y;                                  \
            if (gx < lw && gy < lh) {                                                                           \
                float *dst = pyr + tbl[32 + K] + (size_t)gy * lw + gx;                                          \
                if (WT) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                 \
                else *dst = v;                                                                                  \
            }                                                                                                   \
        }                                                                                                       \
        __syncthreads();                                                                                        \
    }
    HIZ_LDS_LEVEL(1, s2, s3[y][x] = v, 8, false)
    HIZ_LDS_LEVEL(2, s3, s4[y][x] = v, 4, false)
    HIZ_LDS_LEVEL(3, s4, s5[y][x] = v, 2, false)
    HIZ_LDS_LEVEL(4, s5, (void)0, 1, WT_LAST)
#undef HIZ_LDS_LEVEL
}

// Stores an N x N register patch of level K whose top-left texel is (x0, y0).
template <int N>
__device__ __forceinline__ void store_patch(float *__restrict__ pyr, const uint32_t *__restrict__ tbl, uint32_t K,
                                            uint32_t levels, uint32_t x0, uint32_t y0, const float *v) {
    if (levels <= K) return;
    const uint32_t lw = tbl[K], lh = tbl[16 + K];
    float *dst = pyr + tbl[32 + K];
#pragma unroll
    for (int j = 0; j < N; j++)
#pragma unroll
        for (int i = 0; i < N; i++)
            if (x0 + i < lw && y0 + j < lh) dst[(size_t)(y0 + j) * lw + x0 + i] = v[j * N + i];
}

// Tiled kernel.  grid = (tiles_x, tiles_y), 256 threads, thread (tx,ty) = (t&15, t>>4) owns the
// 4x4 level-0 patch at (tile_x0 + 4*tx, tile_y0 + 4*ty).  VEC: rows are 16-byte aligned (w % 4 == 0).
template <bool VEC, bool SINGLE_PASS>
__global__ __launch_bounds__(kBlock) void hiz_tile_kernel(const float *__restrict__ depth, uint32_t w, uint32_t h,
                                                          float *__restrict__ pyr, const uint32_t *__restrict__ tbl,
                                                          uint32_t levels, uint32_t *__restrict__ counter) {
    __shared__ float s2[16][17];
    __shared__ float s3[8][9];
    __shared__ float s4[4][5];
    __shared__ float s5[2][3];
    __shared__ uint32_t s_last;
    const uint32_t t = threadIdx.x, tx = t & 15u, ty = t >> 4;
    const uint32_t x0 = blockIdx.x * kTile + 4 * tx, y0 = blockIdx.y * kTile + 4 * ty;

    // ---- level 0: read the 4x4 patch once, copy it out (shader.comp:19-21)
    float p[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t y = min(y0 + r, h - 1);
        if (VEC && x0 + 3 < w) {
            // the depth image is read once and level 0 is far larger than the caches: streaming hints
            const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(depth + (size_t)y * w + x0));
            p[r][0] = q.x; p[r][1] = q.y; p[r][2] = q.z; p[r][3] = q.w;
            if (y0 + r < h) __builtin_nontemporal_store(q, reinterpret_cast<f4v *>(pyr + (size_t)y * w + x0));
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t x = min(x0 + c, w - 1);
                p[r][c] = depth[(size_t)y * w + x];
                if (y0 + r < h && x0 + c < w) pyr[(size_t)y * w + x] = p[r][c];
            }
        }
    }
    if (levels == 1) return;

    // ---- level 1: 2x2 per thread
    float q1[2][2];
    {
        const uint32_t lw = tbl[1], lh = tbl[16 + 1];
        float *dst = pyr + tbl[32 + 1];
        const uint32_t bx = x0 >> 1, by = y0 >> 1;
        const bool vec2 = ((lw | tbl[32 + 1]) & 1u) == 0u;  // rows and level base 8-byte aligned
#pragma unroll
        for (int j = 0; j < 2; j++) {
#pragma unroll
            for (int i = 0; i < 2; i++)
                q1[j][i] = min4(p[2 * j][2 * i], p[2 * j][2 * i + 1], p[2 * j + 1][2 * i], p[2 * j + 1][2 * i + 1]);
            if (by + j < lh) {
                if (vec2 && bx + 1 < lw) {
                    *reinterpret_cast<float2 *>(dst + (size_t)(by + j) * lw + bx) = make_float2(q1[j][0], q1[j][1]);
                } else {
                    if (bx < lw) dst[(size_t)(by + j) * lw + bx] = q1[j][0];
                    if (bx + 1 < lw) dst[(size_t)(by + j) * lw + bx + 1] = q1[j][1];
                }
            }
        }
    }
    // ---- level 2: one texel per thread, kept in LDS for the levels above
    float q2 = min4(q1[0][0], q1[0][1], q1[1][0], q1[1][1]);
    if (levels > 2) {
        const uint32_t lw = tbl[2], lh = tbl[16 + 2];
        const uint32_t x = (x0 >> 2), y = (y0 >> 2);
        if (x < lw && y < lh) pyr[tbl[32 + 2] + (size_t)y * lw + x] = q2;
    }
    s2[ty][tx] = q2;
    __syncthreads();

    // ---- levels 3..6 through LDS
    lds_levels<SINGLE_PASS>(s2, s3, s4, s5, pyr, tbl, levels, 2u, blockIdx.x * kTile, blockIdx.y * kTile, t);

    if (!SINGLE_PASS || levels <= 7) return;

    // ---- levels 7..: done by the workgroup that arrives last.  Each workgroup produced exactly one
    // level-6 texel, written by thread 0 with a write-through (agent-scope) store; thread 0 drains
    // that store and then takes a ticket.  No L2 write-back is needed on the producer side.  The
    // last arriver acquires at agent scope (one lane), the barrier releases the other waves.
    // Tickets are two-level (up to kTicketGroups group counters + one top counter): a single word
    // sustains only ~90 returning atomics per microsecond, which 16 384 workgroups would queue on.
    if (t == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t n_wg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const uint32_t n_grp = n_wg < kTicketGroups ? n_wg : kTicketGroups;
        const uint32_t grp = lin % n_grp, members = (n_wg - grp + n_grp - 1u) / n_grp;
        uint32_t last = 0u;
        if (__hip_atomic_fetch_add(counter + 1u + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u)
            last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_grp - 1u;
        s_last = last;
    }
    __syncthreads();
    if (s_last == 0u) return;
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // every ticket of this launch has been taken: leave the words zero for the next launch on the
    // stream (they are zeroed once at allocation; no per-call memset is needed)
    if (t <= kTicketGroups) counter[t] = 0u;
    const uint32_t w6 = tbl[6], h6 = tbl[16 + 6];
    if (w6 > 128u || h6 > 128u) {  // images beyond 8192 texels a side: level by level from memory
        reduce_levels_from_global(pyr, tbl, 7, levels);
        return;
    }
    // Thread (tx,ty) owns the 8x8 patch of level 6 at (8*tx, 8*ty): levels 7-9 in registers, then the
    // 16x16 level-9 values go through the same LDS ladder for levels 10-13.  A child index beyond the
    // source extent can only occur when that extent is 1; it then clamps to texel 0.
    {
        const float *l6 = pyr + tbl[32 + 6];
        float q[8][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t y = min(8 * ty + r, h6 - 1);
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const uint32_t x = min(8 * tx + c, w6 - 1);
                q[r][c] = l6[(size_t)y * w6 + x];
            }
        }
        const bool cx6 = (w6 == 1u), cy6 = (h6 == 1u);
        float q7[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int x1 = cx6 ? 2 * i : 2 * i + 1, y1 = cy6 ? 2 * j : 2 * j + 1;
                q7[j][i] = min4(q[2 * j][2 * i], q[2 * j][x1], q[y1][2 * i], q[y1][x1]);
            }
        store_patch<4>(pyr, tbl, 7, levels, 4 * tx, 4 * ty, &q7[0][0]);
        const bool cx7 = (tbl[7] == 1u), cy7 = (tbl[16 + 7] == 1u);
        float q8[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int x1 = cx7 ? 2 * i : 2 * i + 1, y1 = cy7 ? 2 * j : 2 * j + 1;
                q8[j][i] = min4(q7[2 * j][2 * i], q7[2 * j][x1], q7[y1][2 * i], q7[y1][x1]);
            }
        store_patch<2>(pyr, tbl, 8, levels, 2 * tx, 2 * ty, &q8[0][0]);
        const bool cx8 = levels > 8 && tbl[8] == 1u, cy8 = levels > 8 && tbl[16 + 8] == 1u;
        const float q9 = min4(q8[0][0], q8[0][cx8 ? 0 : 1], q8[cy8 ? 0 : 1][0], q8[cy8 ? 0 : 1][cx8 ? 0 : 1]);
        store_patch<1>(pyr, tbl, 9, levels, tx, ty, &q9);
        __syncthreads();  // s2 is free again: every wave has finished the tile's LDS ladder
        s2[ty][tx] = q9;
        __syncthreads();
        lds_levels<false>(s2, s3, s4, s5, pyr, tbl, levels, 9u, 0u, 0u, t);
    }
}

hipError_t launch_hiz(const float *depth, uint32_t w, uint32_t h, float *pyramid, uint32_t levels,
                      const uint32_t *level_table, uint32_t *counter, bool single_pass, hipStream_t st) {
    if (w < 4 || h < 4) {
        hipLaunchKernelGGL(hiz_generic_kernel, dim3(1), dim3(kBlock), 0, st, depth, w, h, pyramid, level_table, levels,
                           0u);
        return hipGetLastError();
    }
    const dim3 grid((w + kTile - 1) / kTile, (h + kTile - 1) / kTile);
    const bool vec = (w % 4u) == 0u && (reinterpret_cast<uintptr_t>(depth) & 15u) == 0u &&
                     (reinterpret_cast<uintptr_t>(pyramid) & 15u) == 0u;
    if (single_pass && levels > 7) {
        if (vec)
            hipLaunchKernelGGL((hiz_tile_kernel<true, true>), grid, dim3(kBlock), 0, st, depth, w, h, pyramid,
                               level_table, levels, counter);
        else
            hipLaunchKernelGGL((hiz_tile_kernel<false, true>), grid, dim3(kBlock), 0, st, depth, w, h, pyramid,
                               level_table, levels, counter);
        return hipGetLastError();
    }
    if (vec)
        hipLaunchKernelGGL((hiz_tile_kernel<true, false>), grid, dim3(kBlock), 0, st, depth, w, h, pyramid,
                           level_table, levels, counter);
    else
        hipLaunchKernelGGL((hiz_tile_kernel<false, false>), grid, dim3(kBlock), 0, st, depth, w, h, pyramid,
                           level_table, levels, counter);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (levels > 7) {
        hipLaunchKernelGGL(hiz_generic_kernel, dim3(1), dim3(kBlock), 0, st, depth, w, h, pyramid, level_table, levels,
                           7u);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace mcull
