// Hi-Z min-depth pyramid for gfx950.  Large images (launch_hiz has the threshold) take two launches:
// a streaming pass over the depth image that produces levels 0-3 from registers (hiz_rows_kernel),
// then the tiled kernel below applied to level 3, which finishes the pyramid.  Smaller images run the
// tiled kernel on the depth image itself (levels 0-6) and a one-workgroup launch for the rest.
//
// Reference: shaders/shader.comp:12-31 run once per level with a barrier in between
// (Engine.cpp:1834-1866): level 0 copies the depth image, level k takes the 2x2 minimum of
// level k-1; level extents are max(1, w>>k) x max(1, h>>k).  Because floor division composes,
// every texel of level k depends only on its own aligned 2^k x 2^k block of level 0, so one
// workgroup can produce levels 0..6 of a 64x64 tile from a single read of the depth image:
// levels 1-2 in registers (4x4 texels per thread), level 3 from LDS and 4-6 with lane shuffles by
// the workgroup's first wave.  Each workgroup contributes exactly one level-6 texel; the remaining
// levels are reduced either by a second one-workgroup launch or inside the same launch by the waves
// that finish last (two tiers of atomic tickets, write-through stores + agent-scope loads).  min() is
// written as (y < x ? y : x), GLSL's definition, so the result is bit-identical to the
// level-by-level evaluation for every input including NaN and signed zero.
// The kernels that read the depth image also report a NaN in it: that reduction can drop a NaN on the way up, so a
// coarse level does not bound the finer ones from below when one is present, which the instance-level occlusion test
// of the cull needs to know.  `nan_seen` points at two words: the kernels that read the image set word 1 when they
// meet a NaN, and the LAST launch of every build moves it to word 0 - what the culls read - and clears it
// (publish_nan_word), so word 0 always describes the image the pyramid was last built from.
#include <cstring>

#include "mcull_internal.h"

namespace mcull {

constexpr int kBlock = 256;
constexpr int kTile = 64;
constexpr uint32_t kMaxSinglePass = 8192;  // 128x128 tiles = 16x16 ticket groups + 1 top word (kHizCounterBytes)
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float glsl_min(float x, float y) { return (y < x) ? y : x; }

// By one thread of a launch that starts after every kernel that reads the depth image of this build has finished
// (stream order), or by a workgroup's thread 0 after a barrier behind its own reads.
__device__ __forceinline__ void publish_nan_word(uint32_t *__restrict__ nan_seen) {
    nan_seen[0] = __hip_atomic_load(nan_seen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    nan_seen[1] = 0u;
    nan_seen[kHizWinWord - kHizNanWord] = 0u;  // window tables made from an earlier build describe another image
}
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
                                                             uint32_t k_begin, uint32_t *__restrict__ nan_seen, uint32_t publish) {
    if (k_begin == 0) {
        const size_t n = (size_t)w * h;
        for (size_t i = threadIdx.x; i < n; i += kBlock) {
            const float d = depth[i];
            pyr[i] = d;  // shader.comp:19-21
            if (d != d) nan_seen[1] = 1u;
        }
        __syncthreads();
        if (threadIdx.x == 0) publish_nan_word(nan_seen);
        reduce_levels_from_global(pyr, tbl, 1, levels);
    } else {
        if (publish && threadIdx.x == 0) publish_nan_word(nan_seen);  // the launches that read the depth image have finished
        reduce_levels_from_global(pyr, tbl, k_begin, levels);
    }
}

// Texels handed from one workgroup to another inside a launch are stored and read back with agent-scope
// atomic accesses (no data race on them in the memory model's sense); what orders them is the ticket.
__device__ __forceinline__ float coherent_load(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One level inside a wave.  Lanes [0, N*N) hold, row-major, the N x N block of level K-1 whose
// top-left texel is (2*ox, 2*oy); returns in lanes [0, N*N/4) the N/2 x N/2 block of level K at
// (ox, oy) and stores its texels that lie inside the level.  Child indices clamp to the last valid
// texel of the source level inside the block (matters only when that extent has collapsed to 1).
// Every lane of the wave must call this (the shuffles are wave-wide).  WT: write-through store.
template <int N, bool WT>
__device__ __forceinline__ float wave_level(float v, uint32_t K, uint32_t levels, uint32_t ox, uint32_t oy,
                                            float *__restrict__ pyr, const uint32_t *__restrict__ tbl, uint32_t lane) {
    if (levels <= K) return v;  // uniform
    constexpr uint32_t H = N / 2;
    const uint32_t x = lane % H, y = (lane / H) % H;
    const uint32_t sw = tbl[K - 1], sh = tbl[16 + K - 1];
    const uint32_t mx = sw > 2u * ox ? min(sw - 1u - 2u * ox, (uint32_t)N - 1u) : 0u;
    const uint32_t my = sh > 2u * oy ? min(sh - 1u - 2u * oy, (uint32_t)N - 1u) : 0u;
    const uint32_t cx0 = min(2u * x, mx), cx1 = min(2u * x + 1u, mx), cy0 = min(2u * y, my), cy1 = min(2u * y + 1u, my);
    const float d0 = __shfl(v, (int)(cy0 * N + cx0)), d1 = __shfl(v, (int)(cy0 * N + cx1));
    const float d2 = __shfl(v, (int)(cy1 * N + cx0)), d3 = __shfl(v, (int)(cy1 * N + cx1));
    const float r = min4(d0, d1, d2, d3);
    const uint32_t lw = tbl[K], lh = tbl[16 + K];
    if (lane < H * H && ox + x < lw && oy + y < lh) {
        float *dst = pyr + tbl[32 + K] + (size_t)(oy + y) * lw + ox + x;
        if (WT) __hip_atomic_store(dst, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = r;
    }
    return r;
}

// Tiled kernel.  grid = (tiles_x, tiles_y), 256 threads, thread (tx,ty) = (t&15, t>>4) owns the
// 4x4 level-0 patch at (tile_x0 + 4*tx, tile_y0 + 4*ty).  VEC: rows are 16-byte aligned (w % 4 == 0).
// COPY: the source is the depth image and is copied out as level 0; otherwise the source is a level
// of the pyramid that is already in place (`depth` points at it, `tbl` at that level's table entries
// and `levels` counts from it), just written by the previous launch, so it is read with plain loads.
template <bool VEC, bool SINGLE_PASS, bool COPY>
__device__ __forceinline__ void hiz_tile_body(const float *__restrict__ depth, uint32_t w, uint32_t h,
                                                          float *__restrict__ pyr, const uint32_t *__restrict__ tbl,
                                                          uint32_t levels, uint32_t *__restrict__ counter,
                                                          uint32_t *__restrict__ nan_seen, float *__restrict__ strips,
                                                          uint32_t strips_w, uint32_t strips_h) {
    __shared__ float s2[16][17];
    const uint32_t t = threadIdx.x, tx = t & 15u, ty = t >> 4;
    const uint32_t x0 = blockIdx.x * kTile + 4 * tx, y0 = blockIdx.y * kTile + 4 * ty;

    // ---- level 0: read the 4x4 patch once, copy it out (shader.comp:19-21)
    float p[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t y = min(y0 + r, h - 1);
        if (VEC && x0 + 3 < w) {
            // the depth image is read once and level 0 is far larger than the caches: streaming hints
            const f4v *src = reinterpret_cast<const f4v *>(depth + (size_t)y * w + x0);
            const f4v q = COPY ? __builtin_nontemporal_load(src) : *src;
            p[r][0] = q.x; p[r][1] = q.y; p[r][2] = q.z; p[r][3] = q.w;
            if (COPY && y0 + r < h) __builtin_nontemporal_store(q, reinterpret_cast<f4v *>(pyr + (size_t)y * w + x0));
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t x = min(x0 + c, w - 1);
                p[r][c] = depth[(size_t)y * w + x];
                if (COPY && y0 + r < h && x0 + c < w) pyr[(size_t)y * w + x] = p[r][c];
            }
        }
    }
    if (COPY) {
        bool nan = false;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) nan |= p[r][c] != p[r][c];
        if (nan) nan_seen[1] = 1u;
    } else if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) {
        publish_nan_word(nan_seen);  // second launch of the streaming scheme: the pass over the depth image has finished
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

    // ---- levels 3..6: the 16x16 level-2 values are in LDS; from here on 64 lanes suffice, so waves
    // 1-3 leave (their wave slots go to the next workgroup's loads) and wave 0 reduces level 3 from
    // LDS and levels 4-6 with lane shuffles, without any further workgroup barrier.
    if (t >= 64u) {
        // Second launch of the streaming scheme (the source is level 3: the tile is 512 x 512 pixels of the image): the
        // strip maxima the first launch wrote are reduced once more, eight strips to one 64 x 64 block of the image,
        // stored behind the strips (strips_w per row of blocks), so that a large screen rectangle is covered by a few
        // blocks.  The tile's 8 x 8 blocks are one per lane of the last wave, which has nothing else left to do; the
        // first wave's ladder and tickets below do not wait for it.
        if (!COPY && strips && t >= 192u) {
            const uint32_t l = t - 192u;
            const uint32_t sc = blockIdx.x * 8u + (l & 7u), br = blockIdx.y * 8u + (l >> 3);
            const uint32_t bw = strips_w, bh = (strips_h + 7u) >> 3;  // blocks per row / rows of blocks of the image
            float mx = -__builtin_inff();  // stays so for a block outside the image
            if (sc < bw && br < bh) {
#pragma unroll
                for (uint32_t r = 0; r < 8; r++)
                    mx = fmaxf(mx, strips[(size_t)min(br * 8u + r, strips_h - 1u) * strips_w + sc]);
                strips[(size_t)strips_w * strips_h + (size_t)br * bw + sc] = mx;
            }
            // ... and three times more, 2 x 2 cells to one, with lane shuffles (every lane of the wave takes part): cells
            // of 128, 256 and 512 pixels, each table behind the last, so that a screen rectangle of any size up to a
            // quarter of the largest image is covered by a handful of cells (hiz_cell_table has the layout)
            size_t off = (size_t)strips_w * strips_h + (size_t)bw * bh;
            uint32_t cw = bw, ch = bh;
#pragma unroll
            for (uint32_t j = 1; j <= 3; j++) {
                mx = fmaxf(mx, __shfl_xor(mx, 1 << (j - 1), 64));
                mx = fmaxf(mx, __shfl_xor(mx, 8 << (j - 1), 64));
                cw = (cw + 1u) >> 1;
                ch = (ch + 1u) >> 1;
                const uint32_t m = (1u << j) - 1u;
                if (((l & 7u) & m) == 0u && ((l >> 3) & m) == 0u && (sc >> j) < cw && (br >> j) < ch && sc < bw && br < bh)
                    strips[off + (size_t)(br >> j) * cw + (sc >> j)] = mx;
                off += (size_t)cw * ch;
            }
        }
        return;
    }
    if (levels > 3) {
        const uint32_t ox2 = (blockIdx.x * kTile) >> 2, oy2 = (blockIdx.y * kTile) >> 2;  // tile origin in level 2
        const uint32_t x = t & 7u, y = t >> 3;
        const uint32_t sw = tbl[2], sh = tbl[16 + 2];
        // child indices clamp to the last valid texel of level 2 inside the tile (matters only when
        // that extent has collapsed to 1)
        const uint32_t mx = sw > ox2 ? min(sw - 1u - ox2, 15u) : 0u, my = sh > oy2 ? min(sh - 1u - oy2, 15u) : 0u;
        const uint32_t cx0 = min(2u * x, mx), cx1 = min(2u * x + 1u, mx), cy0 = min(2u * y, my), cy1 = min(2u * y + 1u, my);
        float v = min4(s2[cy0][cx0], s2[cy0][cx1], s2[cy1][cx0], s2[cy1][cx1]);
        const uint32_t lw = tbl[3], lh = tbl[16 + 3];
        const uint32_t gx3 = (ox2 >> 1) + x, gy3 = (oy2 >> 1) + y;
        if (gx3 < lw && gy3 < lh) pyr[tbl[32 + 3] + (size_t)gy3 * lw + gx3] = v;
        v = wave_level<8, false>(v, 4u, levels, ox2 >> 2, oy2 >> 2, pyr, tbl, t);
        v = wave_level<4, false>(v, 5u, levels, ox2 >> 3, oy2 >> 3, pyr, tbl, t);
        v = wave_level<2, SINGLE_PASS>(v, 6u, levels, ox2 >> 4, oy2 >> 4, pyr, tbl, t);
    }

    if (!SINGLE_PASS || levels <= 7) return;

    // ---- levels 7..: handed to whichever workgroup finishes last, by wave 0 alone (waves 1-3 are done
    // and leave; nothing below uses a workgroup barrier or LDS).  Each workgroup produced exactly one
    // level-6 texel, written by thread 0, which then takes a ticket in the counter of its 8x8 block of
    // tiles with a RELEASE at agent scope.  The last arriver of a block - and only it - executes an
    // ACQUIRE fence, owns that block's 8x8 level-6 texels = one texel per lane, reduces levels 7-9 with
    // lane shuffles, and takes a ticket in the top counter the same way; the last arriver there
    // reduces the (at most 16x16) level-9 texels to the apex.  Release on every ticket of a counter and
    // acquire by the one that reads the final value is the memory model's own pattern for "the last one
    // in sees everybody's writes" (the read-modify-writes of a counter form one release sequence), so the
    // hand-over does not rest on instruction order.  Measured against an earlier form that did (write-
    // through stores drained with a hand-placed s_waitcnt, relaxed tickets, no fences): + 0.5 us at
    // 3840x2160, + 2.3 us (2 %) at 8192^2, nothing at 1920x1080; the acquire by EVERY workgroup, which
    // invalidates an L2 each time, had cost 8 us.  No word receives more than 256 tickets (a single
    // word sustains only ~90 returning atomics per microsecond, which 16 384 workgroups would queue on).
    const uint32_t gx = blockIdx.x >> 3, gy = blockIdx.y >> 3;
    const uint32_t groups_x = (gridDim.x + 7u) >> 3, groups_y = (gridDim.y + 7u) >> 3;
    uint32_t last = 0u;
    if (t == 0) {
        const uint32_t members = min(8u, gridDim.x - 8u * gx) * min(8u, gridDim.y - 8u * gy);
        last = __hip_atomic_fetch_add(counter + 1u + gy * groups_x + gx, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == members - 1u;
    }
    if (__builtin_amdgcn_readfirstlane(last) == 0u) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    {
        const uint32_t w6 = tbl[6], h6 = tbl[16 + 6];
        const uint32_t x = min(8u * gx + (t & 7u), w6 - 1u), y = min(8u * gy + (t >> 3), h6 - 1u);
        float v = coherent_load(pyr + tbl[32 + 6] + (size_t)y * w6 + x);
        v = wave_level<8, false>(v, 7u, levels, 4u * gx, 4u * gy, pyr, tbl, t);
        v = wave_level<4, false>(v, 8u, levels, 2u * gx, 2u * gy, pyr, tbl, t);
        v = wave_level<2, true>(v, 9u, levels, gx, gy, pyr, tbl, t);
    }
    last = 0u;
    if (t == 0) {
        last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == groups_x * groups_y - 1u;
    }
    if (__builtin_amdgcn_readfirstlane(last) == 0u) return;
    // every ticket of this launch has been taken: leave the words zero for the next launch on the
    // stream (they are zeroed once at allocation; no per-call memset is needed)
    for (uint32_t i = t; i <= groups_x * groups_y; i += 64u) counter[i] = 0u;
    if (levels <= 10) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    {
        // lane (x,y) of an 8x8 arrangement owns the 2x2 patch of level 9 at (2x,2y) -> one texel of level 10
        const uint32_t w9 = tbl[9], h9 = tbl[16 + 9];
        const float *l9 = pyr + tbl[32 + 9];
        const uint32_t x0 = min(2u * (t & 7u), w9 - 1u), x1 = min(2u * (t & 7u) + 1u, w9 - 1u);
        const uint32_t y0 = min(2u * (t >> 3), h9 - 1u), y1 = min(2u * (t >> 3) + 1u, h9 - 1u);
        float v = min4(coherent_load(l9 + (size_t)y0 * w9 + x0), coherent_load(l9 + (size_t)y0 * w9 + x1),
                       coherent_load(l9 + (size_t)y1 * w9 + x0), coherent_load(l9 + (size_t)y1 * w9 + x1));
        const uint32_t w10 = tbl[10], h10 = tbl[16 + 10];
        if ((t & 7u) < w10 && (t >> 3) < h10) pyr[tbl[32 + 10] + (size_t)(t >> 3) * w10 + (t & 7u)] = v;
        v = wave_level<8, false>(v, 11u, levels, 0u, 0u, pyr, tbl, t);
        v = wave_level<4, false>(v, 12u, levels, 0u, 0u, pyr, tbl, t);
        v = wave_level<2, false>(v, 13u, levels, 0u, 0u, pyr, tbl, t);
    }
}

template <bool VEC, bool SINGLE_PASS, bool COPY>
__global__ __launch_bounds__(kBlock) void hiz_tile_kernel(const float *__restrict__ depth, uint32_t w, uint32_t h,
                                                          float *__restrict__ pyr, const uint32_t *__restrict__ tbl,
                                                          uint32_t levels, uint32_t *__restrict__ counter,
                                                          uint32_t *__restrict__ nan_seen, float *__restrict__ strips,
                                                          uint32_t strips_w, uint32_t strips_h) {
    hiz_tile_body<VEC, SINGLE_PASS, COPY>(depth, w, h, pyr, tbl, levels, counter, nan_seen, strips, strips_w, strips_h);
}

// Several pyramids of one size in one launch each (launch_hiz_views): blockIdx.z picks the view's buffers.  Built one after
// another, the small second launches of eight 8192^2 pyramids run one by one on a nearly idle chip.
struct HizViews {
    const float *depth[MCULL_MAX_VIEWS];
    float *pyramid[MCULL_MAX_VIEWS];
    const uint32_t *table[MCULL_MAX_VIEWS];
    uint32_t *counter[MCULL_MAX_VIEWS];
    uint32_t *nan_seen[MCULL_MAX_VIEWS];
    float *strips[MCULL_MAX_VIEWS];
};

template <bool VEC>
__global__ __launch_bounds__(kBlock) void hiz_tile_views_kernel(HizViews b, size_t source_offset, uint32_t table_shift, uint32_t w,
                                                                uint32_t h, uint32_t levels, uint32_t strips_w, uint32_t strips_h) {
    const uint32_t v = blockIdx.z;
    hiz_tile_body<VEC, false, false>(b.pyramid[v] + source_offset, w, h, b.pyramid[v], b.table[v] + table_shift, levels,
                                           b.counter[v], b.nan_seen[v], b.strips[v], strips_w, strips_h);
}

// the levels the tiles leave (from k_begin of the shifted table on), one workgroup per view
__global__ __launch_bounds__(kBlock) void hiz_top_views_kernel(HizViews b, uint32_t table_shift, uint32_t levels, uint32_t k_begin) {
    reduce_levels_from_global(b.pyramid[blockIdx.x], b.table[blockIdx.x] + table_shift, k_begin, levels);
}

constexpr uint32_t kRowsPerWave = 8;
constexpr uint64_t kStreamMinTexels = 6ull << 20;
// Eight rows x 4 texels of one lane: levels 0-3 stored, the level-3 texel returned (both lanes of a pair hold it).
template <bool VEC>
__device__ __forceinline__ float hiz_rows8(const float *__restrict__ depth, uint32_t w, uint32_t h, float *__restrict__ pyr,
                                           const uint32_t *__restrict__ tbl, uint32_t x0, uint32_t y0, uint32_t lane,
                                           uint32_t *__restrict__ nan_seen, float *__restrict__ strips, uint32_t strips_w) {
    // ---- level 0: copied out as read (shader.comp:19-21)
    float p[8][4];
    if (VEC) {
        const bool col_in = x0 < w;
        const uint32_t xc = min(x0, w - 4u);  // w % 4 == 0: columns past the edge re-read the last aligned group
        f4v q[8];
#pragma unroll
        for (uint32_t r = 0; r < 8; r++)
            q[r] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(depth + (size_t)min(y0 + r, h - 1u) * w + xc));
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
            if (col_in && y0 + r < h) __builtin_nontemporal_store(q[r], reinterpret_cast<f4v *>(pyr + (size_t)(y0 + r) * w + x0));
            p[r][0] = q[r].x; p[r][1] = q[r].y; p[r][2] = q[r].z; p[r][3] = q[r].w;
        }
    } else {
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
            const size_t row = (size_t)min(y0 + r, h - 1u) * w;
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                p[r][c] = depth[row + min(x0 + c, w - 1u)];
                if (y0 + r < h && x0 + c < w) pyr[row + x0 + c] = p[r][c];
            }
        }
    }
    {
        bool nan = false;
#pragma unroll
        for (uint32_t r = 0; r < 8; r++)
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) nan |= p[r][c] != p[r][c];
        if (nan) nan_seen[1] = 1u;
    }
    // ---- the LARGEST depth of each 64 x 8 strip of the image (16 lanes wide, this wave's 8 rows): what the
    // occlusion test looks at before it reads a texel of levels 0-3, whose footprint lies inside one strip -
    // a pair not behind the strip's largest depth is not behind any texel of it.  Rows and columns read
    // clamped belong to the same strip; lanes wholly past the right edge stay out.  fmaxf skips NaNs.
    if (strips) {
        float mx = -__builtin_inff();
        if (x0 < w) {
#pragma unroll
            for (uint32_t r = 0; r < 8; r++) mx = fmaxf(fmaxf(mx, fmaxf(p[r][0], p[r][1])), fmaxf(p[r][2], p[r][3]));
        }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
        if ((lane & 15u) == 0u && x0 < w) strips[(size_t)(y0 >> 3) * strips_w + (x0 >> 6)] = mx;
    }
    // ---- level 1: 4 rows x 2 texels
    float q1[4][2];
    {
        const uint32_t lw = tbl[1], lh = tbl[16 + 1];
        float *dst = pyr + tbl[32 + 1];
        const uint32_t bx = x0 >> 1, by = y0 >> 1;
        const bool vec2 = ((lw | tbl[32 + 1]) & 1u) == 0u;  // rows and level base 8-byte aligned
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            q1[j][0] = min4(p[2 * j][0], p[2 * j][1], p[2 * j + 1][0], p[2 * j + 1][1]);
            q1[j][1] = min4(p[2 * j][2], p[2 * j][3], p[2 * j + 1][2], p[2 * j + 1][3]);
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
    // ---- level 2: 2 rows x 1 texel
    float q2[2];
    {
        const uint32_t lw = tbl[2], lh = tbl[16 + 2];
        float *dst = pyr + tbl[32 + 2];
        const uint32_t bx = x0 >> 2, by = y0 >> 2;
#pragma unroll
        for (uint32_t j = 0; j < 2; j++) {
            q2[j] = min4(q1[2 * j][0], q1[2 * j][1], q1[2 * j + 1][0], q1[2 * j + 1][1]);
            if (bx < lw && by + j < lh) dst[(size_t)(by + j) * lw + bx] = q2[j];
        }
    }
    // ---- level 3: lanes 2i and 2i+1 hold the left and right children; both end up with the result
    const uint32_t lw = tbl[3], lh = tbl[16 + 3];
    const uint32_t bx = x0 >> 3, by = y0 >> 3;
    const bool right = (lane & 1u) != 0u;
    const float top_o = __shfl_xor(q2[0], 1), bot_o = __shfl_xor(q2[1], 1);
    const float q3 = min4(right ? top_o : q2[0], right ? q2[0] : top_o, right ? bot_o : q2[1], right ? q2[1] : bot_o);
    if (!right && bx < lw && by < lh) pyr[tbl[32 + 3] + (size_t)by * lw + bx] = q3;
    return q3;
}

// Streaming pass: levels 0-3 from one read of the depth image, no LDS and no barrier.  Every wave is
// on its own: lane l owns the 4 texels [4l, 4l+4) of 8 consecutive rows, so each load and store
// instruction of a wave covers one contiguous kilobyte of a row (the 64x64 tiles above touch four
// 256-byte pieces instead, which costs 13 % of the copy bandwidth on a large image).  Levels 1 and 2
// reduce inside the thread, level 3 joins two neighbouring lanes with a quad shuffle.  A workgroup is
// four such waves stacked vertically (256 x 32 texels).  Requires w, h >= 16, so that no level up to
// 3 has collapsed and every stored texel's children lie inside their level (rows and columns beyond
// the image are read clamped and never stored).  (Sixteen rows per wave with level 4 from a second
// shuffle measured slower on large images, in one piece - 114 registers - or as two halves.)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void hiz_rows_kernel(const float *__restrict__ depth, uint32_t w, uint32_t h,
                                                          float *__restrict__ pyr, const uint32_t *__restrict__ tbl,
                                                          uint32_t *__restrict__ nan_seen, float *__restrict__ strips,
                                                          uint32_t strips_w) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t x0 = blockIdx.x * 256u + 4u * lane;
    const uint32_t y0 = (blockIdx.y * (kBlock / 64u) + (threadIdx.x >> 6)) * kRowsPerWave;
    if (y0 >= h) return;  // the whole wave lies below the image
    hiz_rows8<VEC>(depth, w, h, pyr, tbl, x0, y0, lane, nan_seen, strips, strips_w);
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void hiz_rows_views_kernel(HizViews b, uint32_t w, uint32_t h, uint32_t strips_w) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t x0 = blockIdx.x * 256u + 4u * lane;
    const uint32_t y0 = (blockIdx.y * (kBlock / 64u) + (threadIdx.x >> 6)) * kRowsPerWave;
    if (y0 >= h) return;
    const uint32_t v = blockIdx.z;
    hiz_rows8<VEC>(b.depth[v], w, h, b.pyramid[v], b.table[v], x0, y0, lane, b.nan_seen[v], b.strips[v], strips_w);
}

template <bool COPY>
static hipError_t launch_tiles(const float *src, uint32_t w, uint32_t h, float *pyramid, uint32_t levels,
                               const uint32_t *level_table, uint32_t *counter, bool single_pass, uint32_t *nan_seen,
                               float *strips, uint32_t strips_w, uint32_t strips_h, hipStream_t st) {
    const dim3 grid((w + kTile - 1) / kTile, (h + kTile - 1) / kTile);
    const bool vec = (w % 4u) == 0u && (reinterpret_cast<uintptr_t>(src) & 15u) == 0u &&
                     (!COPY || (reinterpret_cast<uintptr_t>(pyramid) & 15u) == 0u);
    // COPY: this launch reads the depth image, so a launch behind it publishes the NaN word (the one-workgroup kernel,
    // which has levels to reduce only when the tiles leave some)
    auto publish = [&](uint32_t k_begin) {
        hipLaunchKernelGGL(hiz_generic_kernel, dim3(1), dim3(kBlock), 0, st, src, w, h, pyramid, level_table, levels, k_begin, nan_seen,
                           COPY ? 1u : 0u);
        return hipGetLastError();
    };
    if (single_pass && levels > 7 && w <= kMaxSinglePass && h <= kMaxSinglePass) {
        if (vec)
            hipLaunchKernelGGL((hiz_tile_kernel<true, true, COPY>), grid, dim3(kBlock), 0, st, src, w, h, pyramid,
                               level_table, levels, counter, nan_seen, strips, strips_w, strips_h);
        else
            hipLaunchKernelGGL((hiz_tile_kernel<false, true, COPY>), grid, dim3(kBlock), 0, st, src, w, h, pyramid,
                               level_table, levels, counter, nan_seen, strips, strips_w, strips_h);
        const hipError_t e = hipGetLastError();
        return (e != hipSuccess || !COPY) ? e : publish(levels);
    }
    if (vec)
        hipLaunchKernelGGL((hiz_tile_kernel<true, false, COPY>), grid, dim3(kBlock), 0, st, src, w, h, pyramid,
                           level_table, levels, counter, nan_seen, strips, strips_w, strips_h);
    else
        hipLaunchKernelGGL((hiz_tile_kernel<false, false, COPY>), grid, dim3(kBlock), 0, st, src, w, h, pyramid,
                           level_table, levels, counter, nan_seen, strips, strips_w, strips_h);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (levels > 7 || COPY) return publish(levels > 7 ? 7u : levels);
    return e;
}

// `mode`: 0 = by size: the streaming pass + tiles on level 3 (tickets hand the top levels over inside
// that launch) from kStreamMinTexels on, below that the tiles on the depth image + a one-workgroup
// launch for levels >= 7 (measured cross-over: 2560x1440 14.6 us against 15.4 us streamed, 3840x2160
// 25.0 us against 22.9 us); 1 = always the latter; 2 = tiles on the depth image with the top levels
// inside the same launch; 3 = always the streaming pass.
// Does launch_hiz(mode) run the streaming pass on a w x h image, i.e. does it write `strips`: the largest depth of
// every 64 x 8 strip of the image ((w + 63) / 64 per row, (h + 7) / 8 rows), followed by the largest depth of every
// 64 x 64 block ((w + 63) / 64 per row, (h + 63) / 64 rows; written for the blocks inside the cover of the second
// launch's tiles), then of every cell of 128, 256 and 512 pixels (each table half the last in both directions, rounded
// up) - hiz_strip_floats(w, h) floats in all?
bool hiz_writes_strips(uint32_t w, uint32_t h, int mode) {
    return w >= 32 && h >= 32 && (mode == 3 || (mode == 0 && (uint64_t)w * h >= kStreamMinTexels));
}

size_t hiz_strip_floats(uint32_t w, uint32_t h) {
    const uint32_t sw = (w + 63u) / 64u;
    size_t n = (size_t)sw * ((h + 7u) / 8u);
    uint32_t cw = sw, ch = (h + 63u) / 64u;
    for (int j = 0; j <= 3; j++) {  // cells of 64, 128, 256, 512 pixels
        n += (size_t)cw * ch;
        cw = (cw + 1u) >> 1;
        ch = (ch + 1u) >> 1;
    }
    return n;
}

// One table of sliding-window minima from the one before it (see kHizWinBase): next(x, y) = min of prev at (x, y),
// (x + half, y), (x, y + half), (x + half, y + half) wherever all four exist.  prev is the pyramid level itself for the
// first table.  `done` (the last table's launch only): the word that tells the culls the tables are complete.
__global__ __launch_bounds__(kBlock) void hiz_window_kernel(const float *__restrict__ prev, float *__restrict__ next, uint32_t cw,
                                                            uint32_t ch, uint32_t half, uint32_t *__restrict__ done) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (done && i == 0) *done = 1u;
    if (i >= cw * ch) return;
    const uint32_t y = i / cw, x = i - y * cw;
    if (x + 2u * half > cw || y + 2u * half > ch) return;
    const float *p = prev + (size_t)y * cw + x;
    next[i] = fminf(fminf(p[0], p[half]), fminf(p[(size_t)half * cw], p[(size_t)half * cw + half]));
}

uint32_t hiz_window_levels(uint32_t w, uint32_t h) {
    const uint32_t cw = w >> kHizWinBase, ch = h >> kHizWinBase, least = cw < ch ? cw : ch;
    uint32_t n = 0;
    while (n < kHizWinMaxLevels && (2u << n) <= least) n++;
    return n;
}

size_t hiz_window_floats(uint32_t w, uint32_t h) {
    return (size_t)hiz_window_levels(w, h) * (w >> kHizWinBase) * (h >> kHizWinBase);
}

hipError_t launch_hiz_windows(const float *pyramid, uint32_t *level_table, uint32_t w, uint32_t h, uint64_t base_offset,
                              float *windows, hipStream_t st) {
    const uint32_t n = hiz_window_levels(w, h), cw = w >> kHizWinBase, ch = h >> kHizWinBase;
    const float *prev = pyramid + base_offset;
    for (uint32_t j = 1; j <= n; j++) {
        float *next = windows + (size_t)(j - 1u) * cw * ch;
        hipLaunchKernelGGL(hiz_window_kernel, dim3((cw * ch + kBlock - 1) / kBlock), dim3(kBlock), 0, st, prev, next, cw, ch, 1u << (j - 1u),
                           j == n ? level_table + kHizWinWord : (uint32_t *)nullptr);
        prev = next;
    }
    return hipGetLastError();
}

hipError_t launch_hiz(const float *depth, uint32_t w, uint32_t h, float *pyramid, uint32_t levels,
                      const uint32_t *level_table, uint32_t *counter, uint32_t *nan_seen, float *strips, int mode, hipStream_t st) {
    if (w < 4 || h < 4) {
        hipLaunchKernelGGL(hiz_generic_kernel, dim3(1), dim3(kBlock), 0, st, depth, w, h, pyramid, level_table, levels,
                           0u, nan_seen, 1u);
        return hipGetLastError();
    }
    // level 3 must itself be at least 4x4, the tiled kernel's own precondition
    if (!hiz_writes_strips(w, h, mode))
        return launch_tiles<true>(depth, w, h, pyramid, levels, level_table, counter, mode == 2, nan_seen, nullptr, 0u, 0u, st);
    constexpr uint32_t base = 3;  // the last level the streaming pass writes
    const dim3 grid((w + 255u) / 256u, (h + 4u * kRowsPerWave - 1u) / (4u * kRowsPerWave));
    const bool vec = (w % 4u) == 0u && (reinterpret_cast<uintptr_t>(depth) & 15u) == 0u &&
                     (reinterpret_cast<uintptr_t>(pyramid) & 15u) == 0u;
    if (vec) hipLaunchKernelGGL((hiz_rows_kernel<true>), grid, dim3(kBlock), 0, st, depth, w, h, pyramid, level_table, nan_seen,
                                strips, (w + 63u) / 64u);
    else hipLaunchKernelGGL((hiz_rows_kernel<false>), grid, dim3(kBlock), 0, st, depth, w, h, pyramid, level_table, nan_seen, strips,
                            (w + 63u) / 64u);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess || levels <= base + 1u) return e;
    // the tiled kernel with the last level written above as its source: the table pointer and the level count shift
    size_t off = 0;
    for (uint32_t i = 0; i < base; i++) off += (size_t)max(1u, w >> i) * max(1u, h >> i);
    return launch_tiles<false>(pyramid + off, w >> base, h >> base, pyramid, levels - base, level_table + base, counter, true, nan_seen,
                               strips, (w + 63u) / 64u, (h + 7u) / 8u, st);
}

// n pyramids of one size, the streaming scheme with its launches batched over the views (three in all).  False: not applicable (another
// scheme for this size, or buffers that the vector forms cannot take) - the caller builds them one by one.
bool hiz_views_batchable(uint32_t w, uint32_t h, uint32_t levels, int mode) {
    // ... and the second launch must be the form that finishes the pyramid inside itself (launch_tiles)
    return hiz_writes_strips(w, h, mode) && levels > 3u + 7u && (w >> 3) <= kMaxSinglePass && (h >> 3) <= kMaxSinglePass;
}

hipError_t launch_hiz_views(uint32_t n, const float *const *depth, uint32_t w, uint32_t h, float *const *pyramid, uint32_t levels,
                            uint32_t *const *level_table, uint32_t *const *counter, float *const *strips, hipStream_t st) {
    constexpr uint32_t base = 3;
    HizViews b;
    std::memset(&b, 0, sizeof b);
    bool vec = (w % 4u) == 0u;
    for (uint32_t i = 0; i < n; i++) {
        b.depth[i] = depth[i];
        b.pyramid[i] = pyramid[i];
        b.table[i] = level_table[i];
        b.counter[i] = counter[i];
        b.nan_seen[i] = level_table[i] + kHizNanWord;
        b.strips[i] = strips[i];
        vec = vec && (reinterpret_cast<uintptr_t>(depth[i]) & 15u) == 0u && (reinterpret_cast<uintptr_t>(pyramid[i]) & 15u) == 0u;
    }
    const dim3 grid((w + 255u) / 256u, (h + 4u * kRowsPerWave - 1u) / (4u * kRowsPerWave), n);
    if (vec) hipLaunchKernelGGL((hiz_rows_views_kernel<true>), grid, dim3(kBlock), 0, st, b, w, h, (w + 63u) / 64u);
    else hipLaunchKernelGGL((hiz_rows_views_kernel<false>), grid, dim3(kBlock), 0, st, b, w, h, (w + 63u) / 64u);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t off = 0;
    for (uint32_t i = 0; i < base; i++) off += (size_t)max(1u, w >> i) * max(1u, h >> i);
    const uint32_t w3 = w >> base, h3 = h >> base;
    const dim3 tiles((w3 + kTile - 1) / kTile, (h3 + kTile - 1) / kTile, n);
    // level 3 starts 16-byte aligned when the pyramid does and the three levels below it hold a multiple of four texels
    const bool vec3 = (w3 % 4u) == 0u && vec && (off % 4u) == 0u;
    // The tiles WITHOUT the ticket hand-over, then one workgroup per view for the levels they leave: with thousands of
    // workgroups in one launch the tickets' releases (an L2 write-back each) queue up - eight 8192^2 pyramids finished
    // inside the tile launch took 55.6 us for it, against 17.7 + 4.2 us this way.
    if (vec3) hipLaunchKernelGGL((hiz_tile_views_kernel<true>), tiles, dim3(kBlock), 0, st, b, off, base, w3, h3, levels - base,
                                 (w + 63u) / 64u, (h + 7u) / 8u);
    else hipLaunchKernelGGL((hiz_tile_views_kernel<false>), tiles, dim3(kBlock), 0, st, b, off, base, w3, h3, levels - base,
                            (w + 63u) / 64u, (h + 7u) / 8u);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(hiz_top_views_kernel, dim3(n), dim3(kBlock), 0, st, b, base, levels - base, 7u);
    return hipGetLastError();
}

}  // namespace mcull
