// Depth source for the Hi-Z pyramid: a depth-only software rasteriser over the visible list.
//
// Stands in for the reference's depth pre-pass (Engine.cpp:2397-2441 running shader.mesh:60-101 on
// the hardware rasteriser): gl_Position = ((proj*view)*model)*vec4(v,1), viewport (0,0,W,H) with
// depth 0..1 (Engine.cpp:2413-2420), back faces culled with counter-clockwise front faces
// (Engine.cpp:961-962), depth test GREATER_OR_EQUAL on a buffer cleared to 0 (Engine.cpp:975,:2404).
// What a hardware rasteriser leaves implementation-defined is fixed here as: float edge functions
// evaluated at pixel centres, top-left fill rule, screen-space barycentric depth, fragments outside
// 0 <= z <= 1 dropped; triangles crossing the near plane z <= w are clipped in clip space
// (Sutherland-Hodgman, intersections always evaluated from the inside vertex towards the outside one,
// t = d_p / (d_p - d_q) with d = w - z), whatever then still has w <= 0 is dropped.  Every
// expression fixes its association order (no fused multiply-add, correctly rounded division), and
// the per-pixel result is a maximum, so the image is independent of the order in which triangles
// arrive and equals a sequential evaluation bit for bit.
//
// One wave per visible (drawID, meshletID) entry: lane l projects local vertex l (a meshlet has at
// most 64), triangles fetch their corners with lane shuffles, lane t rasterises triangle t and
// t + 64.  Triangles whose pixel box exceeds kSmallBox pixels are rasterised by the whole wave, one
// after another, so a single lane never walks a large box alone.  Depth is stored as the float's
// bit pattern and merged with atomicMax (non-negative floats order like unsigned integers).
#include <cstdlib>

#include "mcull_internal.h"

namespace mcull {

constexpr int kRasterBlock = 256;
// pixels a lane rasterises on its own: 2 to 128 measure the same on sub-pixel triangles (4096 instances at 4K); with the
// reference's scene at 800x600 (triangles of tens of pixels) 8 costs 114 us against 77 us for 32, and 128 gains nothing
constexpr uint32_t kSmallBox = 32;

// A box above kHugeBox pixels is not walked by the wave that found it: the triangle goes into a table and its
// box, cut into kJobTile x kJobTile pixel pieces, into a job queue that a second kernel spreads over the whole
// device (one wave per piece).  Without it a single screen-filling quad keeps ONE wave busy for 35 ms at
// 3840x2160 (2 ms at 800x600, 284 ms at 8192^2) while the rest of the device idles, and a mesh seen from close by
// (triangles of hundreds of pixels) leaves most of the device waiting for the few waves that own them.  Same
// expressions on the same screen-space values, merged by the same atomic maximum: the image does not change.
// Limit: kitten close-ups at 3840x2160 (1 instance / 8 instances / 1 very close) take 0.31 / 0.27 / 0.21 ms with
// 512, 0.33 / 0.29 / 0.27 with 1024, 0.84 / 0.31 / 1.50 with 16384, 0.34 / 0.67 / 0.22 with 256 (two reservations and a job per triangle cost more than walking so small a box; a queue four times as large measures the same);
// the reference scene at 800x600 and sub-pixel triangles measure the same with any of them.
constexpr uint32_t kHugeBox = 512u;
constexpr uint32_t kJobTile = 64u;
constexpr uint32_t kQueueTriangles = 1u << 16;   // 3 MB of triangle records
constexpr uint32_t kQueueJobs = 1u << 20;        // 8 MB of jobs; what does not fit is walked in place as before

struct HugeTri {       // 48 bytes
    float x0, y0, z0, x1, y1, z1, x2, y2, z2;
    uint32_t origin;   // box origin  x | y << 16
    uint32_t extent;   // box extent  (w - 1) | (h - 1) << 16
    uint32_t pad;
};
struct HugeJob {
    uint32_t tri;      // index into the triangle table, 0xffffffff = nothing to do
    uint32_t piece;    // piece of the box  x | y << 16, in units of kJobTile
};
// counters[0] = triangles reserved, counters[1] = jobs reserved (both may overshoot the capacities), counters[2] =
// workgroups of the job kernel that are done; all zero between calls (the job kernel's last workgroup re-zeroes them)
struct HugeQueue {
    uint32_t *counters;
    HugeTri *tris;
    HugeJob *jobs;
    uint32_t tri_cap, job_cap;  // <= kQueueTriangles / kQueueJobs
};

struct RasterConst {
    float vp[16];  // proj * view, column-major
    float wf, hf;
    uint32_t w, h;
    uint32_t small_box, huge_box;
};

struct Tri {
    float x0, y0, z0, x1, y1, z1, x2, y2, z2;
};

__device__ __forceinline__ float edge_fn(float ax, float ay, float bx, float by, float px, float py) {
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
__device__ __forceinline__ bool edge_top_left(float ax, float ay, float bx, float by) {
    const float dx = bx - ax, dy = by - ay;
    return dy > 0.0f || (dy == 0.0f && dx < 0.0f);
}

struct ClipV {
    float x, y, z, w;
};
struct ScreenV {
    float x, y, z;
    int ok;
};

__device__ __forceinline__ ScreenV to_screen(const ClipV &c, float wf, float hf) {
    ScreenV r;
    r.ok = c.w > 0.0f;
    const float nx = c.x / c.w, ny = c.y / c.w;
    r.z = c.z / c.w;
    r.x = (nx * 0.5f + 0.5f) * wf;  // viewport transform, Engine.cpp:2413-2420
    r.y = (ny * 0.5f + 0.5f) * hf;
    return r;
}

// intersection of the edge p -> q with the near plane, p inside (w - z >= 0), q outside
__device__ __forceinline__ ClipV clip_intersect(const ClipV &p, const ClipV &q) {
    const float dp = p.w - p.z, dq = q.w - q.z;
    const float t = dp / (dp - dq);
    ClipV r;
    r.x = p.x + t * (q.x - p.x);
    r.y = p.y + t * (q.y - p.y);
    r.z = p.z + t * (q.z - p.z);
    r.w = p.w + t * (q.w - p.w);
    return r;
}

// one pixel of one front-facing triangle (area < 0)
__device__ __forceinline__ void raster_pixel(const Tri &t, float area, bool tl0, bool tl1, bool tl2, uint32_t px, uint32_t py,
                                             uint32_t w, uint32_t *__restrict__ depth_bits) {
    const float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
    const float w0 = edge_fn(t.x1, t.y1, t.x2, t.y2, cx, cy);
    const float w1 = edge_fn(t.x2, t.y2, t.x0, t.y0, cx, cy);
    const float w2 = edge_fn(t.x0, t.y0, t.x1, t.y1, cx, cy);
    if (!((w0 < 0.0f || (w0 == 0.0f && tl0)) && (w1 < 0.0f || (w1 == 0.0f && tl1)) && (w2 < 0.0f || (w2 == 0.0f && tl2)))) return;
    const float z = ((w0 * t.z0 + w1 * t.z1) + w2 * t.z2) / area;
    if (!(z >= 0.0f && z <= 1.0f)) return;
    const uint32_t bits = __float_as_uint(z) & 0x7fffffffu;  // -0 -> +0
    uint32_t *dst = depth_bits + (size_t)py * w + px;
    // no read-and-compare in front of it: the atomic returns nothing, so the pixel loop never waits for
    // memory, whereas a filtering read costs one round trip per pixel (the reference's scene at 800x600:
    // 76 us with the read, 37 us without; 4096 instances at 4K: 0.77 ms and 0.69 ms)
    atomicMax(dst, bits);
}

// What a rasteriser call needs before its waves start, in one launch: the depth image cleared to 0 (n_pixels = 0: the
// caller keeps its contents) and mvp = (proj*view) * model of every instance, element i = 4*c + r summed in the order
// k = 0..3 (one thread per element).  The rasteriser's waves used to evaluate the matrix per visible meshlet - sixteen
// lanes, then sixteen broadcasts into SGPRs, about 40 of the 333 vector instructions a meshlet cost - although a scene
// has hundreds of visible meshlets per instance; they now fetch the sixteen floats of their instance with scalar loads,
// one iteration ahead.  Taking the place of the clear's memset, it adds no launch to a frame.
// VP: proj*view is read from a device record written earlier on the stream instead of the kernel arguments
// (replayable HIP graphs with a moving camera).
template <bool VP>
__global__ __launch_bounds__(kRasterBlock) void raster_prepare_kernel(const float *__restrict__ transforms, uint64_t n_elements,
                                                                     RasterConst rc, const ViewConst *__restrict__ view_dev,
                                                                     float *__restrict__ mvp, uint32_t *__restrict__ depth_bits,
                                                                     uint64_t n_pixels) {
    const uint64_t tid = (uint64_t)blockIdx.x * kRasterBlock + threadIdx.x, stride = (uint64_t)gridDim.x * kRasterBlock;
    const float *vp = VP ? view_dev->vp : rc.vp;
    for (uint64_t i = tid; i < n_elements; i += stride) {
        const uint32_t r = (uint32_t)i & 3u;
        const float4 col = *reinterpret_cast<const float4 *>(transforms + (i & ~(uint64_t)3));  // column c of the instance's model matrix
        mvp[i] = ((vp[r] * col.x + vp[4 + r] * col.y) + vp[8 + r] * col.z) + vp[12 + r] * col.w;
    }
    const uint64_t n4 = n_pixels / 4;  // the image starts on a 16-byte boundary
    for (uint64_t i = tid; i < n4; i += stride) reinterpret_cast<uint4 *>(depth_bits)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < n_pixels - n4 * 4) depth_bits[n4 * 4 + tid] = 0u;
}

__global__ __launch_bounds__(kRasterBlock) void depth_raster_kernel(
    const mcull_visible *__restrict__ list, const unsigned long long *__restrict__ total, uint64_t capacity,
    const mcull_meshlet *__restrict__ meshlets, const uint32_t *__restrict__ meshlet_data,
    const mcull_vertex *__restrict__ vertices, const float *__restrict__ mvp_all, RasterConst rc,
    uint32_t *__restrict__ depth_bits, HugeQueue queue) {
    const uint32_t lane = threadIdx.x & 63u;
    // the wave index is made provably wave-uniform, so that the list entry, the meshlet header and
    // the matrix are fetched with scalar loads into SGPRs
    const uint64_t wave0 = (uint64_t)blockIdx.x * (kRasterBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (kRasterBlock / 64);
    unsigned long long n = *total;
    if (n > capacity) n = capacity;
    if (wave0 >= n) return;
    struct Hdr {
        uint32_t off, vc, tc;
    };
    auto load_entry = [&](uint64_t i) { return list[i < n ? i : n - 1]; };  // past the end: a valid entry that is never processed
    auto load_header = [&](const mcull_visible &en) {
        const mcull_meshlet *m = meshlets + en.meshletID;
        return Hdr{m->dataOffset, m->vertexCount, m->triangleCount};
    };
    auto load_vertex_id = [&](const Hdr &hd) { return lane < hd.vc ? meshlet_data[hd.off + lane] : 0u; };
    // the three local vertex ids of triangle t, packed into one word (0 for t beyond the meshlet's triangles)
    auto load_corners = [&](const Hdr &hd, uint32_t t) {
        if (t >= hd.tc) return 0u;
        const uint8_t *l = reinterpret_cast<const uint8_t *>(meshlet_data + hd.off + hd.vc) + 3u * t;
        return (uint32_t)l[0] | ((uint32_t)l[1] << 8) | ((uint32_t)l[2] << 16);
    };
    // Software pipeline over the dependent loads entry -> header -> vertex id -> vertex: the entry is
    // fetched three iterations ahead, the header two, the vertex id one, so that an iteration waits
    // for a single memory round trip (the vertex itself) instead of four.  The corner indices of the
    // triangles (two per lane: a meshlet has at most 124 <= 128) travel with the vertex id, and so does the
    // instance's matrix (instance_mvp_kernel): sixteen scalar registers.  (One stage more - the position one
    // iteration ahead - costs vector registers, drops a wave per SIMD and measured 7 % slower.)
    struct Mvp {
        float m[16];
    };
    auto load_mvp = [&](const mcull_visible &en) {
        const float *src = mvp_all + (size_t)__builtin_amdgcn_readfirstlane(en.drawID) * 16;
        Mvp r;
#pragma unroll
        for (int j = 0; j < 16; j++) r.m[j] = src[j];
        return r;
    };
    mcull_visible ent_a = load_entry(wave0), ent_b = load_entry(wave0 + n_waves), ent_c = load_entry(wave0 + 2 * n_waves);
    Hdr hdr_a = load_header(ent_a), hdr_b = load_header(ent_b);
    uint32_t gid_a = load_vertex_id(hdr_a);
    uint32_t cor0_a = load_corners(hdr_a, lane), cor1_a = load_corners(hdr_a, lane + 64u);
    Mvp mvp_a = load_mvp(ent_a);
    for (uint64_t e = wave0; e < n; e += n_waves) {
        const mcull_visible ent_d = load_entry(e + 3 * n_waves);
        const Hdr hdr_c = load_header(ent_c);
        const uint32_t gid_b = load_vertex_id(hdr_b);
        const uint32_t cor0_b = load_corners(hdr_b, lane), cor1_b = load_corners(hdr_b, lane + 64u);
        const Mvp cur = mvp_a;
        const float *mvp = cur.m;
        const uint32_t vc = hdr_a.vc, tc = hdr_a.tc, gid = gid_a, cor0 = cor0_a, cor1 = cor1_a;
        ent_a = ent_b; ent_b = ent_c; ent_c = ent_d;
        hdr_a = hdr_b; hdr_b = hdr_c;
        gid_a = gid_b; cor0_a = cor0_b; cor1_a = cor1_b;
        mvp_a = load_mvp(ent_a);
        // lane l: local vertex l -> clip and screen position
        ClipV cv = {0.0f, 0.0f, 0.0f, 0.0f};
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        int flags = 0;  // bit 0: w > 0 (projectable), bit 1: in front of the near plane
        if (lane < vc) {
            const mcull_vertex *v = vertices + gid;
            const float px = v->vx, py = v->vy, pz = v->vz;
            cv.x = ((mvp[0] * px + mvp[4] * py) + mvp[8] * pz) + mvp[12];
            cv.y = ((mvp[1] * px + mvp[5] * py) + mvp[9] * pz) + mvp[13];
            cv.z = ((mvp[2] * px + mvp[6] * py) + mvp[10] * pz) + mvp[14];
            cv.w = ((mvp[3] * px + mvp[7] * py) + mvp[11] * pz) + mvp[15];
            const ScreenV sv = to_screen(cv, rc.wf, rc.hf);
            sx = sv.x; sy = sv.y; sz = sv.z;
            flags = sv.ok | (((cv.w - cv.z) >= 0.0f) ? 2 : 0);
        }
        for (uint32_t base = 0; base < tc; base += 64u) {  // wave-uniform trip count (at most 2: the record's u8 holds <= 124)
            const uint32_t t = base + lane;
            const bool valid = t < tc;
            const uint32_t cor = base == 0u ? cor0 : cor1;
            const uint32_t i0 = cor & 0xffu, i1 = (cor >> 8) & 0xffu, i2 = cor >> 16;
            Tri tr;
            tr.x0 = __shfl(sx, (int)i0); tr.y0 = __shfl(sy, (int)i0); tr.z0 = __shfl(sz, (int)i0);
            tr.x1 = __shfl(sx, (int)i1); tr.y1 = __shfl(sy, (int)i1); tr.z1 = __shfl(sz, (int)i1);
            tr.x2 = __shfl(sx, (int)i2); tr.y2 = __shfl(sy, (int)i2); tr.z2 = __shfl(sz, (int)i2);
            const int f0 = __shfl(flags, (int)i0), f1 = __shfl(flags, (int)i1), f2 = __shfl(flags, (int)i2);
            int tri_ok = f0 & f1 & f2 & 1;
            const int in0 = f0 >> 1, in1 = f1 >> 1, in2 = f2 >> 1;
            const int n_in = in0 + in1 + in2;
            // n_in == 3: the triangle as it is; 0: nothing; 1 or 2: clipped to a triangle or a quad (= two
            // triangles fanned from the polygon's first vertex, polygon order as the edge walk 0->1->2->0 emits it)
            const bool needs_clip = valid && n_in > 0 && n_in < 3;
            Tri tr2 = tr;
            int tri2_ok = 0;
            bool second = false;
            if (__any(needs_clip)) {  // wave-uniform: the shuffles below need every lane
                ClipV v0, v1, v2;
                v0.x = __shfl(cv.x, (int)i0); v0.y = __shfl(cv.y, (int)i0); v0.z = __shfl(cv.z, (int)i0); v0.w = __shfl(cv.w, (int)i0);
                v1.x = __shfl(cv.x, (int)i1); v1.y = __shfl(cv.y, (int)i1); v1.z = __shfl(cv.z, (int)i1); v1.w = __shfl(cv.w, (int)i1);
                v2.x = __shfl(cv.x, (int)i2); v2.y = __shfl(cv.y, (int)i2); v2.z = __shfl(cv.z, (int)i2); v2.w = __shfl(cv.w, (int)i2);
                if (needs_clip) {
                    const ClipV e0 = in0 ? clip_intersect(v0, v1) : clip_intersect(v1, v0);  // edge 0-1 (if it crosses)
                    const ClipV e1 = in1 ? clip_intersect(v1, v2) : clip_intersect(v2, v1);  // edge 1-2
                    const ClipV e2 = in2 ? clip_intersect(v2, v0) : clip_intersect(v0, v2);  // edge 2-0
                    ClipV p0 = v0, p1 = v1, p2 = v2, p3 = v2;
                    switch (in0 | (in1 << 1) | (in2 << 2)) {
                        case 1: p0 = v0; p1 = e0; p2 = e2; break;
                        case 2: p0 = e0; p1 = v1; p2 = e1; break;
                        case 4: p0 = e1; p1 = v2; p2 = e2; break;
                        case 3: p0 = v0; p1 = v1; p2 = e1; p3 = e2; second = true; break;
                        case 6: p0 = e0; p1 = v1; p2 = v2; p3 = e2; second = true; break;
                        default: p0 = v0; p1 = e0; p2 = e1; p3 = v2; second = true; break;  // 5
                    }
                    const ScreenV s0 = to_screen(p0, rc.wf, rc.hf), s1 = to_screen(p1, rc.wf, rc.hf), s2 = to_screen(p2, rc.wf, rc.hf),
                                  s3 = to_screen(p3, rc.wf, rc.hf);
                    tr.x0 = s0.x; tr.y0 = s0.y; tr.z0 = s0.z; tr.x1 = s1.x; tr.y1 = s1.y; tr.z1 = s1.z;
                    tr.x2 = s2.x; tr.y2 = s2.y; tr.z2 = s2.z;
                    tri_ok = s0.ok & s1.ok & s2.ok;
                    tr2.x0 = s0.x; tr2.y0 = s0.y; tr2.z0 = s0.z; tr2.x1 = s2.x; tr2.y1 = s2.y; tr2.z1 = s2.z;
                    tr2.x2 = s3.x; tr2.y2 = s3.y; tr2.z2 = s3.z;
                    tri2_ok = s0.ok & s2.ok & s3.ok;
                }
            }
            const bool first = valid && (n_in == 3 || needs_clip);
            // rasterise one triangle per lane (wave-wide call: the large boxes use shuffles)
            auto raster_round = [&](const Tri &cur, bool cur_valid) {
                const float area = (cur.x1 - cur.x0) * (cur.y2 - cur.y0) - (cur.x2 - cur.x0) * (cur.y1 - cur.y0);
                bool draw = cur_valid && (area < 0.0f);  // front-facing, in front of the eye, not NaN
                const float minx = fminf(fminf(cur.x0, cur.x1), cur.x2), maxx = fmaxf(fmaxf(cur.x0, cur.x1), cur.x2);
                const float miny = fminf(fminf(cur.y0, cur.y1), cur.y2), maxy = fmaxf(fmaxf(cur.y0, cur.y1), cur.y2);
                const float x_lo = fmaxf(ceilf(minx - 0.5f), 0.0f), x_hi = fminf(floorf(maxx - 0.5f), (float)(rc.w - 1u));
                const float y_lo = fmaxf(ceilf(miny - 0.5f), 0.0f), y_hi = fminf(floorf(maxy - 0.5f), (float)(rc.h - 1u));
                draw = draw && (x_lo <= x_hi) && (y_lo <= y_hi);
                if (!__any(draw)) return;  // wave-uniform: nothing in this round covers a pixel centre
                uint32_t bx = 0, by = 0, bw = 0, bh = 0;
                if (draw) {
                    bx = (uint32_t)x_lo; by = (uint32_t)y_lo;
                    bw = (uint32_t)x_hi - bx + 1u; bh = (uint32_t)y_hi - by + 1u;
                }
                const uint64_t box = (uint64_t)bw * bh;
                const bool small = draw && box <= rc.small_box;
                if (small) {
                    const bool tl0 = edge_top_left(cur.x1, cur.y1, cur.x2, cur.y2), tl1 = edge_top_left(cur.x2, cur.y2, cur.x0, cur.y0),
                               tl2 = edge_top_left(cur.x0, cur.y0, cur.x1, cur.y1);
                    uint32_t xx = 0, yy = 0;  // row-major walk of the box without a division per pixel
                    for (uint32_t p = 0; p < (uint32_t)box; p++) {
                        raster_pixel(cur, area, tl0, tl1, tl2, bx + xx, by + yy, rc.w, depth_bits);
                        if (++xx == bw) { xx = 0; ++yy; }
                    }
                }
                // large boxes: the whole wave walks one triangle at a time
                uint64_t big = __ballot(draw && !small);
                while (big) {
                    const int src = __builtin_ctzll(big);
                    big &= big - 1ull;
                    Tri b;
                    b.x0 = __shfl(cur.x0, src); b.y0 = __shfl(cur.y0, src); b.z0 = __shfl(cur.z0, src);
                    b.x1 = __shfl(cur.x1, src); b.y1 = __shfl(cur.y1, src); b.z1 = __shfl(cur.z1, src);
                    b.x2 = __shfl(cur.x2, src); b.y2 = __shfl(cur.y2, src); b.z2 = __shfl(cur.z2, src);
                    const float b_area = __shfl(area, src);
                    const uint32_t ox = __shfl(bx, src), oy = __shfl(by, src), ow = __shfl(bw, src), oh = __shfl(bh, src);
                    if (ow * oh > rc.huge_box) {
                        // hand the box to the job kernel: one reservation by lane 0, the records written by the wave
                        const uint32_t ntx = (ow + kJobTile - 1u) / kJobTile, n_jobs = ntx * ((oh + kJobTile - 1u) / kJobTile);
                        uint32_t tri_idx = 0, job_base = 0;
                        if (lane == 0) {
                            tri_idx = atomicAdd(queue.counters, 1u);
                            job_base = atomicAdd(queue.counters + 1, n_jobs);
                        }
                        tri_idx = __builtin_amdgcn_readfirstlane(tri_idx);
                        job_base = __builtin_amdgcn_readfirstlane(job_base);
                        const bool fits = tri_idx < queue.tri_cap && job_base + n_jobs <= queue.job_cap && job_base + n_jobs >= job_base;
                        if (fits && lane == 0) {
                            HugeTri rec = {b.x0, b.y0, b.z0, b.x1, b.y1, b.z1, b.x2, b.y2, b.z2, ox | (oy << 16), (ow - 1u) | ((oh - 1u) << 16), 0u};
                            queue.tris[tri_idx] = rec;
                        }
                        // reserved job slots inside the queue are always written: real pieces, or "nothing to do" when
                        // the reservation does not fit as a whole (the box is then walked here, as a box below the limit is)
                        for (uint32_t j = lane; j < n_jobs && job_base + j < queue.job_cap && job_base + j >= job_base; j += 64u)
                            queue.jobs[job_base + j] = HugeJob{fits ? tri_idx : 0xffffffffu, (j % ntx) | ((j / ntx) << 16)};
                        if (fits) continue;
                    }
                    const bool t0 = edge_top_left(b.x1, b.y1, b.x2, b.y2), t1 = edge_top_left(b.x2, b.y2, b.x0, b.y0),
                               t2 = edge_top_left(b.x0, b.y0, b.x1, b.y1);
                    // the wave covers the box in 8x8-pixel tiles, lane = (lane & 7, lane >> 3) inside the tile
                    for (uint32_t ty = lane >> 3; ty < oh; ty += 8u)
                        for (uint32_t tx = lane & 7u; tx < ow; tx += 8u)
                            raster_pixel(b, b_area, t0, t1, t2, ox + tx, oy + ty, rc.w, depth_bits);
                }
            };
            raster_round(tr, first && tri_ok);
            if (__any(second)) raster_round(tr2, second && tri2_ok);  // wave-uniform: the second halves of clipped quads
        }
    }
}

// The queued pieces of huge boxes, one wave per piece.  A piece whose four corner pixel centres all lie outside the
// same edge holds no covered pixel - each edge function, as evaluated, is monotonic in x for a fixed y and in y for a
// fixed x (rounding preserves order), so over a rectangle of pixel centres its minimum is taken at a corner - and
// is skipped without touching its pixels.
__global__ __launch_bounds__(kRasterBlock) void depth_raster_jobs_kernel(HugeQueue queue, uint32_t w, uint32_t *__restrict__ depth_bits) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = blockIdx.x * (kRasterBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (kRasterBlock / 64);
    uint32_t n = queue.counters[1];
    if (n == 0u) return;  // the usual case: no huge box this call (then no triangle was reserved either, nothing to re-zero)
    if (n > queue.job_cap) n = queue.job_cap;
    for (uint32_t j = wave0; j < n; j += n_waves) {
        const HugeJob job = queue.jobs[j];
        if (job.tri == 0xffffffffu) continue;
        const HugeTri t = queue.tris[job.tri];
        Tri b = {t.x0, t.y0, t.z0, t.x1, t.y1, t.z1, t.x2, t.y2, t.z2};
        const float area = (b.x1 - b.x0) * (b.y2 - b.y0) - (b.x2 - b.x0) * (b.y1 - b.y0);
        const bool t0 = edge_top_left(b.x1, b.y1, b.x2, b.y2), t1 = edge_top_left(b.x2, b.y2, b.x0, b.y0),
                   t2 = edge_top_left(b.x0, b.y0, b.x1, b.y1);
        const uint32_t ox = (t.origin & 0xffffu) + kJobTile * (job.piece & 0xffffu), oy = (t.origin >> 16) + kJobTile * (job.piece >> 16);
        const uint32_t ex = (t.origin & 0xffffu) + (t.extent & 0xffffu), ey = (t.origin >> 16) + (t.extent >> 16);  // last pixel of the box
        const uint32_t ow = min(kJobTile, ex - ox + 1u), oh = min(kJobTile, ey - oy + 1u);
        // corner test (lanes 0-3 take one corner each)
        {
            const float cx = (float)(ox + ((lane & 1u) ? ow - 1u : 0u)) + 0.5f, cy = (float)(oy + ((lane & 2u) ? oh - 1u : 0u)) + 0.5f;
            const float w0 = edge_fn(b.x1, b.y1, b.x2, b.y2, cx, cy), w1 = edge_fn(b.x2, b.y2, b.x0, b.y0, cx, cy),
                        w2 = edge_fn(b.x0, b.y0, b.x1, b.y1, cx, cy);
            const bool corner = lane < 4u;
            const bool out0 = !(w0 < 0.0f || (w0 == 0.0f && t0)), out1 = !(w1 < 0.0f || (w1 == 0.0f && t1)),
                       out2 = !(w2 < 0.0f || (w2 == 0.0f && t2));
            const uint64_t four = 0xfull;
            if ((__ballot(corner && out0) & four) == four || (__ballot(corner && out1) & four) == four ||
                (__ballot(corner && out2) & four) == four)
                continue;
        }
        for (uint32_t ty = lane >> 3; ty < oh; ty += 8u)
            for (uint32_t tx = lane & 7u; tx < ow; tx += 8u)
                raster_pixel(b, area, t0, t1, t2, ox + tx, oy + ty, w, depth_bits);
    }
    // every workgroup has read the job count before it gets here; the last one leaves the counters zero for the next call
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(queue.counters + 2, 1u) == gridDim.x - 1u) {
        queue.counters[0] = 0u;
        queue.counters[1] = 0u;
        queue.counters[2] = 0u;
    }
}

size_t depth_raster_scratch_bytes() { return 256 + sizeof(HugeTri) * kQueueTriangles + sizeof(HugeJob) * kQueueJobs; }

hipError_t launch_depth_raster(const mcull_visible *list, const uint64_t *total, uint64_t capacity,
                               const mcull_meshlet *meshlets, const uint32_t *meshlet_data, const mcull_vertex *vertices,
                               const float *transforms, uint64_t n_instances, float *mvp, const float vp[16], uint32_t w,
                               uint32_t h, float *depth, bool clear, void *scratch, hipStream_t st, const ViewConst *view_dev) {
    // mvp: 64 bytes per instance, written here for every instance of the scene (the list's drawIDs index it);
    // clear: the depth image (w * h floats, 16-byte aligned) is zeroed first
    // scratch: depth_raster_scratch_bytes() of device memory (counters - zero before the first call -, the triangle
    // table, the job queue)
    HugeQueue queue;
    queue.counters = reinterpret_cast<uint32_t *>(scratch);
    queue.tris = reinterpret_cast<HugeTri *>(reinterpret_cast<uint8_t *>(scratch) + 256);
    queue.jobs = reinterpret_cast<HugeJob *>(reinterpret_cast<uint8_t *>(scratch) + 256 + sizeof(HugeTri) * kQueueTriangles);
    // evaluation switch MCULL_RASTER_QUEUE="<triangles>,<jobs>": a smaller queue, so that tests reach the path taken
    // when it is full (the image does not depend on it)
    static const uint2 caps = [] {
        uint2 c = make_uint2(kQueueTriangles, kQueueJobs);
        if (const char *e = std::getenv("MCULL_RASTER_QUEUE")) {
            char *end = nullptr;
            const unsigned long t = std::strtoul(e, &end, 10);
            const unsigned long j = (end && *end == ',') ? std::strtoul(end + 1, nullptr, 10) : kQueueJobs;
            c.x = (uint32_t)(t < kQueueTriangles ? t : kQueueTriangles);
            c.y = (uint32_t)(j < kQueueJobs ? j : kQueueJobs);
        }
        return c;
    }();
    queue.tri_cap = caps.x;
    queue.job_cap = caps.y;
    hipError_t e;
    RasterConst rc;
    for (int i = 0; i < 16; i++) rc.vp[i] = vp ? vp[i] : 0.0f;
    rc.w = w; rc.h = h;
    rc.wf = (float)w; rc.hf = (float)h;
    rc.small_box = kSmallBox;
    static const uint32_t huge_box = [] {  // evaluation switch: the box size from which triangles are queued
        const char *e = std::getenv("MCULL_RASTER_HUGE");
        return e ? (uint32_t)std::strtoul(e, nullptr, 10) : kHugeBox;
    }();
    rc.huge_box = huge_box;
    // The entry count lives on the device: a fixed grid whose waves stride over the list.  The grid is
    // exactly what the device holds at once (a larger one would run as a full batch followed by a
    // partial one, each wave of either doing the same share of the list).
    static const uint32_t grid = [] {
        int dev = 0, cus = 256, per_cu = 4;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, depth_raster_kernel, kRasterBlock, 0) != hipSuccess || per_cu < 1)
            per_cu = 4;
        const char *e = std::getenv("MCULL_RASTER_GRID");  // evaluation switch: workgroups in the grid (the image does not depend on it)
        return e ? (uint32_t)std::strtoul(e, nullptr, 10) : (uint32_t)(cus * per_cu);
    }();
    const uint64_t n_elements = n_instances * 16, n_pixels = clear ? (uint64_t)w * h : 0;
    if (n_elements || n_pixels) {
        const uint64_t want = ((n_elements > n_pixels / 16 ? n_elements : n_pixels / 16) + kRasterBlock - 1) / kRasterBlock;  // four 16-byte stores per thread
        const dim3 prep_grid((uint32_t)(want < 1 ? 1 : want > 8192 ? 8192 : want));
        if (view_dev)
            hipLaunchKernelGGL(raster_prepare_kernel<true>, prep_grid, dim3(kRasterBlock), 0, st, transforms, n_elements, rc, view_dev, mvp,
                               reinterpret_cast<uint32_t *>(depth), n_pixels);
        else
            hipLaunchKernelGGL(raster_prepare_kernel<false>, prep_grid, dim3(kRasterBlock), 0, st, transforms, n_elements, rc,
                               (const ViewConst *)nullptr, mvp, reinterpret_cast<uint32_t *>(depth), n_pixels);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(depth_raster_kernel, dim3(grid), dim3(kRasterBlock), 0, st, list, (const unsigned long long *)total, capacity,
                       meshlets, meshlet_data, vertices, (const float *)mvp, rc, reinterpret_cast<uint32_t *>(depth), queue);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // a fixed grid of an eighth of the main one: an empty queue (the usual case) costs the same 4.5 us of launch
    // whatever the grid, and a twentieth was slower on a screen-filling quad (0.127 against 0.082 ms at 3840x2160)
    hipLaunchKernelGGL(depth_raster_jobs_kernel, dim3(grid / 8u ? grid / 8u : 1u), dim3(kRasterBlock), 0, st, queue, w,
                       reinterpret_cast<uint32_t *>(depth));
    return hipGetLastError();
}

}  // namespace mcull
