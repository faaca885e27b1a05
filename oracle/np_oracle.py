"""Independent numpy (float32) restatement of the culling path, used ONLY to cross-check the C
oracle (oracle/mcull_oracle.c): written from the reference's shader text and Engine.cpp separately,
vectorised over meshlets instead of scalar loops, so that a slip in one restatement shows up as a
disagreement.  TEST INFRASTRUCTURE; PARITY UNPINNED for what this file restates (the reference
holds no golden vectors for predicates, pyramid or bounds).

numpy float32 arithmetic is IEEE binary32 with one rounding per operation (no fused multiply-add),
which is the arithmetic the oracle and the kernels are defined in.
"""
import numpy as np

f32 = np.float32


def _dot3(ax, ay, az, bx, by, bz):
    return (ax * bx + ay * by) + az * bz


def mat_mul(a, b):
    """Column-major 4x4 product, column c = ((A0*b0 + A1*b1) + A2*b2) + A3*b3."""
    a = np.asarray(a, dtype=f32).reshape(4, 4)  # [col][row]
    b = np.asarray(b, dtype=f32).reshape(4, 4)
    out = np.zeros((4, 4), dtype=f32)
    for c in range(4):
        out[c] = ((a[0] * b[c][0] + a[1] * b[c][1]) + a[2] * b[c][2]) + a[3] * b[c][3]
    return out.reshape(16)


def camera_position(view):
    """inverse(view)[3].xyz by cofactors (shader.task:88-89): inv(r,3) = cof(3,r) / det."""
    m = np.asarray(view, dtype=f32).reshape(4, 4).T  # m[row][col]

    def minor(i, j):
        rows = [r for r in range(4) if r != i]
        cols = [c for c in range(4) if c != j]
        e = [m[r][c] for r in rows for c in cols]
        return f32(f32(e[0] * f32(f32(e[4] * e[8]) - f32(e[5] * e[7]))) - f32(e[1] * f32(f32(e[3] * e[8]) - f32(e[5] * e[6])))) + \
            f32(e[2] * f32(f32(e[3] * e[7]) - f32(e[4] * e[6])))

    cof = [[(-minor(i, j) if (i + j) & 1 else minor(i, j)) for j in range(4)] for i in range(4)]
    det = f32(f32(f32(m[0][0] * cof[0][0]) + f32(m[0][1] * cof[0][1])) + f32(m[0][2] * cof[0][2])) + f32(m[0][3] * cof[0][3])
    inv_det = f32(1.0) / det
    return np.array([cof[3][r] * inv_det for r in range(3)], dtype=f32)


def hiz_levels(depth):
    """shader.comp:12-31 level by level (Engine.cpp:1840-1847); min(x, y) = y < x ? y : x."""
    d = np.ascontiguousarray(depth, dtype=f32)
    h, w = d.shape
    n = int(np.floor(np.log2(max(w, h)))) + 1

    def gmin(x, y):
        return np.where(y < x, y, x)

    out = [d.copy()]
    for i in range(1, n):
        src = out[-1]
        sh, sw = src.shape
        dh, dw = max(1, h >> i), max(1, w >> i)
        ys0 = np.minimum(2 * np.arange(dh), sh - 1); ys1 = np.minimum(2 * np.arange(dh) + 1, sh - 1)
        xs0 = np.minimum(2 * np.arange(dw), sw - 1); xs1 = np.minimum(2 * np.arange(dw) + 1, sw - 1)
        d0 = src[np.ix_(ys0, xs0)]; d1 = src[np.ix_(ys0, xs1)]; d2 = src[np.ix_(ys1, xs0)]; d3 = src[np.ix_(ys1, xs1)]
        out.append(gmin(gmin(d0, d1), gmin(d2, d3)).astype(f32))
    return out


def mip_level(px, levels):
    """clamp(ceil(log2(max(px,1))), 0, levels-1), exact for float32 inputs (float64 log2 of a float32
    is far more accurate than the spacing of float32 values around powers of two)."""
    px = np.asarray(px, dtype=f32)
    x = np.where(px > 1, px, f32(1)).astype(np.float64)
    with np.errstate(over="ignore", invalid="ignore"):
        l = np.ceil(np.log2(x))
    l = np.where(np.isfinite(l), l, 255)
    return np.clip(l, 0, levels - 1).astype(np.int64)


def cull_instance(g, model, cone, sphere, tests=7, pyramid=None):
    """shader.task:85-101 for all meshlets of one instance.  cone/sphere: (N,4) float32.
    Returns (visible bool[N], dict of margins)."""
    g = np.asarray(g).reshape(-1)[0]
    m = np.asarray(model, dtype=f32).reshape(4, 4)  # [col][row]
    cx, cy, cz, cr = (np.asarray(sphere[:, k], dtype=f32) for k in range(4))
    ax, ay, az, cw_ = (np.asarray(cone[:, k], dtype=f32) for k in range(4))
    wx = ((m[0][0] * cx + m[1][0] * cy) + m[2][0] * cz) + m[3][0]
    wy = ((m[0][1] * cx + m[1][1] * cy) + m[2][1] * cz) + m[3][1]
    wz = ((m[0][2] * cx + m[1][2] * cy) + m[2][2] * cz) + m[3][2]
    r = cr * np.sqrt(_dot3(m[0][0], m[0][1], m[0][2], m[0][0], m[0][1], m[0][2]), dtype=f32)
    with np.errstate(all="ignore"):
        fr = np.asarray(g["frustum"], dtype=f32)
        fculled = np.zeros(len(cx), dtype=bool)
        fmargin = np.full(len(cx), np.inf, dtype=f32)
        for i in range(5):
            d = (_dot3(fr[i][0], fr[i][1], fr[i][2], wx, wy, wz) + fr[i][3]) + r
            fculled |= d < 0
            fmargin = np.where(d < fmargin, d, fmargin)
        oculled = np.zeros(len(cx), dtype=bool)
        omargin = np.zeros(len(cx), dtype=f32)
        if pyramid is not None:
            levels = pyramid
            vp = mat_mul(g["proj"], g["view"]).reshape(4, 4)
            clip = [((vp[0][k] * wx + vp[1][k] * wy) + vp[2][k] * wz) + vp[3][k] for k in range(4)]
            nx, ny, nz = clip[0] / clip[3], clip[1] / clip[3], clip[2] / clip[3]
            nr = (np.asarray(g["proj"], dtype=f32)[5] * r) / clip[3]
            u = nx * f32(0.5) + f32(0.5)
            v = f32(1.0) - (ny * f32(0.5) + f32(0.5))
            h0, w0 = levels[0].shape
            px = (nr * f32(2.0)) * f32(max(w0, h0))
            mip = mip_level(px, len(levels))
            d = np.zeros(len(cx), dtype=f32)
            for L in np.unique(mip):
                sel = mip == L
                lh, lw = levels[L].shape
                fx = np.floor(u[sel] * f32(lw)); fy = np.floor(v[sel] * f32(lh))
                okx = np.abs(fx) < f32(1.0e9); oky = np.abs(fy) < f32(1.0e9)
                ix = np.where(okx, np.mod(np.where(okx, fx, 0).astype(np.int64), lw), 0)
                iy = np.where(oky, np.mod(np.where(oky, fy, 0).astype(np.int64), lh), 0)
                d[sel] = levels[L][iy, ix]
            thr = d - f32(0.05)
            oculled = nz < thr
            omargin = nz - thr
        nax = (m[0][0] * ax + m[1][0] * ay) + m[2][0] * az
        nay = (m[0][1] * ax + m[1][1] * ay) + m[2][1] * az
        naz = (m[0][2] * ax + m[1][2] * ay) + m[2][2] * az
        ia = f32(1.0) / np.sqrt(_dot3(nax, nay, naz, nax, nay, naz), dtype=f32)
        nax, nay, naz = nax * ia, nay * ia, naz * ia
        cam = camera_position(g["view"])
        vx, vy, vz = cam[0] - wx, cam[1] - wy, cam[2] - wz
        iv = f32(1.0) / np.sqrt(_dot3(vx, vy, vz, vx, vy, vz), dtype=f32)
        vx, vy, vz = vx * iv, vy * iv, vz * iv
        dp = _dot3(nax, nay, naz, vx, vy, vz)
        cculled = dp < -cw_
        cmargin = dp + cw_
    culled = np.zeros(len(cx), dtype=bool)
    if tests & 1:
        culled |= fculled
    if tests & 4:
        culled |= oculled
    if tests & 2:
        culled |= cculled
    return ~culled, dict(frustum=fmargin, occlusion=omargin, cone=cmargin,
                         culled=(fculled, oculled, cculled))


def meshlet_cone(vertices_pos, meshlet_data, data_offset, vertex_count, triangle_count):
    """buildMeshletCon (Engine.cpp:472-547): returns (cone[4], apex[3], radius)."""
    g = np.asarray(meshlet_data[data_offset: data_offset + vertex_count], dtype=np.int64)
    words = meshlet_data[data_offset + vertex_count: data_offset + vertex_count + (triangle_count * 3 + 3) // 4]
    loc = np.ascontiguousarray(words, dtype="<u4").view("u1")[: triangle_count * 3].reshape(-1, 3).astype(np.int64)
    p = np.asarray(vertices_pos, dtype=f32)[g]  # (vc, 3)
    p0, p1, p2 = p[loc[:, 0]], p[loc[:, 1]], p[loc[:, 2]]
    a, b = p1 - p0, p2 - p0
    with np.errstate(all="ignore"):
        cx = a[:, 1] * b[:, 2] - b[:, 1] * a[:, 2]
        cy = a[:, 2] * b[:, 0] - b[:, 2] * a[:, 0]
        cz = a[:, 0] * b[:, 1] - b[:, 0] * a[:, 1]
        inv = f32(1.0) / np.sqrt(_dot3(cx, cy, cz, cx, cy, cz), dtype=f32)
        n = np.stack([cx * inv, cy * inv, cz * inv], axis=1).astype(f32)
        s = np.zeros(3, dtype=f32)
        for t in range(triangle_count):  # sequential float sum, triangle order
            s = s + n[t]
        axis = s * (f32(1.0) / np.sqrt(_dot3(s[0], s[1], s[2], s[0], s[1], s[2]), dtype=f32))
        dp = (n[:, 0] * axis[0] + n[:, 1] * axis[1]) + n[:, 2] * axis[2]
        half = f32(1.0)
        for t in range(triangle_count):
            if dp[t] < half:
                half = dp[t]
        cone_w = f32(1.0) if half < 0 else np.sqrt(f32(1) - half * half, dtype=f32)
        apex = np.zeros(3, dtype=f32)
        fc = f32(vertex_count)
        for i in range(vertex_count):
            apex = apex + p[i] / fc
        d = p - apex
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        mx = f32(0.0)
        for i in range(vertex_count):
            if mx < d2[i]:
                mx = d2[i]
    return np.array([axis[0], axis[1], axis[2], cone_w], dtype=f32), apex, np.sqrt(mx, dtype=f32)


# ---------------------------------------------------------------------------------------------
# depth pre-pass stand-in (Engine.cpp:2397-2441, shader.mesh:60-101), written separately from the C
# oracle: one triangle at a time, pixels of its box evaluated as arrays.

def _to_screen(c, wf, hf):
    with np.errstate(all="ignore"):
        nx, ny, z = c[0] / c[3], c[1] / c[3], c[2] / c[3]
        return (nx * f32(0.5) + f32(0.5)) * wf, (ny * f32(0.5) + f32(0.5)) * hf, z, bool(c[3] > 0)


def _isect(p, q):
    """Edge p -> q against the near plane z <= w, p inside: p + t (q - p), t = d_p / (d_p - d_q)."""
    with np.errstate(all="ignore"):
        dp, dq = p[3] - p[2], q[3] - q[2]
        t = dp / (dp - dq)
        return p + t * (q - p)


def _raster(depth_bits, a, b, c, w, h):
    (ax, ay, az, aok), (bx, by, bz, bok), (cx, cy, cz, cok) = a, b, c
    if not (aok and bok and cok):
        return
    with np.errstate(all="ignore"):
        area = (bx - ax) * (cy - ay) - (cx - ax) * (by - ay)
        if not area < 0:
            return
        xs, ys = np.array([ax, bx, cx], dtype=f32), np.array([ay, by, cy], dtype=f32)
        if np.isnan(xs).any() or np.isnan(ys).any():
            lo_x, hi_x, lo_y, hi_y = f32(0), f32(w - 1), f32(0), f32(h - 1)  # fmin/fmax ignore NaN operands
            fx, fy = xs[~np.isnan(xs)], ys[~np.isnan(ys)]
            if len(fx):
                lo_x, hi_x = max(np.ceil(fx.min() - f32(0.5)), f32(0)), min(np.floor(fx.max() - f32(0.5)), f32(w - 1))
            if len(fy):
                lo_y, hi_y = max(np.ceil(fy.min() - f32(0.5)), f32(0)), min(np.floor(fy.max() - f32(0.5)), f32(h - 1))
        else:
            lo_x, hi_x = max(np.ceil(xs.min() - f32(0.5)), f32(0)), min(np.floor(xs.max() - f32(0.5)), f32(w - 1))
            lo_y, hi_y = max(np.ceil(ys.min() - f32(0.5)), f32(0)), min(np.floor(ys.max() - f32(0.5)), f32(h - 1))
        if not (lo_x <= hi_x and lo_y <= hi_y):
            return
        px = np.arange(int(lo_x), int(hi_x) + 1, dtype=np.int64)
        py = np.arange(int(lo_y), int(hi_y) + 1, dtype=np.int64)
        X = (px.astype(f32) + f32(0.5))[None, :]
        Y = (py.astype(f32) + f32(0.5))[:, None]

        def edge(x0, y0, x1, y1):
            return (x1 - x0) * (Y - y0) - (y1 - y0) * (X - x0)

        def top_left(x0, y0, x1, y1):
            dx, dy = x1 - x0, y1 - y0
            return bool(dy > 0 or (dy == 0 and dx < 0))

        w0, w1, w2 = edge(bx, by, cx, cy), edge(cx, cy, ax, ay), edge(ax, ay, bx, by)
        inside = ((w0 < 0) | ((w0 == 0) & top_left(bx, by, cx, cy))) & ((w1 < 0) | ((w1 == 0) & top_left(cx, cy, ax, ay))) & \
                 ((w2 < 0) | ((w2 == 0) & top_left(ax, ay, bx, by)))
        z = ((w0 * az + w1 * bz) + w2 * cz) / area
        keep = inside & (z >= 0) & (z <= 1)
        bits = z.astype(f32).view(np.uint32) & np.uint32(0x7FFFFFFF)
        sub = depth_bits[py[0]:py[-1] + 1, px[0]:px[-1] + 1]
        np.maximum(sub, np.where(keep, bits, np.uint32(0)), out=sub)


def rasterize_depth(vertices_pos, meshlet_data, meshlets, transforms, visible, view, proj, w, h):
    """vertices_pos (n,3) f32; meshlets: structured array with dataOffset/vertexCount/triangleCount;
    transforms (n,16); visible: (drawID, meshletID) pairs.  Returns the (h, w) float32 image."""
    vp = mat_mul(proj, view)
    depth_bits = np.zeros((h, w), dtype=np.uint32)
    wf, hf = f32(w), f32(h)
    for draw, mid in visible:
        m = meshlets[int(mid)]
        off, vc, tc = int(m["dataOffset"]), int(m["vertexCount"]), int(m["triangleCount"])
        mvp = mat_mul(vp, transforms[int(draw)]).reshape(4, 4)  # [col][row]
        pos = vertices_pos[meshlet_data[off:off + vc]]
        clip = ((mvp[0][None, :] * pos[:, 0:1] + mvp[1][None, :] * pos[:, 1:2]) + mvp[2][None, :] * pos[:, 2:3]) + mvp[3][None, :]
        local = meshlet_data[off + vc: off + vc + (3 * tc + 3) // 4].view(np.uint8)[: 3 * tc].reshape(tc, 3)
        for tri in local:
            v = [clip[int(i)] for i in tri]
            inn = [bool((p[3] - p[2]) >= 0) for p in v]
            if not any(inn):
                continue
            if all(inn):
                poly = v
            else:
                poly = []
                for k in range(3):
                    q = (k + 1) % 3
                    if inn[k]:
                        poly.append(v[k])
                    if inn[k] != inn[q]:
                        poly.append(_isect(v[k], v[q]) if inn[k] else _isect(v[q], v[k]))
            s = [_to_screen(p, wf, hf) for p in poly]
            _raster(depth_bits, s[0], s[1], s[2], w, h)
            if len(s) == 4:
                _raster(depth_bits, s[0], s[2], s[3], w, h)
    return depth_bits.view(f32)
