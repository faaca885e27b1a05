"""Host-side input builders mirroring the reference's Engine methods for the culling path.

These produce the buffers the path consumes, in the reference's layouts, without touching the
GPU: createGlobals (Engine.cpp:2323-2352), updateTransforms (Engine.cpp:2353-2379),
createIndirectBuffer (Engine.cpp:1466-1484), plus the synthetic scenes the benchmark runs on.
All arithmetic is IEEE binary32 (numpy float32 scalars, no fused multiply-add) in the same
association order as glm evaluates it.
"""
import numpy as np

from . import CMD, DRAW_INDIRECT, GLOBALS, TRANSFORM

f32 = np.float32
_RAD = f32(0.01745329251994329576923690768489)  # glm::radians factor


def _dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _normalize(a):
    inv = f32(f32(1.0) / np.sqrt(_dot(a, a), dtype=f32))
    return np.array([a[0] * inv, a[1] * inv, a[2] * inv], dtype=f32)


def _cross(a, b):
    return np.array(
        [f32(a[1] * b[2]) - f32(b[1] * a[2]), f32(a[2] * b[0]) - f32(b[2] * a[0]), f32(a[0] * b[1]) - f32(b[0] * a[1])],
        dtype=f32,
    )


def mat4_mul(a, b):
    """glm mat4 * mat4 on column-major float32[16]: ((A0*b0 + A1*b1) + A2*b2) + A3*b3 per column."""
    a = np.asarray(a, dtype=f32).reshape(4, 4)  # a[c][r]
    b = np.asarray(b, dtype=f32).reshape(4, 4)
    out = np.zeros((4, 4), dtype=f32)
    for c in range(4):
        for r in range(4):
            out[c][r] = f32(f32(f32(a[0][r] * b[c][0]) + f32(a[1][r] * b[c][1])) + f32(a[2][r] * b[c][2])) + f32(a[3][r] * b[c][3])
    return out.reshape(16)


def extract_frustum(g):
    """Gribb-Hartmann planes of proj*view, each divided by |xyz| (Engine.cpp:2337-2349)."""
    vp = mat4_mul(g["proj"][0], g["view"][0]).reshape(4, 4)  # vp[c][r]
    fr = np.zeros((5, 4), dtype=f32)
    for c in range(4):
        fr[0][c] = vp[c][3] + vp[c][0]
        fr[1][c] = vp[c][3] - vp[c][0]
        fr[2][c] = vp[c][3] + vp[c][1]
        fr[3][c] = vp[c][3] - vp[c][1]
        fr[4][c] = vp[c][3] + vp[c][2]
    for i in range(5):
        ln = np.sqrt(_dot(fr[i][:3], fr[i][:3]), dtype=f32)
        for c in range(4):
            fr[i][c] = f32(fr[i][c] / ln)
    g["frustum"][0] = fr
    return g


def create_globals(eye=(2.0, 2.0, 2.0), centre=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0), fov_deg=45.0, z_near=0.1,
                   width=800, height=600, flip_y=True, meshlet_count=0):
    """Engine::createGlobals: lookAtRH view, infinite-far reverse-Z projection, 5 frustum planes."""
    g = np.zeros(1, dtype=GLOBALS)
    e = np.array(eye, dtype=f32)
    f = _normalize(np.array(centre, dtype=f32) - e)
    s = _normalize(_cross(f, np.array(up, dtype=f32)))
    u = _cross(s, f)
    view = np.zeros((4, 4), dtype=f32)  # view[c][r]
    for c in range(3):
        view[c][0] = s[c]
        view[c][1] = u[c]
        view[c][2] = -f[c]
    view[3][0] = -_dot(s, e)
    view[3][1] = -_dot(u, e)
    view[3][2] = _dot(f, e)
    view[3][3] = 1.0
    fl = f32(f32(1.0) / np.tan(f32(f32(f32(fov_deg) * _RAD) / f32(2.0)), dtype=f32))
    proj = np.zeros((4, 4), dtype=f32)
    proj[0][0] = f32(fl / f32(f32(width) / f32(height)))
    proj[1][1] = -fl if flip_y else fl
    proj[2][3] = -1.0
    proj[3][2] = z_near
    g["view"][0] = view.reshape(16)
    g["proj"][0] = proj.reshape(16)
    g["meshletCount"][0] = meshlet_count
    return extract_frustum(g)


class GlibcRand:
    """glibc's rand() (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31]) after srand(seed)."""

    def __init__(self, seed):
        r = [0] * 344
        r[0] = seed if seed != 0 else 1
        for i in range(1, 31):
            hi, lo = divmod(r[i - 1], 127773)
            w = 16807 * lo - 2836 * hi
            if w < 0:
                w += 2147483647
            r[i] = w
        for i in range(31, 34):
            r[i] = r[i - 31]
        for i in range(34, 344):
            r[i] = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
        self._r = r[-31:]  # last 31 values; next value uses r[-31] and r[-3]

    def rand(self):
        v = (self._r[0] + self._r[28]) & 0xFFFFFFFF
        self._r.pop(0)
        self._r.append(v)
        return v >> 1


def _rotate(angle, axis):
    """glm::rotate(mat4(1), angle, axis) as column-major [c][r]."""
    c, s = np.cos(angle, dtype=f32), np.sin(angle, dtype=f32)
    ax = _normalize(np.array(axis, dtype=f32))
    t = np.array([f32(f32(1.0) - c) * ax[0], f32(f32(1.0) - c) * ax[1], f32(f32(1.0) - c) * ax[2]], dtype=f32)
    R = np.zeros((3, 3), dtype=f32)
    R[0][0] = c + f32(t[0] * ax[0]); R[0][1] = f32(t[0] * ax[1]) + f32(s * ax[2]); R[0][2] = f32(t[0] * ax[2]) - f32(s * ax[1])
    R[1][0] = f32(t[1] * ax[0]) - f32(s * ax[2]); R[1][1] = c + f32(t[1] * ax[1]); R[1][2] = f32(t[1] * ax[2]) + f32(s * ax[0])
    R[2][0] = f32(t[2] * ax[0]) + f32(s * ax[1]); R[2][1] = f32(t[2] * ax[1]) - f32(s * ax[0]); R[2][2] = c + f32(t[2] * ax[2])
    ident = np.eye(4, dtype=f32)
    out = np.eye(4, dtype=f32)
    for k in range(3):
        for r in range(4):
            out[k][r] = f32(f32(ident[0][r] * R[k][0]) + f32(ident[1][r] * R[k][1])) + f32(ident[2][r] * R[k][2])
    return out.reshape(16)


def update_transforms(draw_count=1000, seed=42):
    """Engine::updateTransforms: srand(42); T * Rx * Ry * Rz * Rx(90 deg) * S(1) per instance."""
    rng = GlibcRand(seed)
    rand_max = f32(2147483647)  # RAND_MAX converted to float
    out = np.zeros(draw_count, dtype=TRANSFORM)
    for i in range(draw_count):
        u = [f32(f32(rng.rand()) / rand_max) for _ in range(6)]
        ox, oy, oz = (f32(f32(u[k] * f32(15.0)) - f32(7.0)) for k in range(3))
        rx = f32(u[3] * f32(f32(30.0) * _RAD))
        ry = f32(u[4] * f32(f32(45.0) * _RAD))
        rz = f32(u[5] * f32(f32(180.0) * _RAD))
        model = np.eye(4, dtype=f32)
        ident = np.eye(4, dtype=f32)
        for r in range(4):
            model[3][r] = f32(f32(f32(ident[0][r] * ox) + f32(ident[1][r] * oy)) + f32(ident[2][r] * oz)) + ident[3][r]
        model = model.reshape(16)
        model = mat4_mul(model, _rotate(rx, (1, 0, 0)))
        model = mat4_mul(model, _rotate(ry, (0, 1, 0)))
        model = mat4_mul(model, _rotate(rz, (0, 0, 1)))
        model = mat4_mul(model, _rotate(f32(f32(90.0) * _RAD), (1, 0, 0)))
        model = mat4_mul(model, np.eye(4, dtype=f32).reshape(16))
        out["model"][i] = model
    return out


def create_indirect_buffer(draw_count, meshlet_count, index_count):
    """Engine::createIndirectBuffer: the static commands the reference uploads once."""
    d = np.zeros(draw_count, dtype=DRAW_INDIRECT)
    d["indexCount"] = index_count
    d["instanceCount"] = 1
    d["commandMeshIndirect"]["taskCount"] = (meshlet_count + 31) // 32
    return d


# ------------------------------------------------------------------------------- synthetic inputs

def synthetic_depth(width, height, n_discs=64, seed=1234):
    """Reverse-Z depth image (1 = near, 0 = cleared far): max over random discs on a 0 background."""
    rs = np.random.RandomState(seed)
    d = np.zeros((height, width), dtype=f32)
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(n_discs):
        cx, cy = rs.uniform(0, width), rs.uniform(0, height)
        rad = rs.uniform(0.02, 0.15) * max(width, height)
        val = f32(rs.uniform(0.005, 0.2))
        m = (xx - cx) ** 2 + (yy - cy) ** 2 <= rad * rad
        d[m] = np.maximum(d[m], val)
    return d


def synthetic_bounds(n_meshlets, seed=42):
    """Bounds SoA of the synthetic scene: centres uniform in the unit cube, radius U(0.01, 0.05),
    axis uniform on the sphere, coneW U(0,1) with 10 % forced to 1.0 (never cone-culled)."""
    rs = np.random.RandomState(seed)
    cone = np.zeros((n_meshlets, 4), dtype=f32)
    v = rs.normal(size=(n_meshlets, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    cone[:, :3] = v
    cone[:, 3] = rs.uniform(0, 1, n_meshlets)
    cone[rs.uniform(size=n_meshlets) < 0.1, 3] = 1.0
    sphere = np.zeros((n_meshlets, 4), dtype=f32)
    sphere[:, :3] = rs.uniform(0, 1, (n_meshlets, 3))
    sphere[:, 3] = rs.uniform(0.01, 0.05, n_meshlets)
    return cone, sphere


def synthetic_transforms(n_instances, extent=50.0, seed=43):
    """Random rigid transforms: translation U(-extent, extent)^3, uniformly random rotation, scale 1."""
    rs = np.random.RandomState(seed)
    q = rs.normal(size=(n_instances, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.empty((n_instances, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    m = np.zeros((n_instances, 4, 4), dtype=f32)  # [c][r]
    m[:, :3, :3] = np.transpose(R, (0, 2, 1))
    m[:, 3, :3] = rs.uniform(-extent, extent, (n_instances, 3))
    m[:, 3, 3] = 1.0
    out = np.zeros(n_instances, dtype=TRANSFORM)
    out["model"] = m.reshape(n_instances, 16)
    return out


def proxy_mesh(nu, nv, seed=1, bumps=6.0):
    """Procedural closed surface (a bumpy torus) with 2*nu*nv triangles: the stand-in for the
    reference's dragon.obj / buddha.obj, which are missing from the reference snapshot.  It is NOT
    one of the Stanford scans; only the triangle count is matched (660x660 -> 871 200 triangles,
    680x800 -> 1 088 000).  Returns (positions float32 (n,3), triangle vertex ids int32 (t,3))."""
    rs = np.random.RandomState(seed)
    ph = rs.uniform(0, 2 * np.pi, 3)
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    U, V = np.meshgrid(u, v, indexing="ij")
    r = 0.35 + 0.05 * np.sin(bumps * U + ph[0]) * np.cos((bumps + 1) * V + ph[1]) + 0.02 * np.sin(17 * U + 3 * V + ph[2])
    x = (1 + r * np.cos(V)) * np.cos(U)
    y = (1 + r * np.cos(V)) * np.sin(U)
    z = r * np.sin(V)
    pos = np.stack([x, y, z], -1).reshape(-1, 3).astype(f32)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = i * nv + j
    b = ((i + 1) % nu) * nv + j
    c = i * nv + (j + 1) % nv
    d = ((i + 1) % nu) * nv + (j + 1) % nv
    tris = np.stack([np.stack([a, b, c], -1), np.stack([b, d, c], -1)], 2).reshape(-1, 3).astype(np.int32)
    return pos, tris
