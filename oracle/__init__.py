"""ctypes binding of the CPU oracle (oracle/libmcull_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, the smoke check and the bench's cpu_baseline
leg; the product package never imports this module.  PARITY PINNED ONLY IN PART: the reference
has no tests or golden vectors for the culling path; its two screenshots pin the scene inputs and
the load-time pipeline, the rest is PARITY UNPINNED (see mcull_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmcull_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (make -C oracle)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("mcull_oracle.c", "mcull_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


# ---------------------------------------------------------------- numpy views of the ABI structs
VERTEX = np.dtype([("pos", "<f4", 4), ("n", "u1", 4), ("uv", "<u2", 2)])
MESHLET = np.dtype(
    [("cone", "<f4", 4), ("coneApex", "<f4", 4), ("radius", "<f4"), ("dataOffset", "<u4"),
     ("triangleCount", "u1"), ("vertexCount", "u1"), ("padding", "u1", 6)]
)
GLOBALS = np.dtype([("view", "<f4", 16), ("proj", "<f4", 16), ("frustum", "<f4", (5, 4)), ("meshletCount", "<u4")])
TRANSFORM = np.dtype([("model", "<f4", 16)])
CMD = np.dtype([("taskCount", "<u4"), ("firstTask", "<u4")])
DRAW_INDIRECT = np.dtype(
    [("indexCount", "<u4"), ("instanceCount", "<u4"), ("firstIndex", "<u4"), ("vertexOffset", "<i4"),
     ("firstInstance", "<u4"), ("commandMeshIndirect", CMD)]
)
VISIBLE = np.dtype([("drawID", "<u4"), ("meshletID", "<u4")])
TASK_PAYLOAD = np.dtype([("meshletIndices", "<u4", 32), ("drawID", "<u4"), ("taskCount", "<u4")])
assert VERTEX.itemsize == 24 and MESHLET.itemsize == 48 and GLOBALS.itemsize == 212
assert TRANSFORM.itemsize == 64 and CMD.itemsize == 8 and DRAW_INDIRECT.itemsize == 28
assert VISIBLE.itemsize == 8 and TASK_PAYLOAD.itemsize == 136

TEST_FRUSTUM, TEST_CONE, TEST_HIZ, TEST_ALL = 1, 2, 4, 7


class _Mesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertex_count", C.c_uint64), ("indices", C.c_void_p),
                ("index_count", C.c_uint64), ("meshlets", C.c_void_p), ("meshlet_count", C.c_uint64),
                ("meshlet_data", C.c_void_p), ("meshlet_data_words", C.c_uint64)]


class _View(C.Structure):
    _fields_ = [("g", C.c_uint8 * 212), ("cam_pos", C.c_float * 3), ("vp", C.c_float * 16),
                ("pyramid", C.c_void_p), ("w0", C.c_uint32), ("h0", C.c_uint32), ("levels", C.c_uint32),
                ("lw", C.c_uint32 * 16), ("lh", C.c_uint32 * 16), ("off", C.c_uint64 * 16)]


class _PairResult(C.Structure):
    _fields_ = [("visible", C.c_int), ("frustum_culled", C.c_int), ("occlusion_culled", C.c_int),
                ("cone_culled", C.c_int), ("frustum_margin", C.c_float), ("occlusion_margin", C.c_float),
                ("cone_margin", C.c_float), ("mip", C.c_uint32), ("tx", C.c_uint32), ("ty", C.c_uint32)]


class _Scene(C.Structure):
    _fields_ = [("meshlets", C.c_void_p), ("meshlet_count", C.c_uint64), ("cone_axis_w", C.c_void_p),
                ("centre_radius", C.c_void_p), ("transforms", C.c_void_p), ("instance_count", C.c_uint64),
                ("range_offset", C.c_void_p), ("range_count", C.c_void_p)]


class _CullOutput(C.Structure):
    _fields_ = [("n_groups", C.c_uint64), ("mask", C.c_void_p), ("list", C.c_void_p),
                ("list_capacity", C.c_uint64), ("total", C.c_uint64), ("commands", C.c_void_p),
                ("n_pairs", C.c_uint64), ("culled_frustum", C.c_uint64), ("culled_occlusion", C.c_uint64),
                ("culled_cone", C.c_uint64), ("margins", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_fnv1a64.restype = C.c_uint64
        L.orc_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_vertex_score.restype = C.c_float
        L.orc_vertex_score.argtypes = [C.c_int, C.c_int]
        L.orc_float_to_half.restype = C.c_uint16
        L.orc_float_to_half.argtypes = [C.c_float]
        L.orc_hiz_level_count.restype = C.c_uint32
        L.orc_hiz_level_count.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_hiz_layout.restype = C.c_uint64
        L.orc_mip_level.restype = C.c_uint32
        L.orc_mip_level.argtypes = [C.c_float, C.c_uint32]
        L.orc_task_payloads.restype = C.c_uint64
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def rasterize_depth(vertices, meshlet_data, meshlets, transforms, visible, globals_, w, h, depth=None):
    """Depth pre-pass stand-in over a visible list; returns the (h, w) reverse-Z image (accumulates
    into `depth` when given)."""
    v = np.ascontiguousarray(vertices, dtype=VERTEX)
    d = np.ascontiguousarray(meshlet_data, dtype="<u4")
    m = np.ascontiguousarray(meshlets, dtype=MESHLET)
    t = np.ascontiguousarray(transforms, dtype=TRANSFORM)
    lst = np.ascontiguousarray(visible, dtype=VISIBLE)
    g = np.array(globals_, dtype=GLOBALS).reshape(1).copy()
    out = np.zeros((h, w), dtype="<f4") if depth is None else np.ascontiguousarray(depth, dtype="<f4").copy()
    lib().orc_rasterize_depth(_ptr(v), _ptr(d), _ptr(m), _ptr(t), _ptr(lst), C.c_uint64(len(lst)), _ptr(g),
                              C.c_uint32(w), C.c_uint32(h), _ptr(out))
    return out


def fnv1a64(a):
    a = np.ascontiguousarray(a)
    return int(lib().orc_fnv1a64(_ptr(a), a.nbytes))


def max_threads():
    return int(lib().orc_max_threads())


class Mesh:
    """Result of the reference's CPU geometry pipeline (Engine.cpp:243-628)."""

    def __init__(self, vertices, indices, meshlets=None, meshlet_data=None):
        self.vertices, self.indices, self.meshlets, self.meshlet_data = vertices, indices, meshlets, meshlet_data


def _take(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype=dtype)
    n = count * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * n).from_address(ptr)
    return np.frombuffer(bytes(buf), dtype=dtype).copy()


def _mesh_out(m):
    out = Mesh(_take(m.vertices, m.vertex_count, VERTEX), _take(m.indices, m.index_count, "<u4"),
               _take(m.meshlets, m.meshlet_count, MESHLET), _take(m.meshlet_data, m.meshlet_data_words, "<u4"))
    lib().orc_mesh_free(C.byref(m))
    return out


def process_mesh(m, optimize=True, meshletize=True):
    L = lib()
    if optimize:
        L.orc_optimize_geometry(C.byref(m))
    if meshletize:
        L.orc_create_meshlets(C.byref(m))
    return _mesh_out(m)


def load_obj(path, optimize=True, meshletize=True):
    """loadModel (+ optimizeGeometry) + createMeshlets."""
    m = _Mesh()
    if lib().orc_load_obj(str(path).encode(), C.byref(m)) != 0:
        raise RuntimeError("orc_load_obj failed for %s" % path)
    return process_mesh(m, optimize, meshletize)


def build_mesh(pos, corners, nrm=None, uv=None, optimize=True, meshletize=True):
    """Same pipeline from arrays: pos (n,3) f32, corners (k,3) int32 of (v, t, n) indices, -1 = absent."""
    pos = np.ascontiguousarray(pos, dtype="<f4")
    corners = np.ascontiguousarray(corners, dtype="<i4")
    nrm = None if nrm is None else np.ascontiguousarray(nrm, dtype="<f4")
    uv = None if uv is None else np.ascontiguousarray(uv, dtype="<f4")
    m = _Mesh()
    rc = lib().orc_build_vertices(_ptr(pos), C.c_uint64(len(pos)), _ptr(nrm), C.c_uint64(0 if nrm is None else len(nrm)),
                                  _ptr(uv), C.c_uint64(0 if uv is None else len(uv)), _ptr(corners),
                                  C.c_uint64(len(corners)), C.byref(m))
    if rc != 0:
        raise RuntimeError("orc_build_vertices failed")
    return process_mesh(m, optimize, meshletize)


def build_meshlet_cone(vertices, meshlet_data, meshlet):
    """buildMeshletCon for one MESHLET record (returns a new record)."""
    v = np.ascontiguousarray(vertices, dtype=VERTEX)
    d = np.ascontiguousarray(meshlet_data, dtype="<u4")
    ml = np.array(meshlet, dtype=MESHLET).reshape(1).copy()
    lib().orc_build_meshlet_cone(_ptr(v), _ptr(d), _ptr(ml))
    return ml[0]


def build_meshlet_cones(vertices, meshlet_data, meshlets):
    """buildMeshletCon over an array of MESHLET records (returns a completed copy)."""
    v = np.ascontiguousarray(vertices, dtype=VERTEX)
    d = np.ascontiguousarray(meshlet_data, dtype="<u4")
    ml = np.array(meshlets, dtype=MESHLET).copy()
    lib().orc_build_meshlet_cones(_ptr(v), _ptr(d), _ptr(ml), C.c_uint64(len(ml)))
    return ml


def create_globals(eye=(2.0, 2.0, 2.0), centre=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0), fov_deg=45.0, z_near=0.1,
                   width=800, height=600, flip_y=True, meshlet_count=0):
    g = np.zeros(1, dtype=GLOBALS)
    f3 = C.c_float * 3
    lib().orc_create_globals(f3(*eye), f3(*centre), f3(*up), C.c_float(fov_deg), C.c_float(z_near),
                             C.c_uint32(width), C.c_uint32(height), C.c_int(1 if flip_y else 0),
                             C.c_uint32(meshlet_count), _ptr(g))
    return g


def extract_frustum(g):
    g = np.array(g, dtype=GLOBALS).reshape(1).copy()
    lib().orc_extract_frustum(_ptr(g))
    return g


def update_transforms(draw_count=1000):
    t = np.zeros(draw_count, dtype=TRANSFORM)
    lib().orc_update_transforms(C.c_uint32(draw_count), _ptr(t))
    return t


def create_indirect_buffer(draw_count, meshlet_count, index_count):
    d = np.zeros(draw_count, dtype=DRAW_INDIRECT)
    lib().orc_create_indirect_buffer(C.c_uint32(draw_count), C.c_uint32(meshlet_count), C.c_uint32(index_count), _ptr(d))
    return d


def hiz_layout(w, h):
    levels = C.c_uint32()
    lw, lh, off = (C.c_uint32 * 16)(), (C.c_uint32 * 16)(), (C.c_uint64 * 16)()
    total = lib().orc_hiz_layout(C.c_uint32(w), C.c_uint32(h), C.byref(levels), lw, lh, off)
    n = levels.value
    return n, list(lw)[:n], list(lh)[:n], list(off)[:n], int(total)


def hiz_build(depth, threads=1):
    """Flat pyramid (all levels, layout as hiz_layout) from an (h, w) float32 depth image."""
    depth = np.ascontiguousarray(depth, dtype="<f4")
    h, w = depth.shape
    total = hiz_layout(w, h)[4]
    pyr = np.empty(total, dtype="<f4")
    lib().orc_hiz_build(_ptr(depth), C.c_uint32(w), C.c_uint32(h), _ptr(pyr), C.c_int(threads))
    return pyr


def hiz_levels(pyr, w, h):
    n, lw, lh, off, _ = hiz_layout(w, h)
    return [pyr[off[i]: off[i] + lw[i] * lh[i]].reshape(lh[i], lw[i]) for i in range(n)]


class View:
    def __init__(self, globals_, pyramid=None, w0=0, h0=0):
        self.g = np.array(globals_, dtype=GLOBALS).reshape(1).copy()
        self.pyramid = None if pyramid is None else np.ascontiguousarray(pyramid, dtype="<f4")
        self.c = _View()
        lib().orc_view_init(C.byref(self.c), _ptr(self.g), _ptr(self.pyramid), C.c_uint32(w0), C.c_uint32(h0))

    @property
    def cam_pos(self):
        return np.array(list(self.c.cam_pos), dtype="<f4")

    @property
    def vp(self):
        return np.array(list(self.c.vp), dtype="<f4")


def cull_pair(view, model, cone, centre, radius, tests=TEST_ALL):
    r = _PairResult()
    f = lambda a, n: (C.c_float * n)(*[float(x) for x in np.asarray(a, dtype="<f4").ravel()[:n]])
    lib().orc_cull_pair(C.byref(view.c), f(model, 16), f(cone, 4), f(centre, 3), C.c_float(radius), C.c_uint32(tests), C.byref(r))
    return r


def mip_level(px, levels):
    return int(lib().orc_mip_level(C.c_float(px), C.c_uint32(levels)))


class CullResult:
    pass


def cull_scene(view, transforms, meshlets=None, cone_axis_w=None, centre_radius=None, ranges=None,
               tests=TEST_ALL, threads=1, want_list=True, want_margins=False, list_capacity=None):
    """shader.task over every (instance, meshlet) pair + global compaction."""
    s = _Scene()
    keep = []
    if meshlets is not None:
        meshlets = np.ascontiguousarray(meshlets, dtype=MESHLET)
        s.meshlets, s.meshlet_count = _ptr(meshlets), len(meshlets)
        keep.append(meshlets)
    else:
        a = np.ascontiguousarray(cone_axis_w, dtype="<f4")
        b = np.ascontiguousarray(centre_radius, dtype="<f4")
        s.cone_axis_w, s.centre_radius, s.meshlet_count = _ptr(a), _ptr(b), a.size // 4
        keep += [a, b]
    transforms = np.ascontiguousarray(transforms, dtype=TRANSFORM)
    s.transforms, s.instance_count = _ptr(transforms), len(transforms)
    n_inst = len(transforms)
    if ranges is not None:
        ro = np.ascontiguousarray(ranges[0], dtype="<u4")
        rc = np.ascontiguousarray(ranges[1], dtype="<u4")
        s.range_offset, s.range_count = _ptr(ro), _ptr(rc)
        n_pairs = int(rc.astype(np.uint64).sum())
        n_groups = int(((rc.astype(np.uint64) + 63) // 64).sum())
        keep += [ro, rc]
    else:
        mc = int(view.g["meshletCount"][0])
        n_pairs = mc * n_inst
        n_groups = ((mc + 63) // 64) * n_inst
    out = _CullOutput()
    mask = np.zeros(max(n_groups, 1), dtype="<u8")
    cmds = np.zeros(n_inst, dtype=CMD)
    out.mask, out.commands = _ptr(mask), _ptr(cmds)
    lst = None
    if want_list:
        cap = n_pairs if list_capacity is None else list_capacity
        lst = np.zeros(max(cap, 1), dtype=VISIBLE)
        out.list, out.list_capacity = _ptr(lst), cap
    margins = None
    if want_margins:
        margins = np.zeros((max(n_pairs, 1), 3), dtype="<f4")
        out.margins = _ptr(margins)
    rc_ = lib().orc_cull_scene(C.byref(s), C.byref(view.c), C.c_uint32(tests), C.c_int(threads), C.byref(out))
    r = CullResult()
    r.overflow = rc_ != 0
    r.total, r.n_pairs, r.n_groups = int(out.total), int(out.n_pairs), int(out.n_groups)
    r.mask = mask[:n_groups]
    r.commands = cmds
    r.list = None if lst is None or r.overflow else lst[: r.total]
    r.margins = None if margins is None else margins[:n_pairs]
    r.culled = dict(frustum=int(out.culled_frustum), occlusion=int(out.culled_occlusion), cone=int(out.culled_cone))
    return r


def task_payloads(view, transforms, meshlets, tests=TEST_ALL):
    meshlets = np.ascontiguousarray(meshlets, dtype=MESHLET)
    transforms = np.ascontiguousarray(transforms, dtype=TRANSFORM)
    s = _Scene()
    s.meshlets, s.meshlet_count = _ptr(meshlets), len(meshlets)
    s.transforms, s.instance_count = _ptr(transforms), len(transforms)
    mc = int(view.g["meshletCount"][0])
    out = np.zeros(len(transforms) * ((mc + 31) // 32), dtype=TASK_PAYLOAD)
    n = lib().orc_task_payloads(C.byref(s), C.byref(view.c), C.c_uint32(tests), _ptr(out))
    return out[:n]
