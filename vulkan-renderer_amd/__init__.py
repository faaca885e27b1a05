"""Host-side binding of libmcull (the MI355X culling path) over its C ABI (include/mcull.h).

The package holds only what the path needs: csrc/ (HIP kernels + the C ABI) and this thin ctypes
mirror of the ABI.  There is no CPU fallback: importing works without a GPU (so the ABI can be
inspected), but every compute call goes to the gfx950 kernels and raises McullError otherwise.
A missing libmcull.so raises at import.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmcull.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libmcull.so is missing (%s): build it with `make -C %s/csrc` (hipcc, gfx950)" % (LIB_PATH, _HERE)
    )
# PyTorch ships its own copy of the HIP runtime (same soname as the system one).  When torch is
# going to share this process (device memory, streams, torch.distributed), it must be loaded first
# so that libmcull.so binds to that same runtime instead of bringing in a second one.
try:
    import torch  # noqa: F401
except ImportError:  # a pure C-ABI consumer without torch uses the system runtime
    torch = None
lib = C.CDLL(LIB_PATH)

# ---- layout contract (numpy views of the structs in include/mcull.h)
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

OK, E_INVALID, E_HIP, E_CAPACITY, E_STATE, E_NOMEM, E_IO = 0, -1, -2, -3, -4, -5, -6
MEM_HOST, MEM_DEVICE = 0, 1
TEST_FRUSTUM, TEST_CONE, TEST_HIZ, TEST_ALL = 1, 2, 4, 7
COMPACT_ORDERED, COMPACT_WAVE_ATOMIC, COMPACT_BLOCK_ATOMIC = 0, 1, 2
CMD_NV_MESHLETS, CMD_EXT_GROUPS, CMD_NV_TASK_GROUPS = 0, 1, 2
CULL_NO_INSTANCE_PRECULL, CULL_EXACT_ARITHMETIC, CULL_COUNT_SLOW_PAIRS = 1, 2, 4
CMD_EXT = np.dtype([("groupCountX", "<u4"), ("groupCountY", "<u4"), ("groupCountZ", "<u4")])
MAX_HIZ_LEVELS, MAX_VIEWS = 16, 8
STATIC_MAX_GROUPS, STATIC_MAX_INSTANCES, STATIC_SCRATCH_BYTES = 2097152, 1048576, 21070592

# every symbol include/mcull.h declares
ABI_SYMBOLS = [
    "mcull_create", "mcull_destroy", "mcull_last_error", "mcull_status_string", "mcull_synchronize", "mcull_version",
    "mcull_upload_meshlets", "mcull_upload_bounds_soa", "mcull_upload_geometry", "mcull_compute_bounds",
    "mcull_download_meshlets", "mcull_upload_transforms", "mcull_set_instance_ranges", "mcull_hiz_level_count",
    "mcull_hiz_layout", "mcull_hiz_build", "mcull_hiz_build_views", "mcull_hiz_pyramid", "mcull_hiz_download_level", "mcull_cull",
    "mcull_rebase_commands", "mcull_task_payload_count", "mcull_emit_task_payloads", "mcull_download_count",
    "mcull_download_list", "mcull_download_commands", "mcull_download_mask", "mcull_enable_timing",
    "mcull_last_kernel_ms", "mcull_scene_save", "mcull_scene_info", "mcull_scene_load", "mcull_scene_upload", "mcull_depth_rasterize", "mcull_depth_buffer", "mcull_depth_download",
    "mcull_mesh_load_obj", "mcull_mesh_from_arrays", "mcull_mesh_optimize", "mcull_mesh_build_meshlets", "mcull_mesh_counts",
    "mcull_mesh_vertices", "mcull_mesh_indices", "mcull_mesh_meshlet_data", "mcull_mesh_meshlets", "mcull_mesh_upload",
    "mcull_mesh_save", "mcull_mesh_destroy", "mcull_mesh_from_arrays_gpu", "mcull_mesh_build_meshlets_gpu", "mcull_cull_views", "mcull_select_result",
    "mcull_create_globals", "mcull_extract_frustum", "mcull_reference_transforms", "mcull_create_indirect_buffer",
    "mcull_cull_device_globals", "mcull_depth_rasterize_device_globals",
    "mcull_cull_views_device_globals", "mcull_cull_static", "mcull_precull_stats", "mcull_precull_group_stats", "mcull_slow_path_pairs", "mcull_rebase_gathered_commands",
    "mcull_group_create", "mcull_group_destroy", "mcull_group_size", "mcull_group_ctx", "mcull_group_last_error",
    "mcull_group_cull", "mcull_group_synchronize", "mcull_group_download", "mcull_group_frame",
    "mcull_allgather_commands", "mcull_rccl_unique_id", "mcull_rccl_comm_create", "mcull_rccl_comm_destroy", "mcull_collective_last_error",
]
RCCL_UNIQUE_ID_BYTES = 128
GROUP_MAX_SHARDS = 64


class CullParams(C.Structure):
    _fields_ = [("tests", C.c_uint32), ("compaction", C.c_uint32), ("hiz_view", C.c_uint32), ("cmd_form", C.c_uint32),
                ("cmd_stride", C.c_uint32), ("cmd_offset", C.c_uint32), ("write_mask", C.c_uint32),
                ("flags", C.c_uint32), ("list", C.c_void_p), ("list_capacity", C.c_uint64),
                ("commands", C.c_void_p), ("total", C.c_void_p)]


class CullResult(C.Structure):
    _fields_ = [("list", C.c_void_p), ("commands", C.c_void_p), ("total", C.c_void_p), ("mask", C.c_void_p),
                ("n_groups", C.c_uint64), ("n_pairs", C.c_uint64), ("first_task", C.c_void_p)]


class GroupResult(C.Structure):
    _fields_ = [("local", CullResult), ("commands", C.c_void_p), ("first_task", C.c_void_p), ("totals", C.c_void_p),
                ("n_instances", C.c_uint64), ("first_instance", C.c_uint64)]


def _sig(name, restype, argtypes):
    f = getattr(lib, name)
    f.restype, f.argtypes = restype, argtypes


_P, _U64, _U32, _I = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
_sig("mcull_create", _I, [_I, _P, C.POINTER(_P)])
_sig("mcull_destroy", _I, [_P])
_sig("mcull_last_error", C.c_char_p, [_P])
_sig("mcull_status_string", C.c_char_p, [_I])
_sig("mcull_synchronize", _I, [_P])
_sig("mcull_version", _I, [])
_sig("mcull_upload_meshlets", _I, [_P, _P, _U64, _I])
_sig("mcull_upload_bounds_soa", _I, [_P, _P, _P, _U64, _I, _I])
_sig("mcull_upload_geometry", _I, [_P, _P, _U64, _P, _U64, _I])
_sig("mcull_compute_bounds", _I, [_P])
_sig("mcull_download_meshlets", _I, [_P, _P, _U64])
_sig("mcull_upload_transforms", _I, [_P, _P, _U64, _I])
_sig("mcull_set_instance_ranges", _I, [_P, _P, _P, _U64])
_sig("mcull_hiz_level_count", _U32, [_U32, _U32])
_sig("mcull_hiz_layout", _U64, [_U32, _U32, C.POINTER(_U32), _P, _P, _P])
_sig("mcull_hiz_build", _I, [_P, _U32, _P, _U32, _U32, _I])
_sig("mcull_hiz_build_views", _I, [_P, _U32, _P, _P, _U32, _U32, _I])
_sig("mcull_hiz_pyramid", _I, [_P, _U32, C.POINTER(_P), C.POINTER(_U64)])
_sig("mcull_hiz_download_level", _I, [_P, _U32, _U32, _P])
_sig("mcull_cull", _I, [_P, _P, C.POINTER(CullParams), C.POINTER(CullResult)])
_sig("mcull_cull_static", _I, [_P, _P, C.POINTER(CullParams), C.POINTER(CullResult)])
_sig("mcull_slow_path_pairs", _I, [_P, C.POINTER(_U64)])
_sig("mcull_precull_stats", _I, [_P, C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_I)])
_sig("mcull_precull_group_stats", _I, [_P, C.POINTER(_U64), C.POINTER(_U64)])
_sig("mcull_rebase_commands", _I, [_P, _P, _U32])
_sig("mcull_rebase_gathered_commands", _I, [_P, _P, _U32, _U64, _P, _P])
_sig("mcull_allgather_commands", _I, [_P, _P, _P, _P, _U32, _U64, _P, _P])
_sig("mcull_rccl_unique_id", _I, [_P])
_sig("mcull_rccl_comm_create", _I, [_P, _I, _I, _I, C.POINTER(_P)])
_sig("mcull_rccl_comm_destroy", _I, [_P])
_sig("mcull_collective_last_error", C.c_char_p, [])
_sig("mcull_task_payload_count", _U64, [_P])
_sig("mcull_emit_task_payloads", _I, [_P, _P])
_sig("mcull_download_count", _I, [_P, C.POINTER(_U64)])
_sig("mcull_download_list", _I, [_P, _P, _U64, C.POINTER(_U64)])
_sig("mcull_download_commands", _I, [_P, _P, _U64])
_sig("mcull_download_mask", _I, [_P, _P, _U64])
_sig("mcull_enable_timing", _I, [_P, _I])
_sig("mcull_last_kernel_ms", _I, [_P, C.POINTER(C.c_float)])


class SceneInfo(C.Structure):
    _fields_ = [("vertex_count", C.c_uint64), ("index_count", C.c_uint64), ("meshlet_data_words", C.c_uint64),
                ("meshlet_count", C.c_uint64), ("range_count", C.c_uint64), ("checksum", C.c_uint64)]


_sig("mcull_scene_save", _I, [C.c_char_p, _P, C.c_uint64, _P, C.c_uint64, _P, C.c_uint64, _P, C.c_uint64, _P, _P, C.c_uint64])
_sig("mcull_scene_info", _I, [C.c_char_p, C.POINTER(SceneInfo)])
_sig("mcull_scene_load", _I, [C.c_char_p, C.POINTER(SceneInfo), _P, _P, _P, _P, _P, _P])
_sig("mcull_scene_upload", _I, [_P, C.c_char_p])
_sig("mcull_cull_views", _I, [_P, _U32, _P, _P, _P])
_sig("mcull_cull_views_device_globals", _I, [_P, _U32, _P, _U32, _P, _P])
_sig("mcull_select_result", _I, [_P, _U32])
_sig("mcull_cull_device_globals", _I, [_P, _P, _U32, C.POINTER(CullParams), C.POINTER(CullResult)])
_sig("mcull_depth_rasterize_device_globals", _I, [_P, _P, C.c_uint32, C.c_uint32, _I])
_sig("mcull_depth_rasterize", _I, [_P, _P, C.c_uint32, C.c_uint32, _I])
_sig("mcull_depth_buffer", _I, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)])
_sig("mcull_depth_download", _I, [_P, _P])
_sig("mcull_mesh_load_obj", _I, [C.c_char_p, C.POINTER(_P)])
_sig("mcull_mesh_from_arrays", _I, [_P, C.c_uint64, _P, C.c_uint64, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(_P)])
_sig("mcull_mesh_from_arrays_gpu", _I, [_P, _P, C.c_uint64, _P, C.c_uint64, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(_P)])
_sig("mcull_mesh_build_meshlets_gpu", _I, [_P, _P])
_sig("mcull_mesh_optimize", _I, [_P])
_sig("mcull_mesh_build_meshlets", _I, [_P])
_sig("mcull_mesh_counts", _I, [_P, C.POINTER(SceneInfo)])
for _name in ("mcull_mesh_vertices", "mcull_mesh_indices", "mcull_mesh_meshlet_data", "mcull_mesh_meshlets"):
    _sig(_name, _P, [_P])
_sig("mcull_mesh_upload", _I, [_P, _P])
_sig("mcull_mesh_save", _I, [_P, C.c_char_p])
_sig("mcull_mesh_destroy", None, [_P])
_sig("mcull_create_globals", _I, [_P, _P, _P, C.c_float, C.c_float, _U32, _U32, _I, _U32, _P])
_sig("mcull_extract_frustum", _I, [_P])
_sig("mcull_reference_transforms", _I, [_U32, _U32, _P])
_sig("mcull_create_indirect_buffer", _I, [_U32, _U32, _U32, _P])
_sig("mcull_group_create", _I, [_P, _U32, C.POINTER(_P)])
_sig("mcull_group_destroy", _I, [_P])
_sig("mcull_group_size", _U32, [_P])
_sig("mcull_group_ctx", _P, [_P, _U32])
_sig("mcull_group_last_error", C.c_char_p, [_P])
_sig("mcull_group_cull", _I, [_P, _P, C.POINTER(CullParams), _P])
_sig("mcull_group_frame", _I, [_P, _P, _U32, _U32, _U32, _U32, C.POINTER(CullParams), _P])
_sig("mcull_group_synchronize", _I, [_P])
_sig("mcull_group_download", _I, [_P, _U32, _P, _U64, _P, _P])


def _flags(precull, exact, count_slow):
    return (0 if precull else CULL_NO_INSTANCE_PRECULL) | (CULL_EXACT_ARITHMETIC if exact else 0) | \
        (CULL_COUNT_SLOW_PAIRS if count_slow else 0)


class McullError(RuntimeError):
    def __init__(self, status, text):
        RuntimeError.__init__(self, "%s: %s" % (lib.mcull_status_string(status).decode(), text))
        self.status = status


def version():
    return int(lib.mcull_version())


def _collective_chk(rc, what):
    if rc != OK:
        raise McullError(rc, "%s: %s" % (what, (lib.mcull_collective_last_error() or b"").decode()))


def rccl_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 makes them, every rank of the communicator needs them)."""
    buf = C.create_string_buffer(RCCL_UNIQUE_ID_BYTES)
    _collective_chk(lib.mcull_rccl_unique_id(buf), "mcull_rccl_unique_id")
    return buf.raw


class RcclComm:
    """An ncclComm_t made by the library (mcull_rccl_comm_create: collective over `world` ranks)."""

    def __init__(self, unique_id, world, rank, device):
        self._h = _P()
        _collective_chk(lib.mcull_rccl_comm_create(C.c_char_p(unique_id), world, rank, device, C.byref(self._h)), "mcull_rccl_comm_create")
        self.world, self.rank = world, rank

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            _collective_chk(lib.mcull_rccl_comm_destroy(self._h), "mcull_rccl_comm_destroy")
            self._h = _P()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def allgather_commands(comm, stream, send_ptr, gathered_ptr, world, rows_per_rank, totals_ptr, bases_ptr):
    """mcull_allgather_commands: ncclAllGather of this rank's command block + the rebase of the gathered firstTask
    values, both enqueued on `stream` (a hipStream_t value) by the library; `comm` an RcclComm or a raw ncclComm_t."""
    h = comm.handle if isinstance(comm, RcclComm) else comm
    _collective_chk(lib.mcull_allgather_commands(h, stream, send_ptr, gathered_ptr, world, rows_per_rank, totals_ptr, bases_ptr),
                    "mcull_allgather_commands")


def rebase_gathered_commands(stream, gathered_ptr, world, rows_per_rank, totals_ptr, bases_ptr):
    """mcull_rebase_gathered_commands on device pointers (the current device's); `stream` is a hipStream_t value."""
    rc = lib.mcull_rebase_gathered_commands(stream, gathered_ptr, world, rows_per_rank, totals_ptr, bases_ptr)
    if rc != OK:
        raise McullError(rc, "rebase_gathered_commands")


def hiz_layout(width, height):
    """(levels, widths, heights, float offsets, total floats) of the flat pyramid."""
    n = _U32()
    lw, lh, off = (_U32 * 16)(), (_U32 * 16)(), (_U64 * 16)()
    total = lib.mcull_hiz_layout(width, height, C.byref(n), lw, lh, off)
    k = n.value
    return k, list(lw)[:k], list(lh)[:k], list(off)[:k], int(total)


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def create_globals(width=800, height=600, meshlet_count=0, eye=None, centre=None, up=None, fov_deg=45.0, z_near=0.1,
                   flip_y=True):
    """Engine::createGlobals through the library's host code (Engine.cpp:2323-2352); the defaults are
    the reference's camera and window."""
    g = np.zeros(1, dtype=GLOBALS)
    vecs = [None if v is None else np.ascontiguousarray(v, dtype="<f4") for v in (eye, centre, up)]
    rc = lib.mcull_create_globals(*[None if v is None else _np_ptr(v) for v in vecs], fov_deg, z_near, width, height,
                                  1 if flip_y else 0, meshlet_count, _np_ptr(g))
    if rc != OK:
        raise McullError(rc, "create_globals")
    return g


def extract_frustum(g):
    """Recompute the five planes of a GLOBALS record in place from its view and proj."""
    rc = lib.mcull_extract_frustum(_np_ptr(g))
    if rc != OK:
        raise McullError(rc, "extract_frustum")
    return g


def reference_transforms(draw_count=1000, seed=42):
    """Engine::updateTransforms (Engine.cpp:2353-2377): srand(seed) and six rand() draws per instance."""
    t = np.zeros(draw_count, dtype=TRANSFORM)
    rc = lib.mcull_reference_transforms(draw_count, seed, _np_ptr(t))
    if rc != OK:
        raise McullError(rc, "reference_transforms")
    return t


def create_indirect_buffer(draw_count, meshlet_count, index_count):
    """Engine::createIndirectBuffer (Engine.cpp:1466-1484): the reference's static commands."""
    d = np.zeros(draw_count, dtype=DRAW_INDIRECT)
    rc = lib.mcull_create_indirect_buffer(draw_count, meshlet_count, index_count, _np_ptr(d))
    if rc != OK:
        raise McullError(rc, "create_indirect_buffer")
    return d


def _scene_chk(rc, path):
    if rc != OK:
        raise McullError(rc, "scene file %s" % path)


def save_scene(path, vertices=None, indices=None, meshlet_data=None, meshlets=None, range_offset=None, range_count=None):
    """Write the scene cache file (host code; needs no GPU).  Missing arrays are stored empty."""
    v = np.ascontiguousarray(vertices if vertices is not None else np.zeros(0, VERTEX), dtype=VERTEX)
    i = np.ascontiguousarray(indices if indices is not None else np.zeros(0, "<u4"), dtype="<u4")
    d = np.ascontiguousarray(meshlet_data if meshlet_data is not None else np.zeros(0, "<u4"), dtype="<u4")
    m = np.ascontiguousarray(meshlets if meshlets is not None else np.zeros(0, MESHLET), dtype=MESHLET)
    ro = np.ascontiguousarray(range_offset if range_offset is not None else np.zeros(0, "<u4"), dtype="<u4")
    rc = np.ascontiguousarray(range_count if range_count is not None else np.zeros(0, "<u4"), dtype="<u4")
    if len(ro) != len(rc):
        raise ValueError("range_offset and range_count differ in length")
    _scene_chk(lib.mcull_scene_save(str(path).encode(), _np_ptr(v), len(v), _np_ptr(i), len(i), _np_ptr(d), len(d),
                                    _np_ptr(m), len(m), _np_ptr(ro), _np_ptr(rc), len(ro)), path)


def scene_info(path):
    info = SceneInfo()
    _scene_chk(lib.mcull_scene_info(str(path).encode(), C.byref(info)), path)
    return info


def load_scene(path):
    """dict(vertices, indices, meshlet_data, meshlets, range_offset, range_count) read back and verified."""
    info = scene_info(path)
    out = {"vertices": np.zeros(info.vertex_count, VERTEX), "indices": np.zeros(info.index_count, "<u4"),
           "meshlet_data": np.zeros(info.meshlet_data_words, "<u4"), "meshlets": np.zeros(info.meshlet_count, MESHLET),
           "range_offset": np.zeros(info.range_count, "<u4"), "range_count": np.zeros(info.range_count, "<u4")}
    _scene_chk(lib.mcull_scene_load(str(path).encode(), C.byref(info), *[_np_ptr(out[k]) for k in
                                                          ("vertices", "indices", "meshlet_data", "meshlets", "range_offset",
                                                           "range_count")]), path)
    return out


class Mesh:
    """The reference's load-time pipeline on the host (mcull_mesh): OBJ or arrays -> deduplicated
    vertices -> reordered indices -> meshlets; bounds are filled in by Context.upload_mesh (GPU)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_obj(cls, path):
        h = _P()
        _scene_chk(lib.mcull_mesh_load_obj(str(path).encode(), C.byref(h)), path)
        return cls(h)

    @classmethod
    def from_arrays(cls, positions, corners, normals=None, texcoords=None):
        pos = np.ascontiguousarray(positions, dtype="<f4").reshape(-1, 3)
        cor = np.ascontiguousarray(corners, dtype="<i4").reshape(-1, 3)
        nrm = np.zeros((0, 3), "<f4") if normals is None else np.ascontiguousarray(normals, dtype="<f4").reshape(-1, 3)
        uv = np.zeros((0, 2), "<f4") if texcoords is None else np.ascontiguousarray(texcoords, dtype="<f4").reshape(-1, 2)
        h = _P()
        rc = lib.mcull_mesh_from_arrays(_np_ptr(pos), len(pos), _np_ptr(nrm), len(nrm), _np_ptr(uv), len(uv), _np_ptr(cor),
                                        len(cor), C.byref(h))
        if rc != OK:
            raise McullError(rc, "mesh arrays")
        return cls(h)

    def optimize(self):
        rc = lib.mcull_mesh_optimize(self._h)
        if rc != OK:
            raise McullError(rc, "optimize")
        return self

    def build_meshlets(self):
        rc = lib.mcull_mesh_build_meshlets(self._h)
        if rc != OK:
            raise McullError(rc, "build_meshlets")
        return self

    def _view(self, fn, count, dtype):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        n = count * np.dtype(dtype).itemsize
        buf = (C.c_uint8 * n).from_address(fn(self._h))
        return np.frombuffer(bytes(buf), dtype=dtype).copy()

    def counts(self):
        info = SceneInfo()
        lib.mcull_mesh_counts(self._h, C.byref(info))
        return info

    @property
    def vertices(self):
        return self._view(lib.mcull_mesh_vertices, self.counts().vertex_count, VERTEX)

    @property
    def indices(self):
        return self._view(lib.mcull_mesh_indices, self.counts().index_count, "<u4")

    @property
    def meshlet_data(self):
        return self._view(lib.mcull_mesh_meshlet_data, self.counts().meshlet_data_words, "<u4")

    @property
    def meshlets(self):
        return self._view(lib.mcull_mesh_meshlets, self.counts().meshlet_count, MESHLET)

    def save(self, path):
        _scene_chk(lib.mcull_mesh_save(self._h, str(path).encode()), path)

    def close(self):
        if self._h and lib is not None:
            lib.mcull_mesh_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass


class Context:
    """One device + one stream (mcull_ctx).  Pointer arguments are numpy arrays (host) or
    integer device addresses with mem=MEM_DEVICE."""

    def __init__(self, device=0, stream=None):
        self._h = _P()
        rc = lib.mcull_create(device, stream, C.byref(self._h))
        if rc != OK:
            self._h = None
            raise McullError(rc, "mcull_create(device=%d) failed; an MI355X (gfx950) device is required" % device)
        self.device = device

    @classmethod
    def _borrowed(cls, handle, device):
        """A Context over a handle somebody else owns (a shard of a Group): close() leaves it alone."""
        self = cls.__new__(cls)
        self._h = _P(handle)
        self._owned = False
        self.device = device
        return self

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                lib.mcull_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise McullError(rc, lib.mcull_last_error(self._h).decode())

    def synchronize(self):
        self._chk(lib.mcull_synchronize(self._h))

    # ---- uploads
    def upload_meshlets(self, meshlets, count=None, mem=MEM_HOST):
        if mem == MEM_HOST:
            meshlets = np.ascontiguousarray(meshlets, dtype=MESHLET)
            self._chk(lib.mcull_upload_meshlets(self._h, _np_ptr(meshlets), len(meshlets), MEM_HOST))
        else:
            self._chk(lib.mcull_upload_meshlets(self._h, meshlets, count, MEM_DEVICE))

    def upload_bounds_soa(self, cone_axis_w, centre_radius, count=None, mem=MEM_HOST, adopt=False):
        if mem == MEM_HOST:
            a = np.ascontiguousarray(cone_axis_w, dtype="<f4").reshape(-1, 4)
            b = np.ascontiguousarray(centre_radius, dtype="<f4").reshape(-1, 4)
            assert len(a) == len(b)
            self._chk(lib.mcull_upload_bounds_soa(self._h, _np_ptr(a), _np_ptr(b), len(a), MEM_HOST, 0))
        else:
            self._chk(lib.mcull_upload_bounds_soa(self._h, cone_axis_w, centre_radius, count, MEM_DEVICE, int(adopt)))

    def upload_geometry(self, vertices, meshlet_data):
        v = np.ascontiguousarray(vertices, dtype=VERTEX)
        d = np.ascontiguousarray(meshlet_data, dtype="<u4")
        self._chk(lib.mcull_upload_geometry(self._h, _np_ptr(v), len(v), _np_ptr(d), len(d), MEM_HOST))

    def compute_bounds(self):
        self._chk(lib.mcull_compute_bounds(self._h))

    def depth_rasterize(self, globals_, width, height, clear=True):
        """Rasterise the last cull's visible meshlets into the context's depth image."""
        g = np.ascontiguousarray(globals_, dtype=GLOBALS).reshape(1)
        self._chk(lib.mcull_depth_rasterize(self._h, _np_ptr(g), width, height, int(bool(clear))))

    def depth_rasterize_device_globals(self, device_globals_ptr, width, height, clear=True):
        """depth_rasterize() with the Globals in device memory (see cull_device_globals)."""
        self._chk(lib.mcull_depth_rasterize_device_globals(self._h, device_globals_ptr, width, height, int(bool(clear))))

    def depth_buffer(self):
        """(device pointer, width, height) of the depth image."""
        p, w, h = C.c_void_p(), _U32(), _U32()
        self._chk(lib.mcull_depth_buffer(self._h, C.byref(p), C.byref(w), C.byref(h)))
        return p.value, int(w.value), int(h.value)

    def depth_download(self):
        _, w, h = self.depth_buffer()
        out = np.zeros((h, w), dtype="<f4")
        self._chk(lib.mcull_depth_download(self._h, _np_ptr(out)))
        return out

    def mesh_from_arrays(self, positions, corners, normals=None, texcoords=None):
        """Mesh.from_arrays with the deduplication done on the GPU (same result)."""
        pos = np.ascontiguousarray(positions, dtype="<f4").reshape(-1, 3)
        cor = np.ascontiguousarray(corners, dtype="<i4").reshape(-1, 3)
        nrm = np.zeros((0, 3), "<f4") if normals is None else np.ascontiguousarray(normals, dtype="<f4").reshape(-1, 3)
        uv = np.zeros((0, 2), "<f4") if texcoords is None else np.ascontiguousarray(texcoords, dtype="<f4").reshape(-1, 2)
        h = _P()
        self._chk(lib.mcull_mesh_from_arrays_gpu(self._h, _np_ptr(pos), len(pos), _np_ptr(nrm), len(nrm), _np_ptr(uv), len(uv),
                                                 _np_ptr(cor), len(cor), C.byref(h)))
        return Mesh(h)

    def build_meshlets(self, mesh):
        """createMeshlets + bounds on the GPU; the mesh is left uploaded (same result as the host path)."""
        self._chk(lib.mcull_mesh_build_meshlets_gpu(self._h, mesh._h))

    def upload_mesh(self, mesh):
        """Upload a Mesh and complete its meshlet bounds on the GPU (written back into the mesh)."""
        self._chk(lib.mcull_mesh_upload(self._h, mesh._h))

    def upload_scene(self, path):
        """Load a scene cache file and upload its meshlets, geometry and instance ranges."""
        self._chk(lib.mcull_scene_upload(self._h, str(path).encode()))

    def download_meshlets(self, count):
        out = np.zeros(count, dtype=MESHLET)
        self._chk(lib.mcull_download_meshlets(self._h, _np_ptr(out), count))
        return out

    def upload_transforms(self, transforms, count=None, mem=MEM_HOST):
        if mem == MEM_HOST:
            t = np.ascontiguousarray(transforms, dtype=TRANSFORM)
            self._chk(lib.mcull_upload_transforms(self._h, _np_ptr(t), len(t), MEM_HOST))
        else:
            self._chk(lib.mcull_upload_transforms(self._h, transforms, count, MEM_DEVICE))

    def set_instance_ranges(self, offsets, counts):
        if offsets is None:
            self._chk(lib.mcull_set_instance_ranges(self._h, None, None, 0))
            return
        o = np.ascontiguousarray(offsets, dtype="<u4")
        n = np.ascontiguousarray(counts, dtype="<u4")
        assert len(o) == len(n)
        self._chk(lib.mcull_set_instance_ranges(self._h, _np_ptr(o), _np_ptr(n), len(o)))

    # ---- Hi-Z
    def hiz_build(self, depth, view=0, width=None, height=None, mem=MEM_HOST):
        if mem == MEM_HOST:
            d = np.ascontiguousarray(depth, dtype="<f4")
            height, width = d.shape
            self._chk(lib.mcull_hiz_build(self._h, view, _np_ptr(d), width, height, MEM_HOST))
        else:
            self._chk(lib.mcull_hiz_build(self._h, view, depth, width, height, MEM_DEVICE))
        self._hiz_dims = getattr(self, "_hiz_dims", {})
        self._hiz_dims[view] = (width, height)

    def hiz_build_views(self, depths, views=None, width=None, height=None, mem=MEM_DEVICE):
        """The pyramids of several views of one size (mcull_hiz_build_views).  depths: device pointers (mem=MEM_DEVICE, with
        width and height) or 2-D host arrays of one shape (mem=MEM_HOST); views: the slots, default 0 .. n - 1."""
        n = len(depths)
        slots = (_U32 * n)(*(range(n) if views is None else views))
        keep = []
        if mem == MEM_HOST:
            keep = [np.ascontiguousarray(d, dtype="<f4") for d in depths]
            height, width = keep[0].shape
            assert all(k.shape == keep[0].shape for k in keep)
            ptrs = (_P * n)(*[k.ctypes.data for k in keep])
        else:
            ptrs = (_P * n)(*[int(d) for d in depths])
        self._chk(lib.mcull_hiz_build_views(self._h, n, slots, ptrs, width, height, mem))
        self._hiz_dims = getattr(self, "_hiz_dims", {})
        for v in slots:
            self._hiz_dims[int(v)] = (width, height)

    def hiz_pyramid_ptr(self, view=0):
        p, n = _P(), _U64()
        self._chk(lib.mcull_hiz_pyramid(self._h, view, C.byref(p), C.byref(n)))
        return p.value, n.value

    def hiz_download(self, view=0):
        """All levels as a list of (h, w) float32 arrays."""
        w, h = self._hiz_dims[view]
        n, lw, lh, _, _ = hiz_layout(w, h)
        out = []
        for k in range(n):
            a = np.zeros((lh[k], lw[k]), dtype="<f4")
            self._chk(lib.mcull_hiz_download_level(self._h, view, k, _np_ptr(a)))
            out.append(a)
        return out

    # ---- cull + compact
    def cull(self, globals_, tests=TEST_ALL, compaction=COMPACT_ORDERED, hiz_view=0, cmd_form=CMD_NV_MESHLETS,
             cmd_stride=0, cmd_offset=0, write_mask=False, list_ptr=None, list_capacity=0, commands_ptr=None,
             total_ptr=None, precull=True, exact=False, count_slow=False):
        g = np.ascontiguousarray(globals_, dtype=GLOBALS).reshape(1)
        p = CullParams(tests, compaction, hiz_view, cmd_form, cmd_stride, cmd_offset, int(bool(write_mask)),
                       _flags(precull, exact, count_slow), list_ptr, list_capacity, commands_ptr, total_ptr)
        r = CullResult()
        self._chk(lib.mcull_cull(self._h, _np_ptr(g), C.byref(p), C.byref(r)))
        self._last = (int(r.n_groups), int(r.n_pairs), cmd_stride)
        return r

    def cull_static(self, globals_, list_ptr, list_capacity, commands_ptr, total_ptr, tests=TEST_ALL, hiz_view=0,
                    cmd_form=CMD_NV_MESHLETS, cmd_stride=0, cmd_offset=0):
        """cull() without any device allocation (mcull_cull_static): ordered compaction into caller-owned
        list / commands / total; the intermediate buffers are the GPU's static scratch array.  The context's
        stored result (download_*, rebase, payloads) is not touched."""
        g = np.ascontiguousarray(globals_, dtype=GLOBALS).reshape(1)
        p = CullParams(tests, COMPACT_ORDERED, hiz_view, cmd_form, cmd_stride, cmd_offset, 0, 0,
                       list_ptr, list_capacity, commands_ptr, total_ptr)
        r = CullResult()
        self._chk(lib.mcull_cull_static(self._h, _np_ptr(g), C.byref(p), C.byref(r)))
        return r

    def cull_device_globals(self, device_globals_ptr, meshlet_count, tests=TEST_ALL, hiz_view=0, cmd_form=CMD_NV_MESHLETS,
                            cmd_stride=0, cmd_offset=0, write_mask=False, list_ptr=None, list_capacity=0, commands_ptr=None,
                            total_ptr=None, precull=True, exact=False, count_slow=False):
        """cull() with the 212-byte Globals in device memory, read when the stream reaches the call: a captured
        HIP graph replays with whatever the buffer then holds.  Ordered compaction only."""
        p = CullParams(tests, COMPACT_ORDERED, hiz_view, cmd_form, cmd_stride, cmd_offset, int(bool(write_mask)),
                       _flags(precull, exact, count_slow), list_ptr, list_capacity, commands_ptr, total_ptr)
        r = CullResult()
        self._chk(lib.mcull_cull_device_globals(self._h, device_globals_ptr, meshlet_count, C.byref(p), C.byref(r)))
        self._last = (int(r.n_groups), int(r.n_pairs), cmd_stride)
        return r

    def cull_views(self, globals_list, params_list=None, device_globals_ptr=None, meshlet_count=0):
        """One pass for several views.  params_list: dicts with the keyword arguments of cull()
        (tests, hiz_view, cmd_*, write_mask, list_ptr, list_capacity, commands_ptr, total_ptr).
        With device_globals_ptr, globals_list is just the number of views (or a list of that length) and the
        Globals are read from that device array of 212-byte records when the stream reaches the call."""
        n = globals_list if isinstance(globals_list, int) else len(globals_list)
        if device_globals_ptr is None:
            g = np.zeros(n, dtype=GLOBALS)
            for i, x in enumerate(globals_list):
                g[i] = np.ascontiguousarray(x, dtype=GLOBALS).reshape(1)[0]
        P = (CullParams * n)()
        for i in range(n):
            d = dict(tests=TEST_ALL, compaction=COMPACT_ORDERED, hiz_view=0, cmd_form=CMD_NV_MESHLETS, cmd_stride=0, cmd_offset=0,
                     write_mask=False, list_ptr=None, list_capacity=0, commands_ptr=None, total_ptr=None, precull=True,
                     exact=False, count_slow=False)
            if params_list is not None:
                d.update(params_list[i])
            P[i] = CullParams(d["tests"], d["compaction"], d["hiz_view"], d["cmd_form"], d["cmd_stride"], d["cmd_offset"],
                              int(bool(d["write_mask"])), _flags(d["precull"], d["exact"], d["count_slow"]), d["list_ptr"],
                              d["list_capacity"], d["commands_ptr"], d["total_ptr"])
        R = (CullResult * n)()
        if device_globals_ptr is None:
            self._chk(lib.mcull_cull_views(self._h, n, _np_ptr(g), C.cast(P, _P), C.cast(R, _P)))
        else:
            self._chk(lib.mcull_cull_views_device_globals(self._h, n, device_globals_ptr, meshlet_count, C.cast(P, _P), C.cast(R, _P)))
        self._views = [(int(r.n_groups), int(r.n_pairs)) for r in R]
        self._last = (self._views[0][0], self._views[0][1], 0)
        return list(R)

    def select_result(self, view):
        """Point the download / rebase / payload / rasterise calls at the result of view slot `view`."""
        self._chk(lib.mcull_select_result(self._h, view))
        if getattr(self, "_views", None) and view < len(self._views):
            self._last = (self._views[view][0], self._views[view][1], 0)

    def rebase_commands(self, device_totals_ptr, rank):
        self._chk(lib.mcull_rebase_commands(self._h, device_totals_ptr, rank))

    def slow_path_pairs(self):
        """Pairs of the last cull (run with count_slow=True) that left the fast predicate forms."""
        n = _U64()
        self._chk(lib.mcull_slow_path_pairs(self._h, C.byref(n)))
        return int(n.value)

    def precull_stats(self):
        """(marked, tested, per_group) of the last cull's instance pre-cull: instances (or, for scenes with
        instance ranges, 64-meshlet groups) that were dropped before any meshlet was loaded."""
        m, t, pg = _U64(), _U64(), _I()
        self._chk(lib.mcull_precull_stats(self._h, C.byref(m), C.byref(t), C.byref(pg)))
        return int(m.value), int(t.value), bool(pg.value)

    def precull_group_stats(self):
        """(dropped, tested) of the last cull's group level: 64-meshlet groups of the instances the instance level
        kept, and how many of them went before any of their meshlets was loaded; (0, 0) when it did not run."""
        d, t = _U64(), _U64()
        self._chk(lib.mcull_precull_group_stats(self._h, C.byref(d), C.byref(t)))
        return int(d.value), int(t.value)

    def download_count(self):
        n = _U64()
        rc = lib.mcull_download_count(self._h, C.byref(n))
        if rc == E_CAPACITY:  # 2^32 visible pairs or more: the exact count travels with the error
            err = McullError(rc, lib.mcull_last_error(self._h).decode())
            err.count = int(n.value)
            raise err
        self._chk(rc)
        return int(n.value)

    def download_list(self, capacity=None):
        n = self.download_count() if capacity is None else capacity
        out = np.zeros(max(n, 1), dtype=VISIBLE)
        tot = _U64()
        self._chk(lib.mcull_download_list(self._h, _np_ptr(out), n, C.byref(tot)))
        return out[: int(tot.value)]

    def download_commands(self, n_instances, stride=8):
        raw = np.zeros(n_instances * stride, dtype="u1")
        self._chk(lib.mcull_download_commands(self._h, _np_ptr(raw), raw.nbytes))
        return raw

    def download_mask(self):
        words = self._last[0]
        out = np.zeros(max(words, 1), dtype="<u8")
        self._chk(lib.mcull_download_mask(self._h, _np_ptr(out), words))
        return out[:words]

    def task_payloads(self):
        """Literal per-32 task payloads of the last ordered cull (device buffer from torch)."""
        import torch

        n = int(lib.mcull_task_payload_count(self._h))
        buf = torch.empty(max(n, 1) * TASK_PAYLOAD.itemsize, dtype=torch.uint8, device="cuda:%d" % self.device)
        self._chk(lib.mcull_emit_task_payloads(self._h, buf.data_ptr()))
        self.synchronize()
        return np.frombuffer(buf.cpu().numpy().tobytes(), dtype=TASK_PAYLOAD)[:n]

    def enable_timing(self, on=True):
        self._chk(lib.mcull_enable_timing(self._h, int(bool(on))))

    def last_kernel_ms(self):
        ms = C.c_float()
        self._chk(lib.mcull_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)


class Group:
    """Several shards of one scene driven by one host thread (mcull_group): shard r holds the r-th contiguous
    range of the instances; `shards[r]` is its Context for the uploads."""

    def __init__(self, device_ordinals):
        ords = (C.c_int * len(device_ordinals))(*device_ordinals)
        self._h = _P()
        rc = lib.mcull_group_create(C.cast(ords, _P), len(device_ordinals), C.byref(self._h))
        if rc != OK:
            self._h = None
            raise McullError(rc, "mcull_group_create(%r) failed" % (list(device_ordinals),))
        self.shards = [Context._borrowed(lib.mcull_group_ctx(self._h, r), d) for r, d in enumerate(device_ordinals)]

    def close(self):
        if getattr(self, "_h", None):
            for s in self.shards:
                s.close()
            lib.mcull_group_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise McullError(rc, lib.mcull_group_last_error(self._h).decode())

    def cull(self, globals_, tests=TEST_ALL, hiz_view=0, cmd_form=CMD_NV_MESHLETS, cmd_stride=0, cmd_offset=0,
             compaction=COMPACT_ORDERED, list_ptr=None):
        g = np.ascontiguousarray(globals_, dtype=GLOBALS).reshape(1)
        p = CullParams(tests, compaction, hiz_view, cmd_form, cmd_stride, cmd_offset, 0, 0, list_ptr, 0, None, None)
        n = len(self.shards)
        R = (GroupResult * n)()
        self._chk(lib.mcull_group_cull(self._h, _np_ptr(g), C.byref(p), C.cast(R, _P)))
        self._stride = cmd_stride or (12 if cmd_form == CMD_EXT_GROUPS else 8)
        self._ext = cmd_form == CMD_EXT_GROUPS
        self._n_instances = int(R[0].n_instances)
        for s, r in zip(self.shards, R):
            s._last = (int(r.local.n_groups), int(r.local.n_pairs), cmd_stride)
        return list(R)

    def frame(self, globals_, width, height, prepass_tests=TEST_FRUSTUM | TEST_CONE, view=0, tests=TEST_ALL,
              cmd_form=CMD_NV_MESHLETS, cmd_stride=0, cmd_offset=0):
        """The two-phase frame over the shards: cull, rasterise, merge the depth images, pyramid, cull + exchange."""
        g = np.ascontiguousarray(globals_, dtype=GLOBALS).reshape(1)
        p = CullParams(tests, COMPACT_ORDERED, view, cmd_form, cmd_stride, cmd_offset, 0, 0, None, 0, None, None)
        n = len(self.shards)
        R = (GroupResult * n)()
        self._chk(lib.mcull_group_frame(self._h, _np_ptr(g), width, height, prepass_tests, view, C.byref(p), C.cast(R, _P)))
        for s in self.shards:  # every shard now holds the scene's pyramid in that slot
            s._hiz_dims = getattr(s, "_hiz_dims", {})
            s._hiz_dims[view] = (width, height)
        self._stride = cmd_stride or (12 if cmd_form == CMD_EXT_GROUPS else 8)
        self._ext = cmd_form == CMD_EXT_GROUPS
        self._n_instances = int(R[0].n_instances)
        for s, r in zip(self.shards, R):
            s._last = (int(r.local.n_groups), int(r.local.n_pairs), cmd_stride)
        return list(R)

    def synchronize(self):
        self._chk(lib.mcull_group_synchronize(self._h))

    def download(self, shard):
        """(commands as raw bytes, first_task or None, totals[n_shards + 1]) as device `shard` holds them."""
        raw = np.zeros(max(self._n_instances * self._stride, 1), dtype="u1")
        ft = np.zeros(max(self._n_instances, 1), dtype="<u4") if self._ext else None
        totals = np.zeros(len(self.shards) + 1, dtype="<u8")
        self._chk(lib.mcull_group_download(self._h, shard, _np_ptr(raw), self._n_instances * self._stride,
                                           _np_ptr(ft) if ft is not None else None, _np_ptr(totals)))
        return raw[: self._n_instances * self._stride], (ft[: self._n_instances] if ft is not None else None), totals
