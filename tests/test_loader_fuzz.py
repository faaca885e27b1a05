"""Corrupted scene cache files and malformed OBJ text through the library's host loaders (no GPU): every input
ends in a clean result or an error code, never in a crash, and no flipped payload bit goes unnoticed."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("seed", [1, 2])
def test_corrupted_scene_files(tmp_path, mc, seed):
    d = np.load(os.path.join(GOLDEN, "kitten_scene.npz"))
    path = str(tmp_path / "k.mcs")
    mc.save_scene(path, d["vertices"].view(mc.VERTEX).reshape(-1), None, d["meshlet_data"], d["meshlets"].view(mc.MESHLET).reshape(-1))
    raw = bytearray(open(path, "rb").read())
    header = mc.scene_info(path)
    rs = np.random.RandomState(seed)
    bad = str(tmp_path / "bad.mcs")
    extremes = [0, 1, 2 ** 31, 2 ** 32 - 1, 2 ** 40, 2 ** 63, 2 ** 64 - 1]
    for it in range(160):
        b = bytearray(raw)
        mode = it % 4
        if mode == 0:      # header bytes
            for _ in range(rs.randint(1, 4)):
                b[rs.randint(0, 96)] = rs.randint(0, 256)
        elif mode == 1:    # header words set to extreme values
            off = 8 * rs.randint(0, 12)
            b[off:off + 8] = extremes[rs.randint(0, len(extremes))].to_bytes(8, "little")
        elif mode == 2:    # truncation
            b = b[: rs.randint(0, len(b))]
        else:              # payload bit flips
            for _ in range(3):
                b[rs.randint(96, len(b))] ^= 1 << rs.randint(0, 8)
        with open(bad, "wb") as f:
            f.write(bytes(b))
        try:
            mc.scene_info(bad)
            got = mc.load_scene(bad)
            ok = True
        except mc.McullError as e:
            assert e.status in (mc.E_INVALID, mc.E_IO)
            ok = False
        if mode in (2, 3):
            assert not ok, "a truncated or flipped file was accepted"
        if ok:  # only a mutation of bytes that carry nothing (or rewriting a value with itself) loads: same content
            assert got["meshlets"].tobytes() == d["meshlets"].view(mc.MESHLET).reshape(-1).tobytes()
    assert header is not None


OBJ_LINES = ["v 0 0 0", "v 1 0 0", "v 0 1 0", "vn 0 0 1", "vt 0.5 0.5", "f 1 2 3", "f 1/1/1 2/1/1 3/1/1", "f 1//1 2//1 3//1",
             "f -1 -2 -3", "f 1 2 3 4 5", "f 1 2", "f", "f 999999999999 2 3", "f 0 0 0", "f 1/ 2/ 3/", "f a b c", "v nan inf -inf",
             "v 1e999 0 0", "v", "vn 0 0", "vt", "#c", "", "o x", "g y", "usemtl z", "s off", "f 4 5 6", "f -99 1 2", "v 1 2 3 4 5 6",
             "f 1/2/3/4 2 3", "f 2147483648 1 2", "\x00\x01", "f 1 2 3 " + "9 " * 200]


@pytest.mark.parametrize("seed", [1, 2])
def test_malformed_obj_text(tmp_path, mc, seed):
    rs = np.random.RandomState(seed)
    path = str(tmp_path / "x.obj")
    loaded = 0
    for it in range(160):
        txt = "\n".join(OBJ_LINES[i] for i in rs.randint(0, len(OBJ_LINES), rs.randint(0, 40)))
        if it % 5 == 0:
            txt = txt[: rs.randint(0, len(txt) + 1)]
        with open(path, "w") as f:
            f.write(txt)
        try:
            m = mc.Mesh.from_obj(path)
        except mc.McullError as e:
            assert e.status == mc.E_INVALID
            continue
        loaded += 1
        n_idx = len(m.indices)
        assert n_idx % 3 == 0 and (n_idx == 0 or int(m.indices.max()) < len(m.vertices))
        if n_idx:
            m.optimize().build_meshlets()
            assert int(m.meshlets["triangleCount"].sum()) == n_idx // 3
        m.close()
    assert loaded > 5


def test_a_header_that_promises_more_than_the_file_holds(tmp_path, mc):
    """A crafted header whose counts and payload size agree with each other but not with the file: it must be
    refused from the file's real length, before anybody sizes an allocation from it (2^39 vertices = 12 TiB)."""
    import ctypes as C
    import struct

    d = np.load(os.path.join(GOLDEN, "kitten_scene.npz"))
    path = str(tmp_path / "k.mcs")
    mc.save_scene(path, d["vertices"].view(mc.VERTEX).reshape(-1), None, d["meshlet_data"], d["meshlets"].view(mc.MESHLET).reshape(-1))
    raw = bytearray(open(path, "rb").read())
    pad16 = lambda n: (n + 15) & ~15
    for nv in (2 ** 39, 2 ** 33 + 5, len(d["vertices"]) + 1):
        b = bytearray(raw)
        v, i, w, m, r = struct.unpack_from("<5Q", b, 16)
        struct.pack_into("<Q", b, 16, nv)
        struct.pack_into("<Q", b, 56, pad16(nv * 24) + pad16(i * 4) + pad16(w * 4) + pad16(m * 48) + 2 * pad16(r * 4))
        bad = str(tmp_path / "bad.mcs")
        open(bad, "wb").write(bytes(b))
        with pytest.raises(mc.McullError) as e:
            mc.scene_info(bad)
        assert e.value.status == mc.E_INVALID
        with pytest.raises(mc.McullError) as e:
            mc.load_scene(bad)
        assert e.value.status == mc.E_INVALID
        assert mc.lib.mcull_scene_upload(None, bad.encode()) == mc.E_INVALID
    # a file replaced between sizing and loading: the counts handed back to the loader no longer match
    info = mc.scene_info(path)
    small = str(tmp_path / "small.mcs")
    mc.save_scene(small, d["vertices"].view(mc.VERTEX).reshape(-1)[:100], None, d["meshlet_data"][:64], d["meshlets"].view(mc.MESHLET).reshape(-1)[:2])
    out = np.zeros(info.vertex_count, mc.VERTEX)
    rc = mc.lib.mcull_scene_load(small.encode(), C.byref(info), out.ctypes.data_as(C.c_void_p), None, None, None, None, None)
    assert rc == mc.E_INVALID and not out.view("u1").any(), "nothing was written"
    # trailing garbage makes the length disagree too
    open(small, "ab").write(b"\0" * 16)
    with pytest.raises(mc.McullError):
        mc.scene_info(small)
