"""The product's host-side mesh pipeline (mcull_mesh_*: load/dedup, reorder, meshlets) against the
oracle's statement-for-statement restatement of the same reference code.  Index buffers, vertex
arrays and packed meshlet data must be identical; the product leaves cone/apex/radius zero (they are
computed on the GPU, see tests/test_gpu_mesh.py).  CPU only."""
import os

import numpy as np
import pytest

REF_ASSETS = "/root/reference/assets"


def _headers(ml):
    h = ml.copy()
    h["cone"] = 0; h["coneApex"] = 0; h["radius"] = 0
    return h.tobytes()


def _compare(mesh, ref):
    assert mesh.vertices.tobytes() == ref.vertices.tobytes()
    assert np.array_equal(mesh.indices, ref.indices)
    assert np.array_equal(mesh.meshlet_data, ref.meshlet_data)
    assert _headers(mesh.meshlets) == _headers(ref.meshlets)
    assert not mesh.meshlets["cone"].any() and not mesh.meshlets["radius"].any()


def _corners(tris):
    c = np.full((tris.size, 3), -1, dtype="<i4")
    c[:, 0] = tris.reshape(-1)
    return c


def test_proxy_mesh_matches_oracle(mc, O, H):
    pos, tris = H.proxy_mesh(90, 70, seed=3)
    mesh = mc.Mesh.from_arrays(pos, _corners(tris)).optimize().build_meshlets()
    _compare(mesh, O.build_mesh(pos, _corners(tris)))
    # each stage on its own
    m2 = mc.Mesh.from_arrays(pos, _corners(tris))
    r2 = O.build_mesh(pos, _corners(tris), optimize=False, meshletize=False)
    assert np.array_equal(m2.indices, r2.indices) and m2.vertices.tobytes() == r2.vertices.tobytes()
    m2.build_meshlets()  # meshlets straight from the unoptimised order
    r3 = O.build_mesh(pos, _corners(tris), optimize=False, meshletize=True)
    assert np.array_equal(m2.meshlet_data, r3.meshlet_data) and _headers(m2.meshlets) == _headers(r3.meshlets)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_triangle_soup_with_duplicates_and_degenerates(mc, O, seed):
    """Random soup: duplicate positions with different normals / uvs (seams), -0 vs +0, repeated
    vertices inside a triangle, isolated triangles (the reorderer's full-scan path), NaN positions."""
    rs = np.random.RandomState(seed)
    n_pos, n_tri = 300, 1500
    pos = rs.randint(-4, 5, (n_pos, 3)).astype("<f4") * 0.25     # many coincident positions
    pos[pos == 0] *= rs.choice([1.0, -1.0], size=(pos == 0).sum())  # mix of +0 and -0
    if seed == 3:
        pos[7, 1] = np.nan
    nrm = rs.normal(size=(40, 3)).astype("<f4")
    nrm[5] = 0.0                                                  # normalises to NaN
    uv = rs.uniform(-2, 2, (50, 2)).astype("<f4")
    uv[3] = [70000.0, 1e-9]                                       # half overflow / underflow
    cor = np.empty((n_tri * 3, 3), dtype="<i4")
    cor[:, 0] = rs.randint(0, n_pos, n_tri * 3)
    cor[:, 1] = rs.randint(0, len(uv), n_tri * 3)
    cor[:, 2] = rs.randint(0, len(nrm), n_tri * 3)
    cor[30:33, 0] = cor[30, 0]                                    # a triangle made of one vertex
    cor[60, 0] = cor[61, 0]
    mesh = mc.Mesh.from_arrays(pos, cor, normals=nrm, texcoords=uv).optimize().build_meshlets()
    ref = O.build_mesh(pos, cor, nrm=nrm, uv=uv)
    _compare(mesh, ref)
    assert len(mesh.meshlets) > 10
    # without normals / texcoords (the bunny case: NaN normal -> 0)
    mesh = mc.Mesh.from_arrays(pos, cor).optimize().build_meshlets()
    _compare(mesh, O.build_mesh(pos, cor))


def test_obj_reader_matches_oracle(tmp_path, mc, O):
    """v / vt / vn records, all four face forms, negative indices, polygons, blank lines, comments."""
    rs = np.random.RandomState(2)
    lines = ["# test mesh", "o thing", ""]
    for i in range(40):
        lines.append("v %.6f %.6f %r" % tuple(rs.uniform(-1, 1, 3)))
    for i in range(12):
        lines.append("vt %.5f %.5f" % tuple(rs.uniform(0, 1, 2)))
    for i in range(9):
        lines.append("vn %.4f %.4f %.4f" % tuple(rs.normal(size=3)))
    lines += ["s off", "usemtl none"]
    for i in range(60):
        k = rs.randint(3, 7)
        vs = rs.randint(1, 41, k)
        form = i % 3
        toks = []
        for v in vs:
            t, n = rs.randint(1, 13), rs.randint(1, 10)
            if i % 10 == 9:
                toks.append("%d/%d/%d" % (v - 41, t - 13, n - 10))          # relative indices
            else:
                toks.append("%d/%d/%d" % (v, t, n))
        lines.append(("\tf " if i % 7 == 0 else "f ") + "  ".join(toks) + ("\r" if i % 5 == 0 else ""))
    path = tmp_path / "mesh.obj"
    path.write_text("\n".join(lines) + "\n")
    mesh = mc.Mesh.from_obj(path).optimize().build_meshlets()
    ref = O.load_obj(path)
    _compare(mesh, ref)
    assert len(mesh.indices) > 300
    # position-only OBJ with "v" faces and "v//n"-less forms
    p2 = tmp_path / "plain.obj"
    p2.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0.5\nf 1 2 3\nf 2 4 3\nf -4 -3 -1\n")
    _compare(mc.Mesh.from_obj(p2).optimize().build_meshlets(), O.load_obj(p2))
    with pytest.raises(mc.McullError) as e:
        mc.Mesh.from_obj(tmp_path / "missing.obj")
    assert e.value.status == mc.E_IO
    p3 = tmp_path / "bad.obj"
    p3.write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(mc.McullError) as e:
        mc.Mesh.from_obj(p3)
    assert e.value.status == mc.E_INVALID


@pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="the reference's OBJ assets are not on this machine")
@pytest.mark.parametrize("name", ["kitten", "bunny", "viking_room"])
def test_reference_assets_reproduce_the_committed_fixtures(mc, name):
    """The reference's own assets through the product pipeline equal the committed scene fixtures
    (vertices, packed meshlet data, meshlet headers)."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + "_scene.npz"))
    mesh = mc.Mesh.from_obj(os.path.join(REF_ASSETS, name + ".obj")).optimize().build_meshlets()
    assert mesh.vertices.tobytes() == z["vertices"].tobytes()
    assert np.array_equal(mesh.meshlet_data, z["meshlet_data"])
    assert _headers(mesh.meshlets) == _headers(z["meshlets"].view(mc.MESHLET))


def test_mesh_save_round_trip(tmp_path, mc, H):
    pos, tris = H.proxy_mesh(30, 20, seed=1)
    mesh = mc.Mesh.from_arrays(pos, _corners(tris)).optimize().build_meshlets()
    mesh.save(tmp_path / "m.mcs")
    got = mc.load_scene(tmp_path / "m.mcs")
    assert got["vertices"].tobytes() == mesh.vertices.tobytes() and np.array_equal(got["indices"], mesh.indices)
    assert got["meshlets"].tobytes() == mesh.meshlets.tobytes() and np.array_equal(got["meshlet_data"], mesh.meshlet_data)
