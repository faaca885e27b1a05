"""GPU half of the mesh pipeline: mcull_mesh_upload completes the product-built meshlets with the
bounds kernel; the result equals the oracle's CPU pipeline bit for bit and culls identically."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
_SCALE = int(os.environ.get("MCULL_FUZZ_SCALE", "1"))  # a longer soak than the default run


def test_mesh_upload_completes_bounds_and_culls(mc, O, H, ctx):
    pos, tris = H.proxy_mesh(120, 100, seed=8)
    cor = np.full((tris.size, 3), -1, dtype="<i4")
    cor[:, 0] = tris.reshape(-1)
    rs = np.random.RandomState(1)
    nrm = rs.normal(size=(len(pos), 3)).astype("<f4")
    cor[:, 2] = cor[:, 0]
    mesh = mc.Mesh.from_arrays(pos, cor, normals=nrm).optimize().build_meshlets()
    ref = O.build_mesh(pos, cor, nrm=nrm)
    ctx.set_instance_ranges(None, None)
    ctx.upload_mesh(mesh)
    assert mesh.meshlets.tobytes() == ref.meshlets.tobytes()      # cone, apex, radius from the GPU
    assert mesh.vertices.tobytes() == ref.vertices.tobytes() and np.array_equal(mesh.meshlet_data, ref.meshlet_data)
    n = 128
    t = H.update_transforms(n)
    g = H.create_globals(meshlet_count=len(ref.meshlets))
    ctx.upload_transforms(t)
    want = O.cull_scene(O.View(g), t, ref.meshlets, tests=3)
    ctx.cull(g, tests=3)
    assert ctx.download_list().tobytes() == want.list.tobytes()
    assert 0 < want.total < want.n_pairs


def _soup(seed, n_pos=300, n_tri=1500):
    rs = np.random.RandomState(seed)
    if seed >= 4:  # the soak's extra seeds also vary the sizes
        n_pos, n_tri = int(rs.choice([40, 300, 2000])), int(rs.choice([25, 700, 1500, 6000]))
    pos = rs.randint(-4, 5, (n_pos, 3)).astype("<f4") * 0.25
    pos[pos == 0] *= rs.choice([1.0, -1.0], size=(pos == 0).sum())
    if seed % 2:
        pos[7, 1] = np.nan
    nrm = rs.normal(size=(40, 3)).astype("<f4")
    nrm[5] = 0.0
    uv = rs.uniform(-2, 2, (50, 2)).astype("<f4")
    uv[3] = [70000.0, 1e-9]
    cor = np.empty((n_tri * 3, 3), dtype="<i4")
    cor[:, 0] = rs.randint(0, n_pos, n_tri * 3)
    cor[:, 1] = rs.randint(0, len(uv), n_tri * 3)
    cor[:, 2] = rs.randint(0, len(nrm), n_tri * 3)
    cor[30:33, 0] = cor[30, 0]
    cor[60, 0] = cor[61, 0]
    return pos, cor, nrm, uv


@pytest.mark.parametrize("seed", range(4 * _SCALE))
def test_gpu_dedup_and_meshlets_equal_the_host_pipeline_on_soups(mc, ctx, seed):
    """Seams, -0/+0, NaN positions, degenerate triangles, half overflow: the GPU dedup pass numbers
    vertices exactly like the sequential loop, and the GPU meshlet builder cuts the same meshlets."""
    pos, cor, nrm, uv = _soup(seed)
    for kwargs in ({"normals": nrm, "texcoords": uv}, {}):
        host = mc.Mesh.from_arrays(pos, cor, **kwargs)
        gpu = ctx.mesh_from_arrays(pos, cor, **kwargs)
        assert gpu.vertices.tobytes() == host.vertices.tobytes()
        assert np.array_equal(gpu.indices, host.indices)
        host.optimize().build_meshlets()
        gpu.optimize()
        ctx.build_meshlets(gpu)
        assert np.array_equal(gpu.indices, host.indices)
        assert np.array_equal(gpu.meshlet_data, host.meshlet_data)
        ctx.upload_mesh(host)                      # completes the host mesh's bounds on the GPU
        assert gpu.meshlets.tobytes() == host.meshlets.tobytes()
        assert seed >= 4 or (len(gpu.meshlets) > 10 and gpu.meshlets["radius"].any())


def test_gpu_mesh_front_half_large(mc, O, H, ctx):
    """The 871 200-triangle proxy: GPU dedup + host reorder + GPU meshlets + GPU bounds equals the
    oracle's CPU pipeline in every array."""
    pos, tris = H.proxy_mesh(660, 660, seed=12)
    cor = np.full((tris.size, 3), -1, dtype="<i4")
    cor[:, 0] = tris.reshape(-1)
    mesh = ctx.mesh_from_arrays(pos, cor).optimize()
    ctx.build_meshlets(mesh)
    ref = O.build_mesh(pos, cor)
    assert mesh.vertices.tobytes() == ref.vertices.tobytes()
    assert np.array_equal(mesh.indices, ref.indices)
    assert np.array_equal(mesh.meshlet_data, ref.meshlet_data)
    assert mesh.meshlets.tobytes() == ref.meshlets.tobytes()
    assert len(ref.meshlets) == 13916


def test_gpu_mesh_errors(mc, ctx):
    pos = np.zeros((3, 3), "<f4")
    bad = np.array([[0, -1, -1], [1, -1, -1], [5, -1, -1]], dtype="<i4")
    with pytest.raises(mc.McullError) as e:
        ctx.mesh_from_arrays(pos, bad)
    assert e.value.status == mc.E_INVALID
    empty = ctx.mesh_from_arrays(pos, np.zeros((0, 3), "<i4"))
    ctx.build_meshlets(empty)
    assert len(empty.meshlets) == 0 and len(empty.vertices) == 0


def test_bounds_with_adversarial_vertices(mc, O, ctx, kitten):
    """buildMeshletCon on the GPU with NaN / infinite / denormal / huge positions and degenerate
    triangles: cone, apex and radius still equal the scalar CPU loop bit for bit, a NaN matching any
    NaN."""
    rs = np.random.RandomState(3)
    verts = kitten["vertices"].copy()
    special = np.array([np.nan, np.inf, -np.inf, 3.0e38, -3.0e38, 1e-40, -1e-40, 0.0, -0.0], dtype="<f4")
    for i in rs.choice(len(verts), 400, replace=False):
        verts["pos"][i, rs.randint(0, 3)] = special[rs.randint(len(special))]
    for i in rs.choice(len(verts) - 1, 200, replace=False):   # coincident vertices -> zero-area triangles
        verts["pos"][i] = verts["pos"][i + 1]
    blank = kitten["meshlets"].copy()
    blank["cone"] = 0; blank["coneApex"] = 0; blank["radius"] = 0
    with np.errstate(all="ignore"):
        want = O.build_meshlet_cones(verts, kitten["meshlet_data"], blank)
    ctx.upload_meshlets(blank)
    ctx.upload_geometry(verts, kitten["meshlet_data"])
    ctx.compute_bounds()
    got = ctx.download_meshlets(len(blank))
    def same(a, b):  # NaN equals NaN (sign and payload of a NaN are not part of the contract); everything else bit for bit
        a, b = np.asarray(a, dtype="<f4").reshape(-1), np.asarray(b, dtype="<f4").reshape(-1)
        return bool(np.all((a.view("<u4") == b.view("<u4")) | (np.isnan(a) & np.isnan(b))))

    bad = [(i, f) for i in range(len(blank)) for f in ("cone", "coneApex", "radius") if not same(got[i][f], want[i][f])]
    assert not bad, (bad[:8], [(got[i][f], want[i][f]) for i, f in bad[:4]])
    n_nan = sum(1 for i in range(len(blank)) if np.isnan(want[i]["cone"]).any() or np.isnan(want[i]["radius"]))
    assert n_nan > 20
