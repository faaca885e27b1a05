"""GPU half of the mesh pipeline: mcull_mesh_upload completes the product-built meshlets with the
bounds kernel; the result equals the oracle's CPU pipeline bit for bit and culls identically."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
