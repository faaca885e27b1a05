"""The kernels the library picks for bounds arrays larger than the caches (non-temporal loads, two
groups per step on packed binary32 instructions) are normally reached only from 2 M meshlets up,
where the oracle can check samples at best.  MCULL_FORCE_STREAMING=1 selects them for any size, so
the small scenes with every test combination, hostile values and several views run through them
too, bit for bit against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def sctx(mc, monkeypatch):
    monkeypatch.setenv("MCULL_FORCE_STREAMING", "1")
    c = mc.Context(0)
    monkeypatch.delenv("MCULL_FORCE_STREAMING")
    yield c
    c.close()


def compare(mc, ctx, ref, n_inst):
    assert ctx.download_count() == ref.total
    assert np.array_equal(ctx.download_mask(), ref.mask)
    assert ctx.download_list().tobytes() == ref.list.tobytes()
    cmds = ctx.download_commands(n_inst).view(mc.CMD).reshape(-1)
    assert np.array_equal(cmds["taskCount"], ref.commands["taskCount"])
    assert np.array_equal(cmds["firstTask"], ref.commands["firstTask"])


def test_hostile_values_through_the_streaming_kernels(mc, O, H, sctx):
    rs = np.random.RandomState(77)
    n_inst, per = 101, 192  # an odd number of groups: the last step of the last wave has a single group
    cone, sphere = H.synthetic_bounds(n_inst * per, seed=17)
    cone, sphere = cone.copy(), sphere.copy()
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-40, -1e-40, 3.4e38, -3.4e38, 1.0, -1.0, 0.5], dtype="<f4")
    idx = rs.choice(n_inst * per, 5000, replace=False)
    for k, i in enumerate(idx):
        (cone if k % 2 == 0 else sphere)[i, rs.randint(0, 4)] = special[rs.randint(len(special))]
    cone[idx[:300], :3] = 0.0
    t = H.synthetic_transforms(n_inst, extent=20.0, seed=15)
    m = t["model"].reshape(n_inst, 4, 4).copy()
    m[3, :3, :3] = 0.0
    m[7, 0, 0] = np.nan
    m[11, :3, :3] *= np.float32(-2.5)
    m[13, 3, :3] = [2.0, 2.0, 2.0]
    m[19, 3, 0] = np.inf
    t["model"] = m.reshape(n_inst, 16)
    off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
    cnt = np.full(n_inst, per, dtype="<u4")
    cnt[5], cnt[50], cnt[100] = 0, 1, 65  # empty, single-pair and ragged instances
    w, h = 320, 200
    depth = H.synthetic_depth(w, h, n_discs=12, seed=14)
    depth.flat[rs.randint(0, w * h, 50)] = np.nan
    sctx.upload_bounds_soa(cone, sphere)
    sctx.upload_transforms(t)
    sctx.set_instance_ranges(off, cnt)
    sctx.hiz_build(depth, view=2)
    pyr = O.hiz_build(depth)
    with np.errstate(all="ignore"):
        for flip in (True, False):
            g = H.create_globals(eye=(2.0, 2.0, 2.0), width=w, height=h, flip_y=flip)
            view = O.View(g, pyr, w, h)
            for tests in (7, 5, 6, 3, 4, 2, 1, 0):
                ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=tests)
                sctx.cull(g, tests=tests, hiz_view=2, write_mask=True)
                compare(mc, sctx, ref, n_inst)


def test_views_through_the_streaming_kernels(mc, O, H, sctx, kitten):
    ml = kitten["meshlets"]
    n, w, h = 150, 640, 360
    t = H.update_transforms(n)
    views, params, refs = [], [], []
    for v in range(6):
        ang = 2 * np.pi * v / 6
        g = H.create_globals(eye=(9 * np.cos(ang), 9 * np.sin(ang), 3.0 + v), fov_deg=40.0 + 5 * v, width=w, height=h,
                             flip_y=bool(v & 1), meshlet_count=len(ml))
        depth = H.synthetic_depth(w, h, n_discs=20, seed=60 + v)
        sctx.hiz_build(depth, view=v)
        tests = [7, 3, 5, 1, 6, 2][v]
        views.append(g)
        params.append(dict(tests=tests, hiz_view=v, write_mask=True))
        refs.append(O.cull_scene(O.View(g, O.hiz_build(depth), w, h), t, ml, tests=tests))
    sctx.upload_meshlets(ml)
    sctx.upload_transforms(t)
    sctx.cull_views(views, params)
    for v in range(6):
        sctx.select_result(v)
        compare(mc, sctx, refs[v], n)
    for v in range(6):  # and one view at a time
        sctx.cull(views[v], **params[v])
        compare(mc, sctx, refs[v], n)


@pytest.mark.parametrize("streaming", [False, True])
def test_views_with_globals_in_device_memory(mc, O, H, ctx, sctx, kitten, streaming):
    """mcull_cull_views_device_globals: the views' Globals sit in one device array and are read when the
    stream gets to the call; refreshing the array and calling again gives the new cameras' results."""
    import torch

    c = sctx if streaming else ctx
    ml = kitten["meshlets"]
    n, w, h, nv = 120, 640, 360, 5
    t = H.update_transforms(n)
    c.set_instance_ranges(None, None)
    c.upload_meshlets(ml)
    c.upload_transforms(t)
    params, pyrs = [], []
    for v in range(nv):
        depth = H.synthetic_depth(w, h, n_discs=20, seed=80 + v)
        c.hiz_build(depth, view=v)
        pyrs.append(O.hiz_build(depth))
        params.append(dict(tests=[7, 3, 5, 6, 1][v], hiz_view=v, write_mask=True))
    g_dev = torch.zeros(nv * mc.GLOBALS.itemsize, dtype=torch.uint8, device="cuda:0")
    for frame in range(2):
        views = []
        for v in range(nv):
            ang = 2 * np.pi * v / nv + 0.4 * frame
            views.append(H.create_globals(eye=(9 * np.cos(ang), 9 * np.sin(ang), 3.0 + v), fov_deg=40.0 + 5 * v, width=w,
                                          height=h, flip_y=bool(v & 1), meshlet_count=len(ml)))
        packed = np.concatenate([np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1) for g in views])
        g_dev.copy_(torch.from_numpy(packed.view("u1").reshape(-1).copy()))
        torch.cuda.synchronize()
        c.cull_views(nv, params, device_globals_ptr=g_dev.data_ptr(), meshlet_count=len(ml))
        for v in range(nv):
            c.select_result(v)
            compare(mc, c, O.cull_scene(O.View(views[v], pyrs[v], w, h), t, ml, tests=params[v]["tests"]), n)
