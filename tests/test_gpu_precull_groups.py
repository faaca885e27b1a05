"""The group level of the pre-cull: in scenes whose instances span 32 and more 64-meshlet groups (one large mesh drawn
many times) every group of an instance that survived the instance level is tested once more with an enclosing sphere
of its own - dropped, or told that it needs no frustum / no occlusion test - before any of its meshlets is loaded.
Masks, lists and commands must stay bit-identical to the oracle's, which tests every pair; the meshes here are long
ribbons, so that an instance straddles the frustum's planes and the occluders' edges while its groups do not."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def gctx(mc, monkeypatch):
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    monkeypatch.setenv("MCULL_PRECULL_ADAPTIVE", "0")
    c = mc.Context(0)
    monkeypatch.delenv("MCULL_PRECULL_MIN_GROUPS")
    monkeypatch.delenv("MCULL_PRECULL_ADAPTIVE")
    yield c
    c.close()


def compare(mc, ctx, ref, n_inst):
    assert ctx.download_count() == ref.total
    assert np.array_equal(ctx.download_mask(), ref.mask)
    assert ctx.download_list().tobytes() == ref.list.tobytes()
    cmds = ctx.download_commands(n_inst).view(mc.CMD).reshape(-1)
    assert np.array_equal(cmds["taskCount"], ref.commands["taskCount"])
    assert np.array_equal(cmds["firstTask"], ref.commands["firstTask"])


def ribbon(H, n, seed, length=30.0, radius=(0.05, 0.2)):
    """n meshlets along a spiral: consecutive meshlets are neighbours in space, as a meshletiser leaves them."""
    rs = np.random.RandomState(seed)
    cone, sphere = H.synthetic_bounds(n, seed=seed)
    sphere = sphere.copy()
    s = np.arange(n, dtype=np.float64) / n
    turns = 3.0
    sphere[:, 0] = (4.0 * np.cos(2 * np.pi * turns * s) + rs.normal(0, 0.05, n)).astype("<f4")
    sphere[:, 1] = (4.0 * np.sin(2 * np.pi * turns * s) + rs.normal(0, 0.05, n)).astype("<f4")
    sphere[:, 2] = ((s - 0.5) * length + rs.normal(0, 0.05, n)).astype("<f4")
    sphere[:, 3] = rs.uniform(radius[0], radius[1], n).astype("<f4")
    return cone, sphere


def transforms(H, n_inst, seed, extent=30.0, scales=(0.3, 1.2)):
    rs = np.random.RandomState(seed)
    t = H.synthetic_transforms(n_inst, extent=extent, seed=seed + 2)
    m = t["model"].reshape(n_inst, 4, 4).copy()
    m[:, :3, :3] *= rs.uniform(scales[0], scales[1], n_inst).astype("<f4")[:, None, None]
    t["model"] = m.reshape(n_inst, 16)
    return t


def occluders(H, w, h, seed):
    depth = H.synthetic_depth(w, h, n_discs=10, seed=seed)
    depth[depth > 0] += np.float32(0.1)
    depth[:, : w // 3] = 0.3  # a near wall over the left third
    return depth


def check(mc, O, c, g, view, cone, sphere, t, off, cnt, tests, uniform, **kw):
    from conftest import FORCED

    ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=tests)
    c.cull(g, tests=tests, write_mask=True, **kw)
    compare(mc, c, ref, len(t))
    dropped, tested = c.precull_group_stats()
    marked, units, per_group = c.precull_stats()
    groups = (cnt.astype(np.int64) + 63) // 64
    if not FORCED.group_level:  # the run was started with the level switched off: the outputs above are all there is to check
        assert (dropped, tested) == (0, 0)
    elif tests & 5:
        kept_groups = int(groups.sum()) - marked if per_group else int(groups[0]) * (units - marked)
        assert tested == kept_groups, "the group level looks at every group the instance level kept"
        assert dropped <= int((ref.mask == 0).sum()), "a dropped group has no visible meshlet"
    else:
        assert (dropped, tested) == (0, 0)
    assert per_group == (not uniform)
    c.cull(g, tests=tests, write_mask=True, precull=False, **kw)
    compare(mc, c, ref, len(t))
    assert c.precull_group_stats() == (0, 0)
    return dropped, tested, ref


@pytest.mark.parametrize("uniform", [True, False], ids=["uniform", "ranged"])
@pytest.mark.parametrize("size", [(1280, 720), (1001, 563)])
def test_ribbons_behind_discs_and_across_the_frustum(mc, O, H, gctx, size, uniform, forced):
    import torch

    w, h = size
    n_inst = 150
    if uniform:
        per = 2500  # 40 groups, the last one short
        cone, sphere = ribbon(H, per, seed=w)
        off, cnt = np.zeros(n_inst, dtype="<u4"), np.full(n_inst, per, dtype="<u4")
    else:
        a, b = ribbon(H, 2560, seed=w), ribbon(H, 2100, seed=w + 1, length=18.0, radius=(0.02, 0.4))
        cone, sphere = np.concatenate([a[0], b[0]]), np.concatenate([a[1], b[1]])
        pick = np.arange(n_inst) % 3 == 0
        off = np.where(pick, 2560, 0).astype("<u4")
        cnt = np.where(pick, 2100, 2560).astype("<u4")
        off[7], cnt[7] = 100, 2400  # a range that starts inside a mesh: its groups are not the mesh's own
        off[8], cnt[8] = 2500, 2160  # and one that spans both meshes
    t = transforms(H, n_inst, seed=h)
    depth = occluders(H, w, h, seed=w)
    gctx.upload_bounds_soa(cone, sphere)
    gctx.upload_transforms(t)
    gctx.set_instance_ranges(None, None) if uniform else gctx.set_instance_ranges(off, cnt)
    gctx.hiz_build(depth, view=0)
    pyr = O.hiz_build(depth)
    dropped_by = {}
    for flip in (True, False):
        g = H.create_globals(eye=(38.0, 33.0, 28.0), width=w, height=h, flip_y=flip, meshlet_count=int(cnt[0]))
        view = O.View(g, pyr, w, h)
        for tests in (7, 4, 1, 5, 6, 3, 2):
            dropped, tested, ref = check(mc, O, gctx, g, view, cone, sphere, t, off, cnt, tests, uniform)
            dropped_by[tests] = dropped_by.get(tests, 0) + dropped
        # the same with the view constants derived on the device
        ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=7)
        g_dev = torch.from_numpy(np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1).view("u1").reshape(-1).copy()).to("cuda:0")
        torch.cuda.synchronize()
        gctx.cull_device_globals(g_dev.data_ptr(), int(cnt[0]), tests=7, write_mask=True)
        compare(mc, gctx, ref, n_inst)
        assert gctx.precull_group_stats()[0] > 0 or not forced.group_level
    # the level is not vacuous: each of the two proofs removes groups the instance level had to keep
    if forced.group_level:
        assert dropped_by[1] > 0 and dropped_by[4] > 0 and dropped_by[7] >= max(dropped_by[1], dropped_by[4])


def test_hostile_groups(mc, O, H, gctx):
    """Negative and non-finite radii inside some groups, hostile matrices, an instance at the eye."""
    w, h = 800, 480
    per, n_inst = 2304, 90  # 36 groups
    cone, sphere = ribbon(H, per, seed=5)
    sphere[64 * 3 + 5, 3] = -0.1          # one negative radius: that group proves nothing about the frustum's inside
    sphere[64 * 9: 64 * 10, 3] *= -1.0    # a whole group of them
    sphere[64 * 12 + 1, 0] = np.nan       # a centre that is not a number
    sphere[64 * 20 + 63, 3] = np.inf
    sphere[64 * 21, 3] = 1.0e20
    t = transforms(H, n_inst, seed=9, extent=22.0)
    m = t["model"].reshape(n_inst, 4, 4).copy()
    m[11, 0, 0] = np.nan
    m[12, 3, 2] = np.inf
    m[13, :3, :3] = 0.0
    m[14, :3, :3] *= np.float32(-1.5)
    m[15, 1, :3] += np.float32(0.7) * m[15, 0, :3]
    m[16, 3, :3] = [30.0, 26.0, 22.0]  # at the eye
    m[17, :3, :3] *= np.float32(1.0e-20)
    t["model"] = m.reshape(n_inst, 16)
    off, cnt = np.zeros(n_inst, dtype="<u4"), np.full(n_inst, per, dtype="<u4")
    depth = occluders(H, w, h, seed=2)
    gctx.upload_bounds_soa(cone, sphere)
    gctx.upload_transforms(t)
    gctx.set_instance_ranges(None, None)
    gctx.hiz_build(depth, view=0)
    pyr = O.hiz_build(depth)
    with np.errstate(all="ignore"):
        for flip in (False, True):
            g = H.create_globals(eye=(30.0, 26.0, 22.0), width=w, height=h, flip_y=flip, meshlet_count=per)
            for tests in (7, 4, 1):
                check(mc, O, gctx, g, O.View(g, pyr, w, h), cone, sphere, t, off, cnt, tests, True)
        gctx.set_instance_ranges(off, cnt)  # the same scene described by ranges
        g = H.create_globals(eye=(30.0, 26.0, 22.0), width=w, height=h, flip_y=True)
        check(mc, O, gctx, g, O.View(g, pyr, w, h), cone, sphere, t, off, cnt, 7, False)


@pytest.mark.parametrize("kind", ["far", "wall", "checker", "nan"])
def test_depth_images_that_settle_everything_or_nothing(mc, O, H, gctx, kind):
    """A far plane (every group in the open: no occlusion test left), a near wall (every group hidden: an empty list),
    a fine checkerboard (no group fits a near square: nothing dropped by occlusion) and a wall with NaN holes (the
    occlusion half is off for that pyramid)."""
    w, h = 1024, 640
    per, n_inst = 2560, 80
    cone, sphere = ribbon(H, per, seed=21, radius=(0.02, 0.1))
    t = transforms(H, n_inst, seed=33, extent=16.0, scales=(0.2, 0.5))
    off, cnt = np.zeros(n_inst, dtype="<u4"), np.full(n_inst, per, dtype="<u4")
    if kind == "far":
        depth = np.full((h, w), 0.001, dtype="<f4")
    elif kind == "wall":
        depth = np.full((h, w), 0.9, dtype="<f4")
    elif kind == "checker":
        yy, xx = np.mgrid[0:h, 0:w]
        depth = np.where(((xx // 3) + (yy // 3)) % 2 == 0, 0.9, 0.0).astype("<f4")
    else:
        depth = np.full((h, w), 0.9, dtype="<f4")
        rs = np.random.RandomState(4)
        depth[rs.randint(0, h, 3000), rs.randint(0, w, 3000)] = np.nan
    gctx.upload_bounds_soa(cone, sphere)
    gctx.upload_transforms(t)
    gctx.set_instance_ranges(None, None)
    with np.errstate(all="ignore"):
        gctx.hiz_build(depth, view=1)
        pyr = O.hiz_build(depth)
        for flip in (False, True):
            g = H.create_globals(eye=(24.0, 20.0, 17.0), width=w, height=h, flip_y=flip, meshlet_count=per)
            dropped, tested, ref = check(mc, O, gctx, g, O.View(g, pyr, w, h), cone, sphere, t, off, cnt, 4, True, hiz_view=1)
            if kind == "far":
                assert dropped == 0 and ref.total == n_inst * per
            elif kind == "wall":
                assert ref.total == 0
            elif kind in ("checker", "nan"):
                assert dropped == 0
            check(mc, O, gctx, g, O.View(g, pyr, w, h), cone, sphere, t, off, cnt, 7, True, hiz_view=1)


def test_the_switch_takes_the_level_away(mc, O, H, monkeypatch):
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    monkeypatch.setenv("MCULL_PRECULL_ADAPTIVE", "0")
    monkeypatch.setenv("MCULL_PRECULL_GROUPS", "0")
    w, h = 900, 520
    per, n_inst = 2560, 100
    cone, sphere = ribbon(H, per, seed=8)
    t = transforms(H, n_inst, seed=3)
    off, cnt = np.zeros(n_inst, dtype="<u4"), np.full(n_inst, per, dtype="<u4")
    depth = occluders(H, w, h, seed=6)
    pyr = O.hiz_build(depth)
    g = H.create_globals(eye=(38.0, 33.0, 28.0), width=w, height=h, flip_y=True, meshlet_count=per)
    ref = O.cull_scene(O.View(g, pyr, w, h), t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=7)
    with mc.Context(0) as c:
        c.upload_bounds_soa(cone, sphere)
        c.upload_transforms(t)
        c.hiz_build(depth, view=0)
        c.cull(g, tests=7, write_mask=True)
        compare(mc, c, ref, n_inst)
        assert c.precull_group_stats() == (0, 0) and c.precull_stats()[1] == n_inst


def test_a_group_level_that_removes_nothing_backs_off(mc, O, H, monkeypatch, forced):
    """A group level that hands on (nearly) everything it was given costs its pass for nothing: after a cull that
    reported so it is left out for a while - the instance level stays - and the outputs are the same either way."""
    if not (forced.group_level and forced.adaptive):
        pytest.skip("the run's switches take the group level or its back-off away")
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    with mc.Context(0) as c:
        monkeypatch.delenv("MCULL_PRECULL_MIN_GROUPS")
        n_inst, per = 48, 64 * 40
        # a compact mesh (unrelated meshlets in a unit cube: a group's sphere is the instance's) drawn half in front of
        # the camera and half behind it: the instance level removes the half behind, the group level nothing
        cone, sphere = H.synthetic_bounds(per, seed=3)
        t = H.synthetic_transforms(n_inst, extent=4.0, seed=21)
        m = t["model"].reshape(n_inst, 4, 4).copy()
        m[: n_inst // 2, 3, :3] += np.float32(200.0)  # far behind the camera at (20, 20, 20) looking at the origin
        t["model"] = m.reshape(n_inst, 16)
        w, h = 640, 480
        g = H.create_globals(eye=(20.0, 20.0, 20.0), width=w, height=h, meshlet_count=per)
        c.upload_bounds_soa(cone, sphere)
        c.upload_transforms(t)
        ref = O.cull_scene(O.View(g, None, w, h), t, cone_axis_w=cone, centre_radius=sphere, tests=3)
        ran, skipped = 0, 0
        for k in range(6):
            c.cull(g, tests=3, write_mask=True)
            compare(mc, c, ref, n_inst)  # (the downloads wait for the cull, so the next one sees its report)
            marked = c.precull_stats()[0]
            assert marked >= n_inst // 2, "the instance level keeps running"
            dropped, tested = c.precull_group_stats()
            if tested:
                ran += 1
                assert dropped * 10 < tested, "the scene is built so that the group level finds next to nothing"
            else:
                skipped += 1
        assert ran >= 1 and skipped >= 3, (ran, skipped)


def test_window_tables_follow_the_pyramid_also_under_a_replayed_graph(mc, O, H, gctx, forced):
    """The two levels prove occlusion from sliding-window minima that a cull derives from the pyramid when it finds them
    out of date.  A rebuilt pyramid must never be culled against tables of the image before it: neither when the library
    sees the rebuild (it makes new tables) nor when a captured cull is replayed after a rebuild it knows nothing of (the
    build has cleared the word that says the tables are complete, and the replayed kernels do without them)."""
    import torch

    w, h = 1280, 720
    per, n_inst = 2560, 120
    cone, sphere = ribbon(H, per, seed=31)
    t = transforms(H, n_inst, seed=17)
    off, cnt = np.zeros(n_inst, dtype="<u4"), np.full(n_inst, per, dtype="<u4")
    walls = np.full((h, w), 0.9, dtype="<f4")          # everything hidden
    holes = walls.copy()
    holes[:, w // 2:] = 0.0                            # ... except in the right half
    discs = occluders(H, w, h, seed=9)
    g = H.create_globals(eye=(38.0, 33.0, 28.0), width=w, height=h, flip_y=True, meshlet_count=per)
    gctx.upload_bounds_soa(cone, sphere)
    gctx.upload_transforms(t)
    gctx.set_instance_ranges(None, None)
    refs = {}
    for name, depth in (("walls", walls), ("holes", holes), ("discs", discs)):
        refs[name] = O.cull_scene(O.View(g, O.hiz_build(depth), w, h), t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=7)
    assert refs["walls"].total == 0 < refs["holes"].total
    # rebuilds the library sees: every cull is checked against the image its pyramid was last built from
    dropped = {}
    for name, depth in (("walls", walls), ("holes", holes), ("discs", discs), ("walls", walls), ("discs", discs)):
        gctx.hiz_build(depth, view=0)
        for _ in range(2):  # the second cull finds the tables made by the first
            gctx.cull(g, tests=7, write_mask=True)
            compare(mc, gctx, refs[name], n_inst)
        dropped[name] = gctx.precull_stats()[0]
    assert dropped["walls"] > dropped["holes"], "a wall hides more instances than half a wall"
    # a captured cull (view constants and depth image in device memory) replayed after rebuilds outside the graph
    stream = torch.cuda.Stream()
    with mc.Context(0, stream=stream.cuda_stream) as c, torch.cuda.stream(stream):
        c.upload_bounds_soa(cone, sphere)
        c.upload_transforms(t)
        g_dev = torch.from_numpy(np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1).view("u1").reshape(-1).copy()).to("cuda:0")
        d_dev = {name: torch.from_numpy(depth).to("cuda:0") for name, depth in (("walls", walls), ("holes", holes), ("discs", discs))}
        torch.cuda.synchronize()
        c.hiz_build(d_dev["walls"].data_ptr(), view=0, width=w, height=h, mem=mc.MEM_DEVICE)
        for _ in range(2):  # buffers allocated and tables current before the capture
            c.cull_device_globals(g_dev.data_ptr(), per, tests=7, write_mask=True)
        compare(mc, c, refs["walls"], n_inst)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            c.cull_device_globals(g_dev.data_ptr(), per, tests=7, write_mask=True)
        for name in ("holes", "discs", "walls", "holes"):
            c.hiz_build(d_dev[name].data_ptr(), view=0, width=w, height=h, mem=mc.MEM_DEVICE)
            graph.replay()
            stream.synchronize()
            compare(mc, c, refs[name], n_inst)
