"""The occlusion half of the instance-level pre-cull: an instance is dropped before the meshlet pass when the
occlusion test certainly rejects every one of its meshlets - its depth bound lies behind the minimum of the
pyramid texels that cover every texel any of its meshlets could select.  Masks, lists and commands must stay
bit-identical to the oracle's (which tests every pair), on depth images built so that a bound that is too
small shows: checkerboards whose far squares must not be missed, meshlets large enough to select coarse
levels, image sizes whose levels drop odd columns and rows, both projections, NaNs in the depth image."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["tiles", "rows"])
def pctx(mc, monkeypatch, request):
    """"rows": every pyramid goes through the streaming scheme (normally used from 6 M texels on), which also
    records the largest depth of every 64 x 8 strip of the image; the per-pair occlusion test then skips the
    texel read of pairs that are in front of their whole strip.  Results must not depend on it."""
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    monkeypatch.setenv("MCULL_PRECULL_ADAPTIVE", "0")
    if request.param == "rows":
        monkeypatch.setenv("MCULL_HIZ_STREAM", "1")
    c = mc.Context(0)
    monkeypatch.delenv("MCULL_PRECULL_MIN_GROUPS")
    monkeypatch.delenv("MCULL_PRECULL_ADAPTIVE")
    monkeypatch.delenv("MCULL_HIZ_STREAM", raising=False)
    yield c
    c.close()


def compare(mc, ctx, ref, n_inst):
    assert ctx.download_count() == ref.total
    assert np.array_equal(ctx.download_mask(), ref.mask)
    assert ctx.download_list().tobytes() == ref.list.tobytes()
    cmds = ctx.download_commands(n_inst).view(mc.CMD).reshape(-1)
    assert np.array_equal(cmds["taskCount"], ref.commands["taskCount"])
    assert np.array_equal(cmds["firstTask"], ref.commands["firstTask"])


def scene(H, n_inst, per, seed, radius=(0.01, 0.05), scales=(0.3, 2.0), extent=25.0):
    rs = np.random.RandomState(seed)
    cone, sphere = H.synthetic_bounds(n_inst * per, seed=seed + 1)
    sphere = sphere.copy()
    sphere[:, 3] = rs.uniform(radius[0], radius[1], n_inst * per).astype("<f4")
    t = H.synthetic_transforms(n_inst, extent=extent, seed=seed + 2)
    m = t["model"].reshape(n_inst, 4, 4).copy()
    m[:, :3, :3] *= rs.uniform(scales[0], scales[1], n_inst).astype("<f4")[:, None, None]
    t["model"] = m.reshape(n_inst, 16)
    off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
    cnt = np.full(n_inst, per, dtype="<u4")
    return cone, sphere, t, off, cnt


def checkerboard(w, h, block, near=0.3, far=0.0):
    yy, xx = np.mgrid[0:h, 0:w]
    return np.where(((xx // block) + (yy // block)) % 2 == 0, near, far).astype("<f4")


def run(mc, O, pctx, g, view, cone, sphere, t, off, cnt, tests, hiz_view=0):
    ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=tests)
    pctx.cull(g, tests=tests, hiz_view=hiz_view, write_mask=True)
    compare(mc, pctx, ref, len(t))
    marked, tested, per_group = pctx.precull_stats()
    groups = (cnt.astype(np.int64) + 63) // 64
    assert per_group and tested == int(groups.sum())
    hidden = int(groups[ref.commands["taskCount"] == 0].sum())
    assert marked <= hidden, "a dropped instance has no visible meshlet"
    pctx.cull(g, tests=tests, hiz_view=hiz_view, write_mask=True, precull=False)  # every pair through the per-pair test
    compare(mc, pctx, ref, len(t))
    return marked, hidden, ref


@pytest.mark.parametrize("size", [(1000, 562), (640, 360), (2047, 1153), (1001, 563), (1029, 517)])
def test_instances_behind_discs(mc, O, H, pctx, size):
    w, h = size
    cone, sphere, t, off, cnt = scene(H, 1500, 40, seed=100 + w)
    depth = H.synthetic_depth(w, h, n_discs=24, seed=w)
    depth[depth > 0] += np.float32(0.1)  # near enough that what lies behind is rejected (the test's bias is 0.05)
    pctx.upload_bounds_soa(cone, sphere)
    pctx.upload_transforms(t)
    pctx.set_instance_ranges(off, cnt)
    pctx.hiz_build(depth, view=0)
    pyr = O.hiz_build(depth)
    total_marked = 0
    for flip in (True, False):
        g = H.create_globals(eye=(40.0, 35.0, 30.0), width=w, height=h, flip_y=flip)
        view = O.View(g, pyr, w, h)
        for tests in (4, 7, 6, 5):
            marked, hidden, _ = run(mc, O, pctx, g, view, cone, sphere, t, off, cnt, tests)
            if tests == 4:
                assert marked > 0.3 * hidden > 0, (marked, hidden)  # the bound is not vacuous
            total_marked += marked
    assert total_marked > 0


@pytest.mark.parametrize("block", [3, 16, 100, 256])
@pytest.mark.parametrize("size", [(1024, 512), (1000, 562), (1003, 571)])
def test_checkerboard_depth_never_hides_a_far_square(mc, O, H, pctx, size, block):
    """Near and far squares alternate: an instance may be dropped only when every texel its meshlets can select
    - at every level they can select - lies in near squares.  Large meshlet radii make them select levels up to
    6 or so with the unflipped projection, where a texel's footprint is many pixels wide."""
    w, h = size
    depth = checkerboard(w, h, block)
    pyr = O.hiz_build(depth)
    pctx.hiz_build(depth, view=2)
    marked_sum = 0
    for k, (radius, scales) in enumerate([((0.01, 0.03), (0.2, 0.6)), ((0.1, 0.6), (0.5, 3.0))]):
        cone, sphere, t, off, cnt = scene(H, 1200, 24, seed=7 * block + k, radius=radius, scales=scales, extent=20.0)
        pctx.upload_bounds_soa(cone, sphere)
        pctx.upload_transforms(t)
        pctx.set_instance_ranges(off, cnt)
        for flip in (False, True):
            g = H.create_globals(eye=(34.0, 30.0, 26.0), width=w, height=h, flip_y=flip)
            marked, hidden, ref = run(mc, O, pctx, g, O.View(g, pyr, w, h), cone, sphere, t, off, cnt, 4, hiz_view=2)
            marked_sum += marked
            if block == 3:
                assert marked == 0, "no instance fits inside a 3-pixel square"
    if block == 256:
        assert marked_sum > 0


def test_uniform_scene_and_device_globals(mc, O, H, pctx, kitten):
    import torch

    ml = kitten["meshlets"]
    n, w, h = 400, 800, 600
    t = H.synthetic_transforms(n, extent=14.0, seed=12)
    depth = H.synthetic_depth(w, h, n_discs=14, seed=5)
    depth[depth > 0] += np.float32(0.15)
    depth[:, : w // 2] = 0.35  # a near wall over the left half
    pctx.set_instance_ranges(None, None)
    pctx.upload_meshlets(ml)
    pctx.upload_transforms(t)
    pctx.hiz_build(depth, view=0)
    pyr = O.hiz_build(depth)
    for flip in (True, False):
        g = H.create_globals(eye=(22.0, 20.0, 16.0), width=w, height=h, flip_y=flip, meshlet_count=len(ml))
        ref = O.cull_scene(O.View(g, pyr, w, h), t, ml, tests=7)
        ref_hiz = O.cull_scene(O.View(g, pyr, w, h), t, ml, tests=4)
        pctx.cull(g, tests=4, write_mask=True)
        compare(mc, pctx, ref_hiz, n)
        marked = pctx.precull_stats()[0]
        assert 0 < marked <= int((ref_hiz.commands["taskCount"] == 0).sum())
        pctx.cull(g, tests=7, write_mask=True)
        compare(mc, pctx, ref, n)
        marked7 = pctx.precull_stats()[0]
        g_dev = torch.from_numpy(np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1).view("u1").reshape(-1).copy()).to("cuda:0")
        torch.cuda.synchronize()
        pctx.cull_device_globals(g_dev.data_ptr(), len(ml), tests=7, write_mask=True)
        compare(mc, pctx, ref, n)
        assert pctx.precull_stats()[0] == marked7
        pctx.cull(g, tests=7, write_mask=True, precull=False)
        compare(mc, pctx, ref, n)


def test_a_nan_in_the_depth_image_switches_the_occlusion_level_off(mc, O, H, pctx):
    """min(x, y) = y < x ? y : x can drop a NaN between levels, so a coarse texel says nothing about the fine
    ones under it: a slot whose pyramid was built from an image with a NaN runs without the instance-level
    occlusion test - until the slot is built again from an image without one."""
    w, h = 512, 512
    cone, sphere, t, off, cnt = scene(H, 800, 32, seed=55, extent=18.0)
    depth = np.full((h, w), 0.4, dtype="<f4")  # everything behind a near wall ...
    holes = depth.copy()
    rs = np.random.RandomState(3)
    ys, xs = rs.randint(0, h, 4000), rs.randint(0, w, 4000)
    holes[ys, xs] = np.nan  # ... except through NaN texels: nz < NaN - 0.05 is false, the pair is visible
    g = H.create_globals(eye=(30.0, 26.0, 22.0), width=w, height=h, flip_y=False)
    pctx.upload_bounds_soa(cone, sphere)
    pctx.upload_transforms(t)
    pctx.set_instance_ranges(off, cnt)
    # a clean wall in slot 1: most instances are dropped whole
    pctx.hiz_build(depth, view=1)
    marked, hidden, ref = run(mc, O, pctx, g, O.View(g, O.hiz_build(depth), w, h), cone, sphere, t, off, cnt, 4, hiz_view=1)
    assert marked > 0.5 * hidden > 0
    clean_marked = marked
    with np.errstate(all="ignore"):
        for _ in range(2):  # NaN image -> off, clean rebuild -> on again, twice over
            # the wall with NaN holes in slot 0: identical to the oracle, nothing dropped by occlusion
            pctx.hiz_build(holes, view=0)
            marked, hidden, ref = run(mc, O, pctx, g, O.View(g, O.hiz_build(holes), w, h), cone, sphere, t, off, cnt, 4, hiz_view=0)
            assert marked == 0 and ref.total > 0
            # slot 1 is unaffected
            marked, _, _ = run(mc, O, pctx, g, O.View(g, O.hiz_build(depth), w, h), cone, sphere, t, off, cnt, 4, hiz_view=1)
            assert marked == clean_marked
            # a clean rebuild of slot 0: the word describes the image the pyramid was LAST built from
            pctx.hiz_build(depth, view=0)
            marked, _, _ = run(mc, O, pctx, g, O.View(g, O.hiz_build(depth), w, h), cone, sphere, t, off, cnt, 4, hiz_view=0)
            assert marked == clean_marked


@pytest.mark.parametrize("size", [(64, 48), (100, 100), (3, 200), (512, 512), (1200, 700)])
@pytest.mark.parametrize("scheme", ["MCULL_HIZ_TILES", "MCULL_HIZ_STREAM", "MCULL_HIZ_SINGLE_LAUNCH", None])
def test_the_nan_word_follows_the_last_build_in_every_pyramid_scheme(mc, O, H, monkeypatch, size, scheme):
    """Every scheme and every image size class (one workgroup, tiles without a second launch, tiles + top levels,
    streaming pass) must publish the word from inside its own launches: a NaN image, then a clean one, then a NaN one."""
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    monkeypatch.setenv("MCULL_PRECULL_ADAPTIVE", "0")
    if scheme:
        monkeypatch.setenv(scheme, "1")
    w, h = size
    with mc.Context(0) as ctx:
        cone, sphere, t, off, cnt = scene(H, 300, 16, seed=5, extent=10.0)
        ctx.upload_bounds_soa(cone, sphere)
        ctx.upload_transforms(t)
        ctx.set_instance_ranges(off, cnt)
        g = H.create_globals(eye=(18.0, 16.0, 14.0), width=w, height=h, flip_y=False)
        wall = np.full((h, w), 0.5, dtype="<f4")
        bad = wall.copy()
        bad[h // 2, w // 3] = np.nan
        ref = O.cull_scene(O.View(g, O.hiz_build(wall), w, h), t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=4)
        seen = []
        with np.errstate(all="ignore"):
            for img in (bad, wall, bad, wall):
                ctx.hiz_build(img, view=3)
                ctx.cull(g, tests=4, hiz_view=3, write_mask=True)
                seen.append(ctx.precull_stats()[0])
                if img is wall:
                    compare(mc, ctx, ref, len(t))
        assert seen[0] == 0 and seen[2] == 0, seen
        assert seen[1] == seen[3], seen
        if min(w, h) >= 48:  # (a sliver of an image offers no rectangle to prove anything with)
            assert seen[1] > 0, seen


def test_negative_radii_and_hostile_transforms(mc, O, H, pctx):
    w, h = 640, 400
    cone, sphere, t, off, cnt = scene(H, 600, 30, seed=77, extent=15.0)
    sphere[30 * 5 + 3, 3] = -0.02   # a negative radius: the sign of the projected radius differs inside the range
    sphere[30 * 9: 30 * 10, 3] *= -1.0
    m = t["model"].reshape(len(t), 4, 4).copy()
    m[11, 0, 0] = np.nan
    m[12, 3, 2] = np.inf
    m[13, :3, :3] = 0.0
    m[14, :3, :3] *= np.float32(-1.5)
    m[15, 1, :3] += np.float32(0.7) * m[15, 0, :3]
    m[16, 3, :3] = [30.0, 26.0, 22.0]  # at the eye: cw around zero
    t["model"] = m.reshape(len(t), 16)
    depth = np.full((h, w), 0.5, dtype="<f4")
    depth[:, : w // 3] = 0.0
    pctx.upload_bounds_soa(cone, sphere)
    pctx.upload_transforms(t)
    pctx.set_instance_ranges(off, cnt)
    pctx.hiz_build(depth, view=0)
    pyr = O.hiz_build(depth)
    with np.errstate(all="ignore"):
        for flip in (False, True):
            g = H.create_globals(eye=(30.0, 26.0, 22.0), width=w, height=h, flip_y=flip)
            for tests in (4, 7):
                marked, hidden, _ = run(mc, O, pctx, g, O.View(g, pyr, w, h), cone, sphere, t, off, cnt, tests)
                assert marked > 0


@pytest.mark.parametrize("depth_kind", ["discs", "checker", "far"])
def test_several_views_with_the_instance_level(mc, O, H, pctx, depth_kind):
    """mcull_cull_views classifies every instance per view first (dropped / needs no frustum test / needs no occlusion
    test) and the pass over the bounds leaves those tests out: identical outputs for every view, whatever mix of
    verdicts the views produce ("far": a depth image nothing is ever behind, every instance in the open)."""
    import torch

    w, h, nv = 1024, 640, 6
    cone, sphere, t, off, cnt = scene(H, 900, 48, seed=31, extent=22.0)
    pctx.upload_bounds_soa(cone, sphere)
    pctx.upload_transforms(t)
    pctx.set_instance_ranges(off, cnt)
    views, params, refs = [], [], []
    for v in range(nv):
        if depth_kind == "discs":
            depth = H.synthetic_depth(w, h, n_discs=16, seed=70 + v)
            depth[depth > 0] += np.float32(0.12)
        elif depth_kind == "checker":
            depth = checkerboard(w, h, [8, 64, 200, 16, 300, 128][v])
        else:
            depth = np.full((h, w), 0.04, dtype="<f4")
        ang = 2 * np.pi * v / nv
        g = H.create_globals(eye=(38 * np.cos(ang), 38 * np.sin(ang), 8.0 + 5 * v), fov_deg=25.0 + 8 * v, width=w, height=h,
                             flip_y=bool(v & 1))
        pctx.hiz_build(depth, view=v)
        tests = [7, 7, 5, 6, 3, 4][v]
        views.append(g)
        params.append(dict(tests=tests, hiz_view=v, write_mask=True))
        refs.append(O.cull_scene(O.View(g, O.hiz_build(depth), w, h), t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt),
                                 tests=tests))
    pctx.cull_views(views, params)
    for v in range(nv):
        pctx.select_result(v)
        compare(mc, pctx, refs[v], len(t))
    packed = np.concatenate([np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1) for g in views])
    g_dev = torch.from_numpy(packed.view("u1").reshape(-1).copy()).to("cuda:0")
    torch.cuda.synchronize()
    pctx.cull_views(nv, params, device_globals_ptr=g_dev.data_ptr(), meshlet_count=0)
    for v in range(nv):
        pctx.select_result(v)
        compare(mc, pctx, refs[v], len(t))
    off_params = [dict(p, precull=False) for p in params]
    pctx.cull_views(views, off_params)
    for v in range(nv):
        pctx.select_result(v)
        compare(mc, pctx, refs[v], len(t))


@pytest.mark.parametrize("size", [(1280, 720), (1029, 517), (1543, 1031)])
def test_large_instances(mc, O, H, pctx, size):
    """Instances that cover hundreds of pixels: their screen rectangles are taller than eight 64 x 8 strips, so the
    "nothing of it is occluded" verdict comes from the 64 x 64 block maxima (1029 x 517: the last columns and rows
    of the image lie outside what the blocks cover, instances there get no verdict).  Both kinds of depth image:
    discs (some instances in the open, some straddling, some hidden) and a far plane (all in the open)."""
    w, h = size
    cone, sphere, t, off, cnt = scene(H, 500, 64, seed=200 + w, radius=(0.02, 0.2), scales=(2.0, 9.0), extent=24.0)
    pctx.upload_bounds_soa(cone, sphere)
    pctx.upload_transforms(t)
    pctx.set_instance_ranges(off, cnt)
    for kind in ("discs", "far", "wall"):
        if kind == "discs":
            depth = H.synthetic_depth(w, h, n_discs=10, seed=h)
            depth[depth > 0] += np.float32(0.1)
        elif kind == "far":
            depth = np.full((h, w), 0.03, dtype="<f4")
        else:
            depth = np.full((h, w), 0.02, dtype="<f4")
            depth[:, w // 2:] = 0.6
        pctx.hiz_build(depth, view=3)
        pyr = O.hiz_build(depth)
        for flip in (False, True):
            g = H.create_globals(eye=(45.0, 40.0, 33.0), width=w, height=h, flip_y=flip, fov_deg=50.0)
            for tests in (7, 4, 5):
                run(mc, O, pctx, g, O.View(g, pyr, w, h), cone, sphere, t, off, cnt, tests, hiz_view=3)


@pytest.mark.parametrize("switches", [{"MCULL_PRECULL_FLAGS": "0"}, {"MCULL_HIZ_STRIPS": "0", "MCULL_HIZ_STREAM": "1"},
                                      {"MCULL_PRECULL_ADAPTIVE": "1"}], ids=["no-verdicts", "no-strips", "adaptive"])
def test_evaluation_switches_change_nothing(mc, O, H, monkeypatch, switches):
    """The switches that take parts of the instance level away (measurements) leave every output as it is."""
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    monkeypatch.setenv("MCULL_PRECULL_ADAPTIVE", "0")
    for k, v in switches.items():
        monkeypatch.setenv(k, v)
    w, h = 900, 520
    cone, sphere, t, off, cnt = scene(H, 700, 40, seed=404, scales=(0.3, 4.0), extent=20.0)
    depth = H.synthetic_depth(w, h, n_discs=12, seed=9)
    depth[depth > 0] += np.float32(0.1)
    pyr = O.hiz_build(depth)
    with mc.Context(0) as c:
        c.upload_bounds_soa(cone, sphere)
        c.upload_transforms(t)
        c.set_instance_ranges(off, cnt)
        views, params, refs = [], [], []
        for v in range(3):
            c.hiz_build(depth, view=v)
            g = H.create_globals(eye=(36.0 - 10 * v, 30.0, 24.0 + 6 * v), width=w, height=h, flip_y=bool(v & 1))
            views.append(g)
            params.append(dict(tests=[7, 5, 6][v], hiz_view=v, write_mask=True))
            refs.append(O.cull_scene(O.View(g, pyr, w, h), t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=[7, 5, 6][v]))
        for rep in range(3):  # repeated: the adaptive variant decides from the previous cull
            for v in range(3):
                c.cull(views[v], **params[v])
                compare(mc, c, refs[v], len(t))
            c.cull_views(views, params)
            for v in range(3):
                c.select_result(v)
                compare(mc, c, refs[v], len(t))
