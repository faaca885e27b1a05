"""The instance-level pre-cull in front of the meshlet pass: one enclosing sphere per instance range against
the five planes, with a margin that covers the rounding of the per-meshlet expression.  It may only ever
remove instances whose every meshlet the per-meshlet frustum test removes, so masks, lists and commands
must stay bit-identical to the oracle's (which has no such level) - on scenes built to sit on the margin,
with hostile values, sheared and mirrored transforms, several views, and both kernel families.
MCULL_PRECULL_MIN_GROUPS=1 switches the pre-cull on for scenes of any size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[False, True], ids=["resident", "streaming"])
def pctx(mc, monkeypatch, request):
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    monkeypatch.setenv("MCULL_PRECULL_ADAPTIVE", "0")  # every cull runs the pre-cull, whatever the last one removed
    if request.param:
        monkeypatch.setenv("MCULL_FORCE_STREAMING", "1")
    c = mc.Context(0)
    monkeypatch.delenv("MCULL_PRECULL_MIN_GROUPS")
    monkeypatch.delenv("MCULL_PRECULL_ADAPTIVE")
    monkeypatch.delenv("MCULL_FORCE_STREAMING", raising=False)
    yield c
    c.close()


def compare(mc, ctx, ref, n_inst):
    assert ctx.download_count() == ref.total
    assert np.array_equal(ctx.download_mask(), ref.mask)
    assert ctx.download_list().tobytes() == ref.list.tobytes()
    cmds = ctx.download_commands(n_inst).view(mc.CMD).reshape(-1)
    assert np.array_equal(cmds["taskCount"], ref.commands["taskCount"])
    assert np.array_equal(cmds["firstTask"], ref.commands["firstTask"])


def plane_hugging_transforms(H, g, n, radius, rs):
    """Rigid and non-rigid transforms whose instance spheres sit within +-1e-3 of touching a frustum plane from
    outside: where the pre-cull's margin decides."""
    t = H.synthetic_transforms(n, extent=1.0, seed=int(rs.randint(1 << 30)))
    m = t["model"].reshape(n, 4, 4).copy()
    planes = np.asarray(g["frustum"], dtype=np.float64).reshape(5, 4)
    eye = np.array([2.0, 2.0, 2.0])
    for i in range(n):
        p = planes[i % 5]
        # a point inside the frustum near the view axis, pushed out through plane p by radius + delta
        base = eye * (1.0 - rs.uniform(0.3, 3.0)) + rs.normal(0, 0.3, 3)
        dist = base @ p[:3] + p[3]
        delta = rs.uniform(-1e-3, 1e-3) if i % 3 else rs.uniform(-1e-6, 1e-6)
        centre = base - p[:3] * (dist + radius + delta)
        k = i % 7
        if k == 3:
            m[i, :3, :3] *= np.float32(rs.uniform(0.2, 3.0))          # uniform scale
        elif k == 4:
            m[i, 0, :3] *= np.float32(2.5); m[i, 1, :3] *= np.float32(0.3)  # non-uniform: |col 0| is not the largest stretch
        elif k == 5:
            m[i, 1, :3] += np.float32(0.8) * m[i, 0, :3]              # shear
        elif k == 6:
            m[i, 2, :3] *= np.float32(-1.0)                            # mirrored
        m[i, 3, :3] = centre.astype(np.float32)
    t["model"] = m.reshape(n, 16)
    return t


@pytest.mark.parametrize("tests", [7, 3, 1, 5])
def test_instances_on_the_margin(mc, O, H, pctx, kitten, tests):
    ml = kitten["meshlets"]
    centres = ml["coneApex"][:, :3].astype(np.float64)
    c0 = centres.mean(0)
    radius = float((np.linalg.norm(centres - c0, axis=1) + ml["radius"]).max())
    rs = np.random.RandomState(5 + tests)
    w, h = 640, 480
    g = H.create_globals(width=w, height=h, meshlet_count=len(ml))
    n = 420
    t = plane_hugging_transforms(H, g, n, radius, rs)
    depth = H.synthetic_depth(w, h, n_discs=10, seed=3)
    pctx.upload_meshlets(ml)
    pctx.upload_transforms(t)
    pctx.hiz_build(depth, view=0)
    ref = O.cull_scene(O.View(g, O.hiz_build(depth), w, h), t, ml, tests=tests)
    pctx.cull(g, tests=tests, write_mask=True)
    compare(mc, pctx, ref, n)
    marked, tested, per_group = pctx.precull_stats()
    assert tested == n and not per_group
    if tests & 1:
        assert 0 < marked < n, "the scene is built so that the pre-cull takes some instances and leaves others"
        per_inst = ref.commands["taskCount"]
        assert marked <= int((per_inst == 0).sum()), "a marked instance has no visible meshlet"
    else:
        assert marked == 0
    # switched off: identical outputs, nothing marked
    pctx.cull(g, tests=tests, write_mask=True, precull=False)
    compare(mc, pctx, ref, n)
    assert pctx.precull_stats()[0] == 0


def test_hostile_values_and_ragged_ranges(mc, O, H, pctx):
    rs = np.random.RandomState(91)
    n_inst, per = 160, 200
    cone, sphere = H.synthetic_bounds(n_inst * per, seed=31)
    cone, sphere = cone.copy(), sphere.copy()
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-40, 3.4e38, -3.4e38, 1e19, -1e19, 1.0, 0.5], dtype="<f4")
    # hostile bounds in a few instances only (they make those ranges' spheres infinite: never pre-culled)
    for i in rs.choice(n_inst, 12, replace=False):
        rows = i * per + rs.choice(per, 6, replace=False)
        for r in rows:
            (cone if rs.randint(2) else sphere)[r, rs.randint(0, 4)] = special[rs.randint(len(special))]
    t = H.synthetic_transforms(n_inst, extent=30.0, seed=33)
    m = t["model"].reshape(n_inst, 4, 4).copy()
    m[3, :3, :3] = 0.0
    m[7, 0, 0] = np.nan
    m[8, 3, 1] = np.nan
    m[11, :3, :3] *= np.float32(-2.5)
    m[19, 3, 0] = np.inf
    m[20, 3, 2] = -np.inf
    m[21, 1, 1] = 3.0e38
    m[22, 3, :3] = [1e30, -1e30, 1e30]
    m[23, 0, :3] *= np.float32(1e-30)
    t["model"] = m.reshape(n_inst, 16)
    off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
    cnt = np.full(n_inst, per, dtype="<u4")
    cnt[5], cnt[50], cnt[100], cnt[101] = 0, 1, 65, 64
    w, h = 320, 200
    depth = H.synthetic_depth(w, h, n_discs=12, seed=14)
    pctx.upload_bounds_soa(cone, sphere)
    pctx.upload_transforms(t)
    pctx.set_instance_ranges(off, cnt)
    pctx.hiz_build(depth, view=1)
    pyr = O.hiz_build(depth)
    marked_any = 0
    with np.errstate(all="ignore"):
        for flip in (True, False):
            g = H.create_globals(eye=(40.0, 35.0, 30.0), width=w, height=h, flip_y=flip)
            view = O.View(g, pyr, w, h)
            for tests in (7, 3, 1):
                ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=tests)
                pctx.cull(g, tests=tests, hiz_view=1, write_mask=True)
                compare(mc, pctx, ref, n_inst)
                marked, tested, per_group = pctx.precull_stats()
                assert per_group and tested == int(((cnt.astype(np.int64) + 63) // 64).sum())
                marked_any += marked
    assert marked_any > 0


def test_views_and_device_globals(mc, O, H, pctx, kitten):
    """The multi-view pass has no instance level (it made that pass slower); the single-view cull with its
    Globals in device memory has, from the planes it reads on the device."""
    import torch

    ml = kitten["meshlets"]
    n, w, h, nv = 200, 640, 360, 6
    t = H.synthetic_transforms(n, extent=12.0, seed=8)
    pctx.upload_meshlets(ml)
    pctx.upload_transforms(t)
    views, params, refs = [], [], []
    for v in range(nv):
        ang = 2 * np.pi * v / nv
        g = H.create_globals(eye=(9 * np.cos(ang), 9 * np.sin(ang), 3.0 + v), fov_deg=30.0 + 5 * v, width=w, height=h,
                             flip_y=bool(v & 1), meshlet_count=len(ml))
        depth = H.synthetic_depth(w, h, n_discs=20, seed=60 + v)
        pctx.hiz_build(depth, view=v)
        tests = [7, 3, 5, 1, 6, 2][v]
        views.append(g)
        params.append(dict(tests=tests, hiz_view=v, write_mask=True))
        refs.append(O.cull_scene(O.View(g, O.hiz_build(depth), w, h), t, ml, tests=tests))
    pctx.cull_views(views, params)
    for v in range(nv):
        pctx.select_result(v)
        compare(mc, pctx, refs[v], n)
        assert pctx.precull_stats()[:2] == (0, 0)
    packed = np.concatenate([np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1) for g in views])
    g_dev = torch.from_numpy(packed.view("u1").reshape(-1).copy()).to("cuda:0")
    torch.cuda.synchronize()
    for v in (0, 1, 3):
        pctx.cull(views[v], **params[v])
        compare(mc, pctx, refs[v], n)
        marked = pctx.precull_stats()[0]
        assert marked > 0
        pctx.cull_device_globals(g_dev.data_ptr() + v * mc.GLOBALS.itemsize, len(ml), **params[v])
        compare(mc, pctx, refs[v], n)
        assert pctx.precull_stats()[0] == marked


def test_spheres_follow_the_bounds_and_the_ranges(mc, O, H, pctx, kitten, bunny):
    """The range spheres are recomputed when the meshlets or the ranges change (a stale sphere would drop
    visible instances)."""
    n = 300
    t = H.synthetic_transforms(n, extent=6.0, seed=4)
    g = H.create_globals(eye=(7.0, 6.0, 5.0), meshlet_count=0)
    pctx.upload_transforms(t)
    for scene in (kitten, bunny, kitten):
        ml = scene["meshlets"]
        g["meshletCount"] = len(ml)
        pctx.set_instance_ranges(None, None)
        pctx.upload_meshlets(ml)
        pctx.cull(g, tests=3, write_mask=True)
        compare(mc, pctx, O.cull_scene(O.View(g), t, ml, tests=3), n)
        # a sub-range of the same array: another sphere
        g2 = g.copy()
        g2["meshletCount"] = 70
        pctx.cull(g2, tests=3, write_mask=True)
        compare(mc, pctx, O.cull_scene(O.View(g2), t, ml[:70], tests=3), n)
        off = np.where(np.arange(n) % 2 == 0, 0, 64).astype("<u4")
        cnt = np.where(np.arange(n) % 2 == 0, 64, len(ml) - 64).astype("<u4")
        pctx.set_instance_ranges(off, cnt)
        pctx.cull(g, tests=3, write_mask=True)
        compare(mc, pctx, O.cull_scene(O.View(g), t, ml, ranges=(off, cnt), tests=3), n)


def test_a_precull_that_removes_nothing_backs_off(mc, O, H, kitten, monkeypatch, forced):
    """The default context drops the pre-cull for a while after one that kept more than 90 % of the units, and
    tries again later; the results never change."""
    if not forced.adaptive:
        pytest.skip("the run's switches take the back-off away")
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    ml = kitten["meshlets"]
    n = 64
    t = H.synthetic_transforms(n, extent=1.0, seed=2)
    g_all = H.create_globals(eye=(6.0, 6.0, 6.0), meshlet_count=len(ml))                           # every instance in view
    g_none = H.create_globals(eye=(6.0, 6.0, 6.0), centre=(12.0, 12.0, 12.0), meshlet_count=len(ml))  # looking away
    ref_all = O.cull_scene(O.View(g_all), t, ml, tests=3)
    ref_none = O.cull_scene(O.View(g_none), t, ml, tests=3)
    assert ref_none.total == 0 < ref_all.total
    with mc.Context(0) as c:
        c.upload_meshlets(ml)
        c.upload_transforms(t)
        ran = []
        for k in range(40):
            c.cull(g_all, tests=3, write_mask=True)
            compare(mc, c, ref_all, n)  # synchronises: the next call sees what this cull kept
            marked, tested, _ = c.precull_stats()
            assert marked == 0
            ran.append(tested > 0)
        assert ran[0] and not any(ran[1:16]) and ran[16] and not any(ran[17:32]) and ran[32], ran
        # a camera for which it pays keeps it switched on once it is tried again
        ran = []
        for k in range(40):
            c.cull(g_none, tests=3, write_mask=True)
            compare(mc, c, ref_none, n)
            ran.append(c.precull_stats()[1] > 0)
        first = ran.index(True)
        assert first <= 15 and all(ran[first:]), ran


@pytest.mark.parametrize("estimate", [1, 300, 100000])
def test_grid_sized_for_an_estimate_of_the_list(mc, O, H, kitten, monkeypatch, estimate):
    """The streaming mask pass sizes its grid for the list length an earlier cull reported and leaves what lies
    beyond to a second, walking launch: with the estimate forced far too low (1 group), in the middle of the list
    and above the whole scene, ranged and uniform scenes give the oracle's outputs."""
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    monkeypatch.setenv("MCULL_PRECULL_ADAPTIVE", "0")
    monkeypatch.setenv("MCULL_FORCE_STREAMING", "1")
    monkeypatch.setenv("MCULL_PRECULL_ESTIMATE", str(estimate))
    with mc.Context(0) as c:
        for k in ("MCULL_PRECULL_MIN_GROUPS", "MCULL_PRECULL_ADAPTIVE", "MCULL_FORCE_STREAMING", "MCULL_PRECULL_ESTIMATE"):
            monkeypatch.delenv(k)
        # uniform scene: every instance draws the kitten (6 groups each), entries are instances
        ml = kitten["meshlets"]
        n = 400
        w, h = 640, 480
        g = H.create_globals(eye=(25.0, 20.0, 18.0), width=w, height=h, meshlet_count=len(ml))
        t = H.synthetic_transforms(n, extent=30.0, seed=77)
        depth = H.synthetic_depth(w, h, n_discs=10, seed=3)
        c.upload_meshlets(ml)
        c.upload_transforms(t)
        c.hiz_build(depth, view=0)
        view = O.View(g, O.hiz_build(depth), w, h)
        for tests in (7, 3):
            ref = O.cull_scene(view, t, ml, tests=tests)
            for _ in range(2):  # the second cull meets the first one's list contents
                c.cull(g, tests=tests, write_mask=True)
                compare(mc, c, ref, n)
            marked, tested, per_group = c.precull_stats()
            assert tested == n and 0 < marked < n
            # the same with the Globals read on the device (the kernels that take their view constants from memory)
            import torch

            g_dev = torch.from_numpy(np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1).view("u1").reshape(-1).copy()).to("cuda:0")
            torch.cuda.synchronize()
            c.cull_device_globals(g_dev.data_ptr(), len(ml), tests=tests, write_mask=True)
            compare(mc, c, ref, n)
            assert c.precull_stats()[0] == marked
        # ranged scene: ragged and empty ranges, entries are groups with their records
        rs = np.random.RandomState(4)
        n_inst, per = 300, 200
        cone, sphere = H.synthetic_bounds(n_inst * per, seed=8)
        t2 = H.synthetic_transforms(n_inst, extent=30.0, seed=9)
        off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
        cnt = rs.randint(0, per + 1, n_inst).astype("<u4")
        cnt[3], cnt[4], cnt[200] = 0, 1, 65
        c.upload_bounds_soa(cone, sphere)
        c.upload_transforms(t2)
        c.set_instance_ranges(off, cnt)
        g2 = H.create_globals(eye=(40.0, 35.0, 30.0), width=w, height=h)
        view2 = O.View(g2, O.hiz_build(depth), w, h)
        for tests in (7, 1):
            ref = O.cull_scene(view2, t2, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=tests)
            c.cull(g2, tests=tests, write_mask=True)
            compare(mc, c, ref, n_inst)
            marked, tested, per_group = c.precull_stats()
            assert per_group and 0 < marked < tested


def test_alternating_test_sets_keep_their_own_list_estimates(mc, O, H, monkeypatch):
    """The two phases of a frame alternate on one context: frustum + cone (a long work list), then all three tests (a short
    one).  Each set of tests has its own estimate of the list's length, takes its turn at the probe that measures it, and
    the estimate-sized mask pass, its walking second launch and the scatter pass's form for sparse lists give the oracle's
    outputs frame after frame."""
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    monkeypatch.setenv("MCULL_FORCE_STREAMING", "1")  # the kernels that size their grid for the estimate
    with mc.Context(0) as c:
        monkeypatch.delenv("MCULL_PRECULL_MIN_GROUPS")
        monkeypatch.delenv("MCULL_FORCE_STREAMING")
        n_inst, per = 3000, 256
        cone, sphere = H.synthetic_bounds(n_inst * per, seed=31)
        t = H.synthetic_transforms(n_inst, extent=60.0, seed=32)
        off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
        cnt = np.full(n_inst, per, dtype="<u4")
        w, h = 640, 360
        depth = H.synthetic_depth(w, h, n_discs=30, seed=33)
        depth[:, : w // 2] = 0.9  # a wall over half the image: most of what the planes keep is hidden
        g = H.create_globals(eye=(70.0, 60.0, 50.0), width=w, height=h)
        c.upload_bounds_soa(cone, sphere)
        c.upload_transforms(t)
        c.set_instance_ranges(off, cnt)
        c.hiz_build(depth, view=0)
        view = O.View(g, O.hiz_build(depth), w, h)
        refs = {k: O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=k) for k in (3, 7, 1)}
        assert refs[7].total * 3 < refs[3].total, "the scene should separate the two lists"
        for frame in range(8):
            for k in (3, 7) if frame != 5 else (1, 7, 3):  # a third set of tests turns up once
                c.cull(g, tests=k, hiz_view=0, write_mask=True)
                compare(mc, c, refs[k], n_inst)


def test_the_back_off_of_one_test_set_leaves_the_other_alone(mc, O, H, kitten, monkeypatch, forced):
    """Every instance in view and hidden behind a wall: the planes remove nothing, so the frustum + cone culls drop the
    instance level for a while - the culls with the occlusion test, which alternate with them on the same context and
    whose instance level removes nearly everything, keep it."""
    if not forced.adaptive:
        pytest.skip("the run's switches take the back-off away")
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    ml = kitten["meshlets"]
    n = 64
    w, h = 320, 240
    t = H.synthetic_transforms(n, extent=1.0, seed=2)
    g = H.create_globals(eye=(6.0, 6.0, 6.0), width=w, height=h, meshlet_count=len(ml))
    depth = np.full((h, w), 0.9, dtype="<f4")  # reverse-Z: nearer than everything
    view = O.View(g, O.hiz_build(depth), w, h)
    ref3 = O.cull_scene(O.View(g), t, ml, tests=3)
    ref7 = O.cull_scene(view, t, ml, tests=7)
    assert ref7.total == 0 < ref3.total
    with mc.Context(0) as c:
        c.upload_meshlets(ml)
        c.upload_transforms(t)
        c.hiz_build(depth, view=0)
        ran3, ran7 = [], []
        for frame in range(20):
            c.cull(g, tests=3, write_mask=True)
            compare(mc, c, ref3, n)
            ran3.append(c.precull_stats()[1] > 0)
            c.cull(g, tests=7, hiz_view=0, write_mask=True)
            compare(mc, c, ref7, n)
            marked, tested, _ = c.precull_stats()
            ran7.append(tested > 0 and marked == tested)
        assert ran3[0] and not any(ran3[2:12]), ran3   # backed off (the probe's answer may take a frame to be looked at)
        assert all(ran7), ran7


def test_new_matrices_for_the_same_instances_keep_the_hints_and_the_results_right(mc, O, H, monkeypatch):
    """An animated scene uploads new matrices for the same instances every frame: what the last frame's pre-cull reported
    stays the guess for the next one (a different instance count starts from scratch).  Frames that drift, a frame in which
    nearly everything leaves the view and the frame after it, in which everything is back - the estimate far too long, then
    far too short: the oracle's outputs every time."""
    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    monkeypatch.setenv("MCULL_FORCE_STREAMING", "1")
    with mc.Context(0) as c:
        monkeypatch.delenv("MCULL_PRECULL_MIN_GROUPS")
        monkeypatch.delenv("MCULL_FORCE_STREAMING")
        n_inst, per = 2500, 256
        cone, sphere = H.synthetic_bounds(n_inst * per, seed=41)
        off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
        cnt = np.full(n_inst, per, dtype="<u4")
        g = H.create_globals(eye=(70.0, 60.0, 50.0), width=640, height=360)
        view = O.View(g)
        c.upload_bounds_soa(cone, sphere)
        base = H.synthetic_transforms(n_inst, extent=60.0, seed=42)
        c.upload_transforms(base)
        c.set_instance_ranges(off, cnt)
        totals = []
        for frame in range(10):
            t = base.copy()
            m = t["model"].reshape(n_inst, 4, 4).copy()
            m[:, 3, :3] += np.float32(0.5 * frame)               # drift
            if frame == 6:
                m[: n_inst - 20, 3, :3] += np.float32(5000.0)    # nearly everything far outside the planes
            t["model"] = m.reshape(n_inst, 16)
            c.upload_transforms(t)
            ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=3)
            for _ in range(2):
                c.cull(g, tests=3, write_mask=True)
                compare(mc, c, ref, n_inst)
            totals.append(ref.total)
        assert totals[6] * 20 < totals[5] and totals[7] > totals[6] * 20
        c.upload_transforms(base[: n_inst // 2])                 # another count: from scratch
        c.set_instance_ranges(off[: n_inst // 2], cnt[: n_inst // 2])
        ref = O.cull_scene(view, base[: n_inst // 2], cone_axis_w=cone, centre_radius=sphere, ranges=(off[: n_inst // 2], cnt[: n_inst // 2]), tests=3)
        c.cull(g, tests=3, write_mask=True)
        compare(mc, c, ref, n_inst // 2)


def test_a_captured_cull_ignores_the_back_off_and_survives_a_wrong_estimate(mc, O, H, monkeypatch, forced):
    """A cull captured into a graph is replayed for cameras the capturing call never saw.  Captured while the instance
    level is in its back-off (the camera of the capture sees everything) and with the list estimate of a camera that sees
    almost nothing, the replays still run the instance level and give the oracle's outputs for a camera that sees
    everything, one that sees nothing and one in between."""
    import torch

    monkeypatch.setenv("MCULL_PRECULL_MIN_GROUPS", "1")
    monkeypatch.setenv("MCULL_FORCE_STREAMING", "1")
    stream = torch.cuda.Stream()
    with mc.Context(0, stream=stream.cuda_stream) as c, torch.cuda.stream(stream):
        monkeypatch.delenv("MCULL_PRECULL_MIN_GROUPS")
        monkeypatch.delenv("MCULL_FORCE_STREAMING")
        n_inst, per = 2500, 256
        cone, sphere = H.synthetic_bounds(n_inst * per, seed=51)
        t = H.synthetic_transforms(n_inst, extent=20.0, seed=52)
        off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
        cnt = np.full(n_inst, per, dtype="<u4")
        cams = {"all": H.create_globals(eye=(150.0, 140.0, 120.0), width=640, height=360),
                "none": H.create_globals(eye=(150.0, 140.0, 120.0), centre=(300.0, 280.0, 240.0), width=640, height=360),
                "some": H.create_globals(eye=(30.0, 5.0, 8.0), width=640, height=360)}
        refs = {k: O.cull_scene(O.View(g), t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=3) for k, g in cams.items()}
        assert refs["none"].total == 0 < refs["some"].total < refs["all"].total
        c.upload_bounds_soa(cone, sphere)
        c.upload_transforms(t)
        c.set_instance_ranges(off, cnt)
        g_dev = {k: torch.from_numpy(np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1).view("u1").reshape(-1).copy()).to("cuda:0")
                 for k, g in cams.items()}
        live = torch.empty_like(g_dev["all"])
        torch.cuda.synchronize()
        for k in ("none", "none", "none", "all", "all"):  # the estimate of "none", then the back-off that "all" earns
            live.copy_(g_dev[k])
            c.cull_device_globals(live.data_ptr(), per, tests=3, write_mask=True)
            compare(mc, c, refs[k], n_inst)
        if forced.adaptive:
            c.cull_device_globals(live.data_ptr(), per, tests=3, write_mask=True)
            c.synchronize()
            assert c.precull_stats()[1] == 0, "the capture below should fall into the back-off"
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            c.cull_device_globals(live.data_ptr(), per, tests=3, write_mask=True)
        for k in ("all", "none", "some", "all"):
            live.copy_(g_dev[k])
            graph.replay()
            stream.synchronize()
            compare(mc, c, refs[k], n_inst)
            marked, tested, _ = c.precull_stats()
            assert tested > 0 and (k != "none" or marked == tested), (k, marked, tested)
