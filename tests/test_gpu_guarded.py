"""The guarded-exact cone and occlusion tests: approximate reciprocals with error bounds first, the IEEE
expressions only where a bound could change a decision.  Every output must equal the oracle's (which evaluates
the IEEE expressions throughout) - on ordinary scenes, and on scenes built so that thousands of pairs sit
within a few ulps of each decision's boundary: the cone margin, a power-of-two pixel radius (mip selection),
the depth threshold.  MCULL_CULL_COUNT_SLOW_PAIRS shows that those pairs did take the IEEE path, and
MCULL_CULL_EXACT_ARITHMETIC (every pair through it) gives the same bytes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def compare(mc, ctx, ref, n_inst):
    assert ctx.download_count() == ref.total
    assert np.array_equal(ctx.download_mask(), ref.mask)
    assert ctx.download_list().tobytes() == ref.list.tobytes()
    cmds = ctx.download_commands(n_inst).view(mc.CMD).reshape(-1)
    assert np.array_equal(cmds["taskCount"], ref.commands["taskCount"])
    assert np.array_equal(cmds["firstTask"], ref.commands["firstTask"])


def ulps(a, k):
    """float32 array moved by k units in the last place (k: integer array)."""
    i = np.ascontiguousarray(a, dtype="<f4").view("<i4").astype(np.int64)
    i = np.where(i < 0, -(i & 0x7fffffff), i)  # sign-magnitude -> ordered integers
    i = i + k
    i = np.where(i < 0, (-i) | 0x80000000, i)
    return i.astype("<u4").view("<f4")


@pytest.fixture(params=[False, True], ids=["resident", "streaming"])
def gctx(mc, monkeypatch, request):
    if request.param:
        monkeypatch.setenv("MCULL_FORCE_STREAMING", "1")
    c = mc.Context(0)
    monkeypatch.delenv("MCULL_FORCE_STREAMING", raising=False)
    yield c
    c.close()


def run_all_forms(mc, ctx, g, ref, n_inst, tests, hiz_view=0):
    """fast forms (counted), all-IEEE form: both equal to the oracle.  Returns the slow-path count."""
    ctx.cull(g, tests=tests, hiz_view=hiz_view, write_mask=True, count_slow=True)
    compare(mc, ctx, ref, n_inst)
    slow = ctx.slow_path_pairs()
    ctx.cull(g, tests=tests, hiz_view=hiz_view, write_mask=True, exact=True, count_slow=True)
    compare(mc, ctx, ref, n_inst)
    assert ctx.slow_path_pairs() >= slow
    return slow


def test_config1_takes_the_slow_path_for_a_few_pairs(mc, O, H, gctx, kitten, config1):
    g, t = config1["globals"], config1["transforms"]
    depth = H.synthetic_depth(800, 600, n_discs=24, seed=5)
    ref = O.cull_scene(O.View(g, O.hiz_build(depth), 800, 600), t, kitten["meshlets"], tests=7)
    gctx.upload_meshlets(kitten["meshlets"])
    gctx.upload_transforms(t)
    gctx.hiz_build(depth, view=0)
    slow = run_all_forms(mc, gctx, g, ref, len(t), 7)
    n_pairs = len(t) * len(kitten["meshlets"])
    assert 0 < slow < n_pairs // 50, "a few pairs in a thousand sit inside a guard band, not more: %d of %d" % (slow, n_pairs)


def scene(H, n_inst, per, seed):
    cone, sphere = H.synthetic_bounds(n_inst * per, seed=seed)
    t = H.synthetic_transforms(n_inst, extent=20.0, seed=seed + 1)
    off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
    cnt = np.full(n_inst, per, dtype="<u4")
    return cone.copy(), sphere.copy(), t, off, cnt


def test_pairs_on_the_cone_boundary(mc, O, H, gctx):
    n_inst, per = 300, 128
    cone, sphere, t, off, cnt = scene(H, n_inst, per, 41)
    g = H.create_globals(eye=(40.0, 35.0, 30.0), fov_deg=50.0, width=512, height=256)
    view = O.View(g)
    base = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=2, want_margins=True)
    # coneW := -dot, i.e. margin 0 up to the rounding of the subtraction, then -6 .. +6 ulps around it
    rs = np.random.RandomState(1)
    dot = base.margins[:, 2].astype(np.float64) - cone[:, 3].astype(np.float64)
    cone[:, 3] = ulps((-dot).astype("<f4"), rs.randint(-6, 7, size=len(dot)))
    ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=2, want_margins=True)
    assert float(np.abs(ref.margins[:, 2]).max()) < 1e-5 and 0.2 < ref.total / ref.n_pairs < 0.8
    gctx.upload_bounds_soa(cone, sphere)
    gctx.upload_transforms(t)
    gctx.set_instance_ranges(off, cnt)
    slow = run_all_forms(mc, gctx, g, ref, n_inst, 2)
    assert slow == ref.n_pairs, "every pair sits inside the cone guard band"
    ref3 = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=3)
    run_all_forms(mc, gctx, g, ref3, n_inst, 3)


def test_pixel_radii_on_powers_of_two(mc, O, H, gctx):
    """Unflipped projection (real mip levels): every meshlet's radius is rescaled so that its pixel radius lands
    within a few ulps of a power of two, where ceil(log2) jumps."""
    n_inst, per = 300, 128
    cone, sphere, t, off, cnt = scene(H, n_inst, per, 51)
    w, h = 1024, 512
    g = H.create_globals(eye=(40.0, 35.0, 30.0), fov_deg=50.0, width=w, height=h, flip_y=False)
    depth = H.synthetic_depth(w, h, n_discs=40, seed=3)
    pyr = O.hiz_build(depth)
    view = O.View(g, pyr, w, h)
    m = t["model"].reshape(n_inst, 4, 4).astype(np.float64)  # [inst][column][row]
    inst = np.repeat(np.arange(n_inst), per)
    c = sphere[:, :3].astype(np.float64)
    world = np.einsum("ncr,nc->nr", m[inst][:, :3, :3], c) + m[inst][:, 3, :3]
    vp = view.vp.astype(np.float64).reshape(4, 4)  # [column][row]
    cw = world @ vp[:3, 3] + vp[3, 3]
    scale = np.linalg.norm(m[inst][:, 0, :3], axis=1)
    proj11 = float(np.asarray(g["proj"]).reshape(-1)[5])
    px = proj11 * sphere[:, 3].astype(np.float64) * scale / cw * 2.0 * max(w, h)
    ok = (cw > 0.1) & (px > 0.01)
    k = np.clip(np.round(np.log2(np.where(ok, px, 1.0))), 0, 7)
    rs = np.random.RandomState(2)
    new_r = np.where(ok, sphere[:, 3] * (2.0 ** k) / np.where(ok, px, 1.0), sphere[:, 3]).astype("<f4")
    sphere[:, 3] = ulps(new_r, rs.randint(-5, 6, size=len(new_r)))
    ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=4)
    gctx.upload_bounds_soa(cone, sphere)
    gctx.upload_transforms(t)
    gctx.set_instance_ranges(off, cnt)
    gctx.hiz_build(depth, view=3)
    slow = run_all_forms(mc, gctx, g, ref, n_inst, 4, hiz_view=3)
    assert slow > int(ok.sum()) // 2, (slow, int(ok.sum()))
    ref7 = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=7)
    run_all_forms(mc, gctx, g, ref7, n_inst, 7, hiz_view=3)


def test_depths_on_the_threshold(mc, O, H, gctx):
    """Flipped projection (mip level 0): the depth texel each sampled pair reads is set to its own nz + 0.05, a few
    ulps either way, so that nz < d - 0.05 is decided in the last place."""
    n_inst, per = 64, 64
    cone, sphere, t, off, cnt = scene(H, n_inst, per, 61)
    w, h = 1024, 512
    g = H.create_globals(eye=(40.0, 35.0, 30.0), fov_deg=50.0, width=w, height=h, flip_y=True)
    depth = H.synthetic_depth(w, h, n_discs=10, seed=5).copy()
    view = O.View(g, O.hiz_build(depth), w, h)
    rs = np.random.RandomState(3)
    touched = 0
    for p in range(n_inst * per):
        i = p // per
        r = O.cull_pair(view, t["model"][i], cone[p], sphere[p, :3], float(sphere[p, 3]), tests=4)
        if r.mip != 0 or not np.isfinite(r.occlusion_margin):
            continue
        nz = np.float64(r.occlusion_margin) + (np.float64(depth[r.ty, r.tx]) - 0.05)
        if not (0.0 < nz < 1.0):
            continue
        depth[r.ty, r.tx] = ulps(np.array([nz + 0.05], dtype="<f4"), rs.randint(-3, 4, size=1))[0]
        touched += 1
    assert touched > 1000
    ref = O.cull_scene(O.View(g, O.hiz_build(depth), w, h), t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt),
                       tests=4, want_margins=True)
    assert int((np.abs(ref.margins[:, 1]) < 1e-6).sum()) > touched // 2
    gctx.upload_bounds_soa(cone, sphere)
    gctx.upload_transforms(t)
    gctx.set_instance_ranges(off, cnt)
    gctx.hiz_build(depth, view=0)
    slow = run_all_forms(mc, gctx, g, ref, n_inst, 4)
    # the depth guard is relative to nz itself (about 1e-9 here, nz being 0.002): only the pairs whose threshold
    # came out closer than that needed the division, the others were decided - correctly, see above - without
    assert slow > 50, (slow, touched)
