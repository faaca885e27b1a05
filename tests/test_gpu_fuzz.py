"""Seeded random scenes through the C ABI against the oracle: instance counts, ragged / empty /
overlapping meshlet ranges, cameras, projection flips, pyramid sizes (odd ones included), test
combinations, compaction modes, command layouts, and both kernel families (cache-resident and
streamed bounds).  Everything is compared bit for bit; the atomic compaction modes after sorting."""
import os

import numpy as np
import pytest

from test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu

# a longer soak than the default run: MCULL_FUZZ_SCALE=20 runs 20 times as many seeds
_SCALE = int(os.environ.get("MCULL_FUZZ_SCALE", "1"))


def _context_with(mc, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return mc.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def contexts(mc):
    """Cache-resident kernels, streamed kernels, and the streaming pyramid scheme with its strip / block maxima; in all
    three the instance level runs for scenes of any size and never backs off, so that the random scenes go through
    it (the library's own threshold is the size from which it pays)."""
    inst = dict(MCULL_PRECULL_MIN_GROUPS="1", MCULL_PRECULL_ADAPTIVE="0")
    ctxs = (_context_with(mc, **inst), _context_with(mc, MCULL_FORCE_STREAMING="1", **inst),
            _context_with(mc, MCULL_HIZ_STREAM="1", **inst), _context_with(mc, MCULL_FORCE_STREAMING="1", MCULL_HIZ_STREAM="1", **inst))
    yield ctxs
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_random_scene(mc, O, H, contexts, seed):
    rs = np.random.RandomState(1000 + seed)
    ctx = contexts[seed % len(contexts)]
    n_meshlets = int(rs.choice([1, 63, 64, 65, 200, 1000, 5000]))
    n_inst = int(rs.choice([1, 2, 7, 64, 257, 900]))
    cone, sphere = H.synthetic_bounds(n_meshlets, seed=seed)
    sphere = sphere.copy()
    sphere[:, :3] = (sphere[:, :3] - 0.5) * rs.uniform(0.5, 4.0)
    sphere[:, 3] *= rs.uniform(0.2, 5.0)
    t = H.synthetic_transforms(n_inst, extent=float(rs.uniform(2.0, 40.0)), seed=seed + 1)
    if rs.rand() < 0.5:  # non-uniform scale on some instances: the radius follows length(model[0].xyz) as in the reference
        m = t["model"].reshape(n_inst, 4, 4).copy()
        k = rs.randint(0, n_inst, max(1, n_inst // 3))
        m[k, :3, :3] *= rs.uniform(0.1, 3.0, (len(k), 3, 1)).astype("<f4")
        t["model"] = m.reshape(n_inst, 16)
    ranged = rs.rand() < 0.7
    if ranged:
        off = rs.randint(0, n_meshlets, n_inst).astype("<u4")
        cnt = np.minimum(rs.randint(0, min(n_meshlets, 300) + 1, n_inst), n_meshlets - off).astype("<u4")
        cnt[rs.rand(n_inst) < 0.15] = 0
        ranges = (off, cnt)
    else:
        ranges = None
    w, h = int(rs.choice([64, 100, 257, 640, 1023])), int(rs.choice([48, 37, 128, 360, 513]))
    depth = H.synthetic_depth(w, h, n_discs=int(rs.randint(1, 20)), seed=seed + 2)
    eye = tuple(rs.uniform(-30, 30, 3))
    g = H.create_globals(eye=eye, centre=tuple(rs.uniform(-5, 5, 3)), fov_deg=float(rs.uniform(20, 100)),
                         z_near=float(rs.choice([0.01, 0.1, 1.0])), width=w, height=h, flip_y=bool(rs.rand() < 0.5),
                         meshlet_count=n_meshlets)
    view_slot = int(rs.randint(0, mc.MAX_VIEWS))
    ctx.upload_bounds_soa(cone, sphere)
    ctx.upload_transforms(t)
    ctx.set_instance_ranges(*(ranges if ranged else (None, None)))
    ctx.hiz_build(depth, view=view_slot)
    view = O.View(g, O.hiz_build(depth), w, h)
    for _ in range(3):
        tests = int(rs.randint(0, 8))
        compaction = int(rs.choice([0, 0, 1, 2]))
        stride, offset = [(8, 0), (28, 20), (16, 4)][int(rs.randint(0, 3))]
        ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=ranges, tests=tests)
        ctx.cull(g, tests=tests, hiz_view=view_slot, compaction=compaction, write_mask=True, cmd_stride=stride, cmd_offset=offset)
        assert np.array_equal(ctx.download_mask(), ref.mask), (seed, tests, compaction)
        check_against_oracle(mc, ctx, ref, n_inst, compaction, stride=stride, offset=offset)
    ctx.set_instance_ranges(None, None)


def random_meshlets(mc, rs, n_meshlets, n_vertices, spread):
    """Meshlets of random triangles over random vertices (not a surface: what matters here is the set of
    triangle shapes, sizes and depths the rasteriser meets, degenerate ones included)."""
    verts = np.zeros(n_vertices, dtype=mc.VERTEX)
    verts["pos"][:, :3] = rs.uniform(-spread, spread, (n_vertices, 3)).astype("<f4")
    meshlets = np.zeros(n_meshlets, dtype=mc.MESHLET)
    data = []
    for m in range(n_meshlets):
        nv = int(rs.randint(3, 65))
        nt = int(rs.randint(1, 125))
        ids = rs.choice(n_vertices, nv, replace=nv > n_vertices).astype("<u4")
        local = rs.randint(0, nv, (nt, 3)).astype("u1")
        if nt > 4:
            local[0] = [0, 0, 1]                      # degenerate: two equal corners
            local[1] = [2, 2, 2]
        packed = np.zeros((3 * nt + 3) // 4 * 4, dtype="u1")
        packed[: 3 * nt] = local.reshape(-1)
        meshlets[m]["dataOffset"] = sum(len(x) for x in data)
        meshlets[m]["vertexCount"], meshlets[m]["triangleCount"] = nv, nt
        data += [ids, packed.view("<u4")]
    return verts, meshlets, np.concatenate(data).astype("<u4")


@pytest.mark.parametrize("seed", range(12 * _SCALE))
def test_random_triangles_through_the_rasteriser(mc, O, H, contexts, seed):
    """Random triangle soups at random scales and cameras - sub-pixel to screen-filling, behind the eye,
    across the near plane, degenerate - rasterised on the GPU and by the oracle: every pixel bit for bit."""
    rs = np.random.RandomState(5000 + seed)
    ctx = contexts[0]
    n_inst = int(rs.choice([1, 3, 20]))
    verts, meshlets, data = random_meshlets(mc, rs, int(rs.choice([1, 5, 40])), int(rs.choice([8, 64, 500])),
                                            float(rs.choice([0.05, 0.5, 3.0])))
    t = H.synthetic_transforms(n_inst, extent=float(rs.choice([0.5, 4.0])), seed=seed)
    w, h = int(rs.choice([33, 160, 800])), int(rs.choice([17, 120, 600]))
    eye = tuple(rs.uniform(-3, 3, 3))  # often inside the soup: triangles cross the near plane and surround the eye
    g = H.create_globals(eye=eye, centre=tuple(rs.uniform(-1, 1, 3)), fov_deg=float(rs.uniform(30, 110)),
                         z_near=float(rs.choice([0.01, 0.1])), width=w, height=h, flip_y=bool(rs.rand() < 0.5),
                         meshlet_count=len(meshlets))
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(meshlets)
    ctx.upload_geometry(verts, data)
    ctx.compute_bounds()
    # cone, apex and radius of these random (partly degenerate) meshlets against buildMeshletCon; where an invalid
    # operation made a NaN its sign and payload follow the machine, so NaNs compare alike
    got_b, want_b = ctx.download_meshlets(len(meshlets)), O.build_meshlet_cones(verts, data, meshlets)
    for f in ("cone", "coneApex", "radius"):
        a, b = np.asarray(got_b[f], dtype="<f4"), np.asarray(want_b[f], dtype="<f4")
        same = (a.view("<u4") == b.view("<u4")) | (np.isnan(a) & np.isnan(b))
        assert same.all(), (seed, f)
    ctx.upload_transforms(t)
    ctx.cull(g, tests=0)  # every pair: the rasteriser has to cope with what a cull would have removed
    lst = ctx.download_list()
    assert len(lst) == n_inst * len(meshlets)
    ctx.depth_rasterize(g, w, h)
    got = ctx.depth_download()
    with np.errstate(all="ignore"):
        want = O.rasterize_depth(verts, data, meshlets, t, lst, g, w, h)
    assert np.array_equal(got.view("<u4"), want.view("<u4")), "seed %d: %d pixels differ" % (seed, (got.view("<u4") != want.view("<u4")).sum())
    # accumulate a second view into the same image without clearing
    g2 = H.create_globals(eye=tuple(rs.uniform(-3, 3, 3)), width=w, height=h, meshlet_count=len(meshlets))
    ctx.cull(g2, tests=0)
    ctx.depth_rasterize(g2, w, h, clear=False)
    with np.errstate(all="ignore"):
        want2 = O.rasterize_depth(verts, data, meshlets, t, lst, g2, w, h, depth=want.copy())
    assert np.array_equal(ctx.depth_download().view("<u4"), want2.view("<u4"))


def test_mask_of_a_scene_without_groups(mc, O, H, contexts):
    """Found by the long soak (seed 39): every instance owns zero meshlets, a mask is asked for - it has zero words."""
    ctx = contexts[0]
    cone, sphere = H.synthetic_bounds(64, seed=1)
    t = H.synthetic_transforms(2, extent=3.0, seed=2)
    ctx.upload_bounds_soa(cone, sphere)
    ctx.upload_transforms(t)
    ctx.set_instance_ranges(np.array([3, 9], dtype="<u4"), np.zeros(2, dtype="<u4"))
    g = H.create_globals(meshlet_count=64)
    for compaction in (0, 1, 2):
        r = ctx.cull(g, tests=3, compaction=compaction, write_mask=True)
        assert r.n_groups == 0 and not r.mask
        assert len(ctx.download_mask()) == 0 and ctx.download_count() == 0 and len(ctx.download_list()) == 0
        assert not ctx.download_commands(2).any()
    ctx.set_instance_ranges(None, None)


@pytest.fixture(scope="module")
def hiz_contexts(mc):
    out = [mc.Context(0)]
    for switch in ("MCULL_HIZ_TILES", "MCULL_HIZ_SINGLE_LAUNCH", "MCULL_HIZ_STREAM"):
        os.environ[switch] = "1"
        try:
            out.append(mc.Context(0))
        finally:
            del os.environ[switch]
    yield out
    for c in out:
        c.close()


@pytest.mark.parametrize("seed", range(6 * _SCALE))
def test_random_pyramids(mc, O, hiz_contexts, seed):
    """Random image sizes (tiny, odd, one-texel-wide, wider than a tile row) with special values, through the
    size-chosen scheme and each forced one, two builds into the same slot: every level bit for bit."""
    rs = np.random.RandomState(9000 + seed)
    w = int(rs.choice([1, 2, 3, 5, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 511, 777, 1024, 1500, 2049]))
    h = int(rs.choice([1, 2, 4, 7, 31, 32, 33, 64, 65, 127, 128, 129, 300, 513, 1000]))
    slot = int(rs.randint(0, mc.MAX_VIEWS))
    for build in range(2):
        depth = rs.uniform(0, 1, (h, w)).astype("<f4")
        k = min(h * w, 12)
        depth.flat[rs.choice(h * w, k, replace=False)] = rs.choice(
            np.array([np.nan, -0.0, 0.0, np.inf, -np.inf, 1e-40, -1e-40, 1.0, 0.5, 2.0, -1.0, 3e38], dtype="<f4"), k)
        ref = O.hiz_levels(O.hiz_build(depth), w, h)
        for c in hiz_contexts:
            c.hiz_build(depth, view=slot)
            got = c.hiz_download(view=slot)
            assert len(got) == len(ref)
            for lvl, (a, b) in enumerate(zip(got, ref)):
                assert a.view("<u4").tobytes() == b.view("<u4").tobytes(), "seed %d %dx%d level %d" % (seed, w, h, lvl)


@pytest.mark.parametrize("seed", range(4 * _SCALE))
def test_random_groups(mc, O, H, seed):
    """Random shard counts and cut points (empty shards included), ranged or uniform scenes, NV and EXT records at
    random strides, several frames per group: every device's scene-wide buffers against the oracle's unsharded cull."""
    rs = np.random.RandomState(7000 + seed)
    shards = int(rs.choice([1, 2, 3, 5, 8]))
    n_meshlets = int(rs.choice([1, 64, 65, 300, 2000]))
    n_inst = int(rs.choice([1, 5, 64, 300]))
    cone, sphere = H.synthetic_bounds(n_meshlets, seed=seed)
    sphere = sphere.copy()
    sphere[:, :3] = (sphere[:, :3] - 0.5) * rs.uniform(0.5, 4.0)
    t = H.synthetic_transforms(n_inst, extent=float(rs.uniform(2.0, 30.0)), seed=seed + 1)
    ranged = rs.rand() < 0.6
    if ranged:
        off = rs.randint(0, n_meshlets, n_inst).astype("<u4")
        cnt = np.minimum(rs.randint(0, min(n_meshlets, 200) + 1, n_inst), n_meshlets - off).astype("<u4")
        cnt[rs.rand(n_inst) < 0.2] = 0
    cuts = np.sort(rs.randint(0, n_inst + 1, shards - 1)).tolist()
    cuts = [0] + cuts + [n_inst]
    w, h = int(rs.choice([64, 257, 640])), int(rs.choice([48, 128, 360]))
    depth = H.synthetic_depth(w, h, n_discs=int(rs.randint(1, 20)), seed=seed + 2)
    pyr = O.hiz_build(depth)
    with mc.Group([0] * shards) as grp:
        for r, s in enumerate(grp.shards):
            s.upload_bounds_soa(cone, sphere)
            s.upload_transforms(t[cuts[r]:cuts[r + 1]])
            if ranged:
                s.set_instance_ranges(off[cuts[r]:cuts[r + 1]], cnt[cuts[r]:cuts[r + 1]])
            s.hiz_build(depth, view=2)
        for _ in range(3):
            g = H.create_globals(eye=tuple(rs.uniform(-25, 25, 3)), centre=tuple(rs.uniform(-5, 5, 3)), fov_deg=float(rs.uniform(20, 100)),
                                 width=w, height=h, flip_y=bool(rs.rand() < 0.5), meshlet_count=n_meshlets)
            tests = int(rs.randint(0, 8))
            ext = bool(rs.rand() < 0.4)
            stride, offset = [(0, 0), (28, 16 if ext else 20), (16, 4)][int(rs.randint(0, 3))]
            ref = O.cull_scene(O.View(g, pyr, w, h), t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt) if ranged else None, tests=tests)
            grp.cull(g, tests=tests, hiz_view=2, cmd_form=mc.CMD_EXT_GROUPS if ext else mc.CMD_NV_MESHLETS, cmd_stride=stride, cmd_offset=offset)
            rec = 12 if ext else 8
            st = stride or rec
            own = np.where(cnt > 0)[0] if ranged else np.arange(n_inst if n_meshlets else 0)
            lists = []
            for r, s in enumerate(grp.shards):
                raw, ft, totals = grp.download(r)
                assert int(totals[-1]) == ref.total, (seed, r)
                words = raw.reshape(n_inst, st)[:, offset:offset + rec].copy().view("<u4").reshape(n_inst, rec // 4)
                assert np.array_equal(words[:, 0], ref.commands["taskCount"])
                if ext:
                    assert (words[:, 1:] == 1).all()
                    assert np.array_equal(ft[own], ref.commands["firstTask"][own])
                else:
                    assert np.array_equal(words[own, 1], ref.commands["firstTask"][own])
                lst = s.download_list().copy()
                lst["drawID"] += cuts[r]
                lists.append(lst)
            assert np.concatenate(lists).tobytes() == ref.list.tobytes()


@pytest.mark.parametrize("seed", range(4 * _SCALE))
def test_random_views_and_payloads(mc, O, H, contexts, seed):
    """1-8 views with their own cameras, tests and pyramids in one pass, the Globals handed over from the host and
    from a device array; every view equals its own single cull.  Then the literal per-32 task payloads of one view."""
    import torch

    rs = np.random.RandomState(11000 + seed)
    ctx = contexts[seed % len(contexts)]
    n_views = int(rs.randint(1, mc.MAX_VIEWS + 1))
    n_meshlets = int(rs.choice([1, 33, 64, 97, 500]))
    n_inst = int(rs.choice([1, 3, 70, 400]))
    cone, sphere = H.synthetic_bounds(n_meshlets, seed=seed + 5)
    sphere = sphere.copy()
    sphere[:, :3] = (sphere[:, :3] - 0.5) * rs.uniform(0.5, 4.0)
    t = H.synthetic_transforms(n_inst, extent=float(rs.uniform(2.0, 25.0)), seed=seed + 6)
    ml = np.zeros(n_meshlets, dtype=mc.MESHLET)  # AoS upload: payloads and multi-view use the same records
    ml["cone"], ml["coneApex"][:, :3], ml["radius"] = cone, sphere[:, :3], sphere[:, 3]
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(ml)
    ctx.upload_transforms(t)
    views, params, refs = [], [], []
    for v in range(n_views):
        w, h = int(rs.choice([64, 200, 640])), int(rs.choice([48, 150, 360]))
        depth = H.synthetic_depth(w, h, n_discs=int(rs.randint(1, 16)), seed=seed * 10 + v)
        ctx.hiz_build(depth, view=v)
        g = H.create_globals(eye=tuple(rs.uniform(-20, 20, 3)), centre=tuple(rs.uniform(-4, 4, 3)), fov_deg=float(rs.uniform(25, 95)),
                             width=w, height=h, flip_y=bool(rs.rand() < 0.5), meshlet_count=n_meshlets)
        tests = int(rs.randint(0, 8))
        views.append(g)
        params.append(dict(tests=tests, hiz_view=v, write_mask=True))
        refs.append(O.cull_scene(O.View(g, O.hiz_build(depth), w, h), t, ml, tests=tests))
    packed = np.concatenate([np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1) for g in views])
    g_dev = torch.from_numpy(packed.view("u1").reshape(-1).copy()).to("cuda:0")
    torch.cuda.synchronize()
    for device_globals in (False, True):
        if device_globals:
            ctx.cull_views(n_views, params, device_globals_ptr=g_dev.data_ptr(), meshlet_count=n_meshlets)
        else:
            ctx.cull_views(views, params)
        for v in range(n_views):
            ctx.select_result(v)
            assert np.array_equal(ctx.download_mask(), refs[v].mask), (seed, v, device_globals)
            check_against_oracle(mc, ctx, refs[v], n_inst, 0)
    v = int(rs.randint(0, n_views))
    ctx.select_result(v)
    want = O.task_payloads(O.View(views[v]), t, ml, tests=params[v]["tests"] & 3)  # frustum / cone only: no pyramid involved
    ctx.cull(views[v], tests=params[v]["tests"] & 3)
    assert ctx.task_payloads().tobytes() == want.tobytes()



def _compare(mc, c, ref, n_inst, tag):
    assert c.download_count() == ref.total, tag
    assert np.array_equal(c.download_mask(), ref.mask), tag
    assert c.download_list().tobytes() == ref.list.tobytes(), tag
    cmds = c.download_commands(n_inst).view(mc.CMD).reshape(-1)
    assert np.array_equal(cmds["taskCount"], ref.commands["taskCount"]) and np.array_equal(cmds["firstTask"], ref.commands["firstTask"]), tag


@pytest.mark.parametrize("seed", range(6 * _SCALE))
def test_random_large_instances(mc, O, H, seed):
    """Instances of 32 and more groups - the scenes that get a group level, the meshes' boxes and the pyramid's window
    tables: meshes that are flat, elongated, curled or scattered (where box and sphere differ most), image sizes that are
    and are not multiples of the texel reach, depth images from empty to walled, several cameras and test subsets per
    pyramid, pyramids rebuilt in between.  Everything against the oracle, which tests every pair."""
    rs = np.random.RandomState(9100 + seed)
    per = int(rs.choice([2048, 2100, 2560, 3000, 4500]))
    n_inst = int(rs.choice([8, 40, 120]))
    cone, sphere = H.synthetic_bounds(per, seed=seed + 5)
    sphere = sphere.copy()
    shape = int(rs.randint(0, 4))
    s = np.arange(per, dtype=np.float64) / per
    if shape == 0:    # a flat slab
        sphere[:, :3] = (sphere[:, :3] - 0.5) * np.array([6.0, 5.0, 0.2])
    elif shape == 1:  # a long bar, consecutive meshlets neighbours
        sphere[:, 0] = ((s - 0.5) * rs.uniform(5, 40) + rs.normal(0, 0.05, per)).astype("<f4")
        sphere[:, 1:3] = rs.normal(0, 0.2, (per, 2)).astype("<f4")
    elif shape == 2:  # a spiral
        turns = rs.uniform(1, 6)
        sphere[:, 0] = (4.0 * np.cos(2 * np.pi * turns * s)).astype("<f4")
        sphere[:, 1] = (4.0 * np.sin(2 * np.pi * turns * s)).astype("<f4")
        sphere[:, 2] = ((s - 0.5) * rs.uniform(2, 30)).astype("<f4")
    else:             # unrelated meshlets in a cube
        sphere[:, :3] = (sphere[:, :3] - 0.5) * rs.uniform(1.0, 8.0)
    sphere[:, 3] = rs.uniform(0.02, rs.uniform(0.05, 0.6), per).astype("<f4")
    if rs.rand() < 0.3:   # a few hostile bounds inside some groups
        k = rs.randint(0, per, 6)
        sphere[k[0], 3] = -0.1
        sphere[k[1], 0] = np.inf
        sphere[k[2], 3] = np.nan
        cone[k[3], :3] = 0.0
    t = H.synthetic_transforms(n_inst, extent=float(rs.uniform(5.0, 40.0)), seed=seed + 1)
    m = t["model"].reshape(n_inst, 4, 4).copy()
    m[:, :3, :3] *= rs.uniform(0.2, 1.5, n_inst).astype("<f4")[:, None, None]
    if rs.rand() < 0.4:  # and anisotropic scale on a few
        k = rs.randint(0, n_inst, max(1, n_inst // 4))
        m[k, :3, :3] *= rs.uniform(0.3, 2.0, (len(k), 3, 1)).astype("<f4")
    t["model"] = m.reshape(n_inst, 16)
    ranged = rs.rand() < 0.4
    off = np.zeros(n_inst, dtype="<u4")
    cnt = np.full(n_inst, per, dtype="<u4")
    if ranged:
        off = rs.randint(0, per // 8, n_inst).astype("<u4")
        cnt = (per - off - rs.randint(0, per // 8, n_inst)).astype("<u4")
    c = _context_with(mc, MCULL_PRECULL_MIN_GROUPS="1", MCULL_PRECULL_ADAPTIVE="0")
    try:
        c.upload_bounds_soa(cone, sphere)
        c.upload_transforms(t)
        c.set_instance_ranges(off, cnt) if ranged else c.set_instance_ranges(None, None)
        with np.errstate(all="ignore"):
            for build in range(2):
                w = int(rs.choice([640, 1001, 1280, 1920, 333]))
                h = int(rs.choice([360, 563, 720, 1080, 250]))
                kind = int(rs.randint(0, 4))
                depth = H.synthetic_depth(w, h, n_discs=int(rs.randint(1, 24)), seed=seed * 3 + build)
                if kind == 1:
                    depth[depth > 0] += np.float32(rs.uniform(0.0, 0.3))
                    depth[:, : w // int(rs.randint(2, 5))] = np.float32(rs.uniform(0.1, 0.9))
                elif kind == 2:
                    depth[:] = np.float32(rs.uniform(0.2, 0.9))
                    depth[rs.randint(0, h, 40), rs.randint(0, w, 40)] = 0.0  # pinholes in a wall
                elif kind == 3 and rs.rand() < 0.5:
                    depth[rs.randint(0, h), rs.randint(0, w)] = np.nan
                slot = int(rs.randint(0, mc.MAX_VIEWS))
                c.hiz_build(depth, view=slot)
                pyr = O.hiz_build(depth)
                for cam in range(3):
                    g = H.create_globals(eye=tuple(rs.uniform(-45, 45, 3)), centre=tuple(rs.uniform(-8, 8, 3)),
                                         fov_deg=float(rs.uniform(25, 95)), z_near=float(rs.choice([0.05, 0.1, 1.0])), width=w, height=h,
                                         flip_y=bool(rs.rand() < 0.5), meshlet_count=per)
                    tests = int(rs.choice([7, 7, 5, 4, 6, 3]))
                    ref = O.cull_scene(O.View(g, pyr, w, h), t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=tests)
                    c.cull(g, tests=tests, hiz_view=slot, write_mask=True)
                    _compare(mc, c, ref, n_inst, (seed, build, cam, tests, w, h))
                    dropped, tested = c.precull_group_stats()  # (tested is 0 when the instance level has left nothing)
                    assert dropped <= int((ref.mask == 0).sum()), "a dropped group has no visible meshlet"
    finally:
        c.close()


@pytest.mark.parametrize("seed", range(4 * _SCALE))
def test_random_pyramids_of_several_views(mc, O, seed):
    """mcull_hiz_build_views on random sizes, with the streaming scheme forced so that sizes from 1 024 texels a side take
    the batched launches (narrower ones fall back to single builds): widths that are and are not multiples of four or of
    the tile size, 1-6 views into random distinct slots, NaNs and infinities, every level of every slot against the oracle."""
    import torch

    rs = np.random.RandomState(9900 + seed)
    w = int(rs.choice([1024, 1028, 1031, 1500, 2049, 2560, 700]))
    h = int(rs.choice([32, 67, 256, 600, 1025]))
    n = int(rs.randint(1, 7))
    slots = [int(x) for x in rs.permutation(mc.MAX_VIEWS)[:n]]
    images = []
    for k in range(n):
        d = rs.uniform(0, 1, (h, w)).astype("<f4")
        if rs.rand() < 0.5:
            d.flat[rs.randint(0, h * w, 5)] = [np.nan, np.inf, -np.inf, -0.0, 1e-42]
        images.append(d)
    c = _context_with(mc, MCULL_HIZ_STREAM="1")
    try:
        dev = [torch.from_numpy(d).to("cuda:0") for d in images]
        torch.cuda.synchronize()
        for repeat in range(2):
            c.hiz_build_views([t.data_ptr() for t in dev], views=slots, width=w, height=h, mem=mc.MEM_DEVICE)
            for slot, d in zip(slots, images):
                ref = O.hiz_levels(O.hiz_build(d), w, h)
                got = c.hiz_download(view=slot)
                assert len(got) == len(ref)
                for lvl, (a, b) in enumerate(zip(got, ref)):
                    assert a.view("<u4").tobytes() == b.view("<u4").tobytes(), "seed %d %dx%d slot %d level %d" % (seed, w, h, slot, lvl)
    finally:
        c.close()


@pytest.mark.parametrize("seed", range(4 * _SCALE))
def test_random_call_sequences(mc, O, H, seed):
    """The adaptive parts of a context - list-length estimates and back-off per set of tests, the probe taken in turns,
    window tables made when a pyramid is culled against twice, hints kept across new matrices - see a random sequence of
    what a renderer does between culls: other tests, another camera, a rebuilt pyramid, moved instances, a changed
    instance count.  Every cull against the oracle (the back-off left on: these contexts are the library's defaults but
    for the scene size from which the instance level runs)."""
    rs = np.random.RandomState(7700 + seed)
    env = dict(MCULL_PRECULL_MIN_GROUPS="1")
    if rs.rand() < 0.5:
        env["MCULL_FORCE_STREAMING"] = "1"
    c = _context_with(mc, **env)
    try:
        per = int(rs.choice([128, 256, 2304]))                       # 2304: 36 groups per instance, the group level's scenes
        n_all = int(rs.choice([300, 1500])) if per < 2304 else int(rs.choice([40, 150]))
        cone, sphere = H.synthetic_bounds(n_all * per, seed=seed + 3)
        base = H.synthetic_transforms(n_all, extent=float(rs.uniform(10.0, 50.0)), seed=seed + 4)
        w, h = [(320, 200), (640, 360), (501, 333)][rs.randint(3)]
        cams = [H.create_globals(eye=tuple(rs.uniform(20.0, 90.0, 3)), width=w, height=h, flip_y=bool(rs.rand() < 0.5)),
                H.create_globals(eye=tuple(rs.uniform(1.0, 8.0, 3)), width=w, height=h),
                H.create_globals(eye=(60.0, 55.0, 40.0), centre=(120.0, 110.0, 80.0), width=w, height=h)]  # looking away
        depths = [H.synthetic_depth(w, h, n_discs=int(rs.randint(1, 40)), seed=seed + k) for k in range(2)]
        depths.append(np.full((h, w), 0.9, dtype="<f4"))             # a wall
        pyrs = [O.hiz_build(d) for d in depths]
        c.upload_bounds_soa(cone, sphere)
        n_inst, t, cur_depth = n_all, base, 0
        off = (np.arange(n_all, dtype=np.uint64) * per).astype("<u4")
        cnt = np.full(n_all, per, dtype="<u4")

        def scene():
            c.upload_transforms(t)
            c.set_instance_ranges(off[:n_inst], cnt[:n_inst])

        scene()
        c.hiz_build(depths[0], view=0)
        refs = {}
        version = 0
        for step in range(30):
            op = rs.rand()
            if op < 0.12:                                            # the instances move: same count, new matrices
                t = t.copy()
                m = t["model"].reshape(n_inst, 4, 4).copy()
                m[:, 3, :3] += rs.normal(0.0, 2.0, (n_inst, 3)).astype("<f4")
                t["model"] = m.reshape(n_inst, 16)
                c.upload_transforms(t)
                version += 1
            elif op < 0.17:                                          # another instance count
                n_inst = int(rs.choice([n_all, n_all // 2, max(1, n_all // 7)]))
                t = base[:n_inst].copy()
                scene()
                version += 1
            elif op < 0.30:                                          # the pyramid is rebuilt
                cur_depth = int(rs.randint(len(depths)))
                c.hiz_build(depths[cur_depth], view=0)
            if rs.rand() < 0.15:                                     # several views in one pass, all against pyramid 0
                nv = int(rs.randint(2, 5))
                vc = [int(rs.randint(len(cams))) for _ in range(nv)]
                vt = [int(rs.choice([7, 3, 1, 5])) for _ in range(nv)]
                c.cull_views([cams[k] for k in vc], [dict(tests=k, hiz_view=0, write_mask=True) for k in vt])
                for v in range(nv):
                    key = (version, vc[v], vt[v], cur_depth if vt[v] & 4 else -1)
                    if key not in refs:
                        refs[key] = O.cull_scene(O.View(cams[vc[v]], pyrs[cur_depth], w, h), t, cone_axis_w=cone, centre_radius=sphere,
                                                 ranges=(off[:n_inst], cnt[:n_inst]), tests=vt[v])
                    c.select_result(v)
                    _compare(mc, c, refs[key], n_inst, (step, "view", v, key))
                c.select_result(0)
            tests = int(rs.choice([7, 7, 3, 3, 1, 5, 6, 2]))
            cam = int(rs.randint(len(cams)))
            key = (version, cam, tests, cur_depth if tests & 4 else -1)
            if key not in refs:
                refs[key] = O.cull_scene(O.View(cams[cam], pyrs[cur_depth], w, h), t, cone_axis_w=cone, centre_radius=sphere,
                                         ranges=(off[:n_inst], cnt[:n_inst]), tests=tests)
            ref = refs[key]
            c.cull(cams[cam], tests=tests, hiz_view=0, write_mask=True)
            if rs.rand() < 0.6:                                      # not every frame is looked at: the probe's answer may be late
                assert c.download_count() == ref.total, (step, key)
                assert np.array_equal(c.download_mask(), ref.mask), (step, key)
                assert c.download_list().tobytes() == ref.list.tobytes(), (step, key)
                cmds = c.download_commands(n_inst).view(mc.CMD).reshape(-1)
                assert np.array_equal(cmds["taskCount"], ref.commands["taskCount"]) and np.array_equal(cmds["firstTask"], ref.commands["firstTask"])
    finally:
        c.close()
