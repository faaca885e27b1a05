"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bars: visibility masks, compacted lists (after sorting where the mode leaves order open), indirect
commands and Hi-Z pyramids bit-exact; bounds bit-exact (the contract allows 1e-5 relative).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def sorted_list(lst):
    a = np.asarray(lst).view("<u4").reshape(-1, 2).astype(np.uint64)
    return np.sort((a[:, 0] << np.uint64(32)) | a[:, 1])


def check_against_oracle(mc, ctx, r_ref, n_inst, compaction, stride=8, offset=0):
    total = ctx.download_count()
    assert total == r_ref.total
    lst = ctx.download_list()
    if compaction == mc.COMPACT_ORDERED:
        assert lst.tobytes() == r_ref.list.tobytes(), "ordered list must equal the oracle's sorted list"
    else:
        assert np.array_equal(sorted_list(lst), sorted_list(r_ref.list))
    raw = ctx.download_commands(n_inst, stride).reshape(n_inst, stride)
    cmds = raw[:, offset:offset + 8].copy().view(mc.CMD).reshape(-1)
    assert np.array_equal(cmds["taskCount"], r_ref.commands["taskCount"])
    if compaction == mc.COMPACT_ORDERED:
        assert np.array_equal(cmds["firstTask"], r_ref.commands["firstTask"])
    return cmds


@pytest.mark.parametrize("tests", [7, 3, 1, 2, 4, 0])
@pytest.mark.parametrize("compaction", [0, 1, 2])
def test_config1_kitten(mc, O, ctx, kitten, config1, H, tests, compaction):
    """Config 1: kitten x 1000 instances at 800x600 with every test combination and compaction mode."""
    g, t = config1["globals"], config1["transforms"]
    depth = H.synthetic_depth(800, 600, n_discs=24, seed=5)
    pyr = O.hiz_build(depth)
    view = O.View(g, pyr, 800, 600)
    ref = O.cull_scene(view, t, kitten["meshlets"], tests=tests)
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(kitten["meshlets"])
    ctx.upload_transforms(t)
    ctx.hiz_build(depth, view=0)
    ctx.cull(g, tests=tests, compaction=compaction, write_mask=True)
    mask = ctx.download_mask()
    assert np.array_equal(mask, ref.mask), "visibility mask must be bit-exact"
    check_against_oracle(mc, ctx, ref, len(t), compaction)


def test_config1_golden_hashes(mc, O, ctx, kitten, config1, manifest):
    """The committed golden hashes of config 1 (cleared depth) reproduced by the GPU."""
    g, t = config1["globals"], config1["transforms"]
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(kitten["meshlets"])
    ctx.upload_transforms(t)
    ctx.hiz_build(np.zeros((600, 800), dtype="<f4"), view=0)
    for label, tests in (("all", 7), ("frustum_cone", 3), ("frustum", 1), ("cone", 2)):
        want = manifest["config1"]["results"][label]
        ctx.cull(g, tests=tests)
        assert ctx.download_count() == want["total"]
        assert "%016x" % O.fnv1a64(ctx.download_list()) == want["fnv_list"]
        assert "%016x" % O.fnv1a64(ctx.download_mask()) == want["fnv_mask"]
        assert "%016x" % O.fnv1a64(ctx.download_commands(len(t))) == want["fnv_commands"]


def test_draw_indirect_layout(mc, O, ctx, kitten, config1):
    """Commands written at the reference's DrawIndirect stride/offset (28/20, Engine.cpp:2439)."""
    g, t = config1["globals"], config1["transforms"]
    view = O.View(g)
    ref = O.cull_scene(view, t, kitten["meshlets"], tests=3)
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(kitten["meshlets"])
    ctx.upload_transforms(t)
    ctx.cull(g, tests=3, cmd_stride=28, cmd_offset=20)
    raw = ctx.download_commands(len(t), 28).reshape(len(t), 28)
    assert not raw[:, :20].any(), "bytes outside the NV command are left untouched"
    check_against_oracle(mc, ctx, ref, len(t), 0, 28, 20)


def test_task_payloads(mc, O, ctx, kitten, config1):
    """The literal per-32 task payloads of shader.task:28-35,:103-118."""
    g, t = config1["globals"], config1["transforms"][:100]
    view = O.View(g)
    ref = O.task_payloads(view, t, kitten["meshlets"], tests=3)
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(kitten["meshlets"])
    ctx.upload_transforms(t)
    ctx.cull(g, tests=3)
    got = ctx.task_payloads()
    assert got.tobytes() == ref.tobytes()


def test_mixed_meshes_and_ragged_ranges(mc, O, ctx, kitten, bunny, H):
    """Config 2 shape: instances drawing different meshes through per-instance ranges, including an
    empty instance, a 1-meshlet instance and ranges that are not multiples of 64."""
    meshlets = np.concatenate([kitten["meshlets"], bunny["meshlets"]])
    nk, nb = len(kitten["meshlets"]), len(bunny["meshlets"])
    n = 600
    t = H.update_transforms(n)
    off = np.zeros(n, dtype="<u4"); cnt = np.zeros(n, dtype="<u4")
    for i in range(n):
        k = i % 5
        if k == 0: off[i], cnt[i] = 0, nk
        elif k == 1: off[i], cnt[i] = nk, nb
        elif k == 2: off[i], cnt[i] = 17, 64
        elif k == 3: off[i], cnt[i] = nk + 5, 1
        else: off[i], cnt[i] = 0, 0
    g = H.create_globals(meshlet_count=0)
    view = O.View(g)
    ctx.upload_meshlets(meshlets)
    ctx.upload_transforms(t)
    ctx.set_instance_ranges(off, cnt)
    for compaction in (0, 1, 2):
        for tests in (3, 1):
            ref = O.cull_scene(view, t, meshlets, ranges=(off, cnt), tests=tests)
            ctx.cull(g, tests=tests, compaction=compaction, write_mask=True)
            assert np.array_equal(ctx.download_mask(), ref.mask)
            check_against_oracle(mc, ctx, ref, n, compaction)
    ctx.set_instance_ranges(None, None)


def test_synthetic_scene_soa(mc, O, ctx, H):
    """Config 4 shape at a size the oracle finishes in seconds: unique meshlets per instance (SoA
    upload), Hi-Z on, unflipped projection so that mip levels above 0 are exercised."""
    n_inst, per = 4096, 128
    cone, sphere = H.synthetic_bounds(n_inst * per, seed=1)
    t = H.synthetic_transforms(n_inst, extent=50.0, seed=2)
    off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
    cnt = np.full(n_inst, per, dtype="<u4")
    w, h = 1024, 512
    depth = H.synthetic_depth(w, h, n_discs=40, seed=3)
    for flip in (True, False):
        g = H.create_globals(eye=(150.0, 140.0, 130.0), fov_deg=50.0, width=w, height=h, flip_y=flip)
        pyr = O.hiz_build(depth)
        view = O.View(g, pyr, w, h)
        ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=7, threads=8)
        ctx.upload_bounds_soa(cone, sphere)
        ctx.upload_transforms(t)
        ctx.set_instance_ranges(off, cnt)
        ctx.hiz_build(depth, view=1)
        for compaction in (0, 1, 2):
            ctx.cull(g, tests=7, hiz_view=1, compaction=compaction, write_mask=True)
            assert np.array_equal(ctx.download_mask(), ref.mask)
            check_against_oracle(mc, ctx, ref, n_inst, compaction)
        assert 0 < ref.total < ref.n_pairs
    ctx.set_instance_ranges(None, None)


@pytest.mark.parametrize("shape", [(600, 800), (3, 5), (1, 1), (64, 64), (65, 127), (768, 1024), (1080, 1920),
                                   (37, 4000), (2160, 3840), (1, 300), (2, 2), (4, 4), (129, 130)])
def test_hiz_pyramid_bit_exact(mc, O, ctx, shape):
    """Every texel of every level, bit for bit, including odd sizes, collapse cases and NaN/-0."""
    h, w = shape
    rs = np.random.RandomState(h * 7919 + w)
    depth = rs.uniform(0, 1, (h, w)).astype("<f4")
    if h * w > 16:
        depth.flat[rs.randint(0, h * w, 8)] = [np.nan, -0.0, 0.0, np.inf, -np.inf, 1e-40, np.nan, 0.5]
    ref = O.hiz_levels(O.hiz_build(depth), w, h)
    ctx.hiz_build(depth, view=2)
    got = ctx.hiz_download(view=2)
    assert len(got) == len(ref) == mc.hiz_layout(w, h)[0]
    for k, (a, b) in enumerate(zip(got, ref)):
        assert a.shape == b.shape
        assert a.view("<u4").tobytes() == b.view("<u4").tobytes(), "level %d differs" % k


@pytest.mark.parametrize("shape,mem", [((2160, 3840), "device"), ((2306, 4100), "device"), ((2160, 3840), "host"), ((600, 800), "device"), ((5, 3), "host")])
def test_pyramids_of_several_views_in_one_call(mc, O, H, shape, mem):
    """mcull_hiz_build_views = one mcull_hiz_build per view: every texel of every level of every slot (one image holds
    NaNs, infinities and -0), twice in a row, and a slot that is not named stays as it was.  Large images in device
    memory take the batched launches; host images, small ones and odd widths the others."""
    import torch

    h, w = shape
    rs = np.random.RandomState(h + w)
    images = []
    for k in range(3):
        d = rs.uniform(0, 1, (h, w)).astype("<f4")
        if k == 1 and h * w > 16:
            d.flat[rs.randint(0, h * w, 6)] = [np.nan, -0.0, np.inf, -np.inf, 1e-40, np.nan]
        images.append(d)
    refs = [O.hiz_levels(O.hiz_build(d), w, h) for d in images]
    slots = [5, 0, 3]
    with mc.Context(0) as c:
        other = rs.uniform(0, 1, (40, 60)).astype("<f4")
        c.hiz_build(other, view=1)
        for repeat in range(2):  # the second build finds every buffer in place (and the ticket words as the first left them)
            if mem == "device":
                dev = [torch.from_numpy(d).to("cuda:0") for d in images]
                torch.cuda.synchronize()
                c.hiz_build_views([t.data_ptr() for t in dev], views=slots, width=w, height=h, mem=mc.MEM_DEVICE)
            else:
                c.hiz_build_views(images, views=slots, mem=mc.MEM_HOST)
            for slot, ref in zip(slots, refs):
                got = c.hiz_download(view=slot)
                assert len(got) == len(ref)
                for k, (a, b) in enumerate(zip(got, ref)):
                    assert a.view("<u4").tobytes() == b.view("<u4").tobytes(), "slot %d level %d differs" % (slot, k)
        untouched = c.hiz_download(view=1)
        for a, b in zip(untouched, O.hiz_levels(O.hiz_build(other), 60, 40)):
            assert a.view("<u4").tobytes() == b.view("<u4").tobytes()
        with pytest.raises(mc.McullError):
            c.hiz_build_views(images[:2], views=[2, 2], mem=mc.MEM_HOST)  # one slot named twice
        with pytest.raises(mc.McullError):
            c.hiz_build_views(images[:1], views=[8], mem=mc.MEM_HOST)


@pytest.mark.parametrize("shape", [(1080, 1920), (2160, 3840)])
def test_hiz_rebuild_same_buffer_sees_no_stale_texels(mc, O, ctx, shape):
    """Rebuilding into the same pyramid with different depth must not pick up the previous build's
    level-6/level-9 texels from any cache: those are the texels one workgroup hands to another."""
    h, w = shape
    rs = np.random.RandomState(5)
    a = rs.uniform(0, 1, (h, w)).astype("<f4")
    depths = [a, (1.0 - a).astype("<f4"), np.roll(a, 97, axis=1).copy(), a]
    for depth in depths:
        ref = O.hiz_levels(O.hiz_build(depth), w, h)
        ctx.hiz_build(depth, view=2)
        for _ in range(2):  # the second download reads what the first left in the caches
            got = ctx.hiz_download(view=2)
            for k, (x, y) in enumerate(zip(got, ref)):
                assert x.view("<u4").tobytes() == y.view("<u4").tobytes(), "level %d differs" % k


@pytest.mark.parametrize("switch", ["MCULL_HIZ_SINGLE_LAUNCH", "MCULL_HIZ_STREAM", "MCULL_HIZ_TILES"])
def test_hiz_forced_schemes(mc, O, monkeypatch, switch):
    """The pyramid schemes the library chooses between by image size, each forced for every size:
    MCULL_HIZ_STREAM (streaming pass for levels 0-3, then tiles on level 3 with the top levels handed
    to the last-arriving waves by tickets), MCULL_HIZ_TILES (tiles on the depth image + a one-workgroup
    launch), MCULL_HIZ_SINGLE_LAUNCH (tiles on the depth image, tickets).  Same pyramids bit for bit,
    NaN / signed zero / infinities included, also when rebuilt into the same buffer."""
    monkeypatch.setenv(switch, "1")
    c = mc.Context(0)
    monkeypatch.delenv(switch)
    rs = np.random.RandomState(17)
    for h, w in ((600, 800), (65, 127), (1080, 1920), (2160, 3840), (37, 4000), (1, 300), (4096, 4096), (130, 129),
                 (32, 32), (33, 47), (31, 64), (768, 1366), (40, 8192), (1023, 1025), (64, 4099)):
        a = rs.uniform(0, 1, (h, w)).astype("<f4")
        a.flat[rs.randint(0, h * w, 8)] = [np.nan, -0.0, 0.0, np.inf, -np.inf, 1e-40, np.nan, 0.5]
        for depth in (a, (1.0 - a).astype("<f4")):
            ref = O.hiz_levels(O.hiz_build(depth, threads=8), w, h)
            c.hiz_build(depth, view=1)
            for k, (x, y) in enumerate(zip(c.hiz_download(view=1), ref)):
                assert x.view("<u4").tobytes() == y.view("<u4").tobytes(), "%dx%d level %d" % (w, h, k)
    c.close()


def test_hiz_large_closed_form(mc, ctx):
    """8192^2 (config 5's size): level k equals the block minimum of level 0 (size-independent property)."""
    import torch

    w = h = 8192
    gen = torch.Generator(device="cuda").manual_seed(11)
    d = torch.rand((h, w), device="cuda", dtype=torch.float32, generator=gen)
    ctx.hiz_build(d.data_ptr(), view=3, width=w, height=h, mem=mc.MEM_DEVICE)
    ctx.synchronize()
    ptr, n = ctx.hiz_pyramid_ptr(3)
    levels, lw, lh, off, total = mc.hiz_layout(w, h)
    assert n == total and levels == 14
    pyr = torch.empty(total, device="cuda", dtype=torch.float32)
    import ctypes
    from conftest import loaded_hip_runtime
    hip = loaded_hip_runtime()
    assert hip.hipMemcpy(ctypes.c_void_p(pyr.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(total * 4), 3) == 0
    cur = d
    for k in range(levels):
        lvl = pyr[off[k]: off[k] + lw[k] * lh[k]].view(lh[k], lw[k])
        assert torch.equal(lvl, cur), "level %d" % k
        if k + 1 < levels:
            cur = -torch.nn.functional.max_pool2d(-cur[None, None], 2)[0, 0]


def test_bounds_bit_exact(mc, O, ctx, kitten, bunny):
    """mcull_compute_bounds vs buildMeshletCon on kitten and bunny: cone, apex, radius."""
    for scene in (kitten, bunny):
        blank = scene["meshlets"].copy()
        blank["cone"] = 0; blank["coneApex"] = 0; blank["radius"] = 0
        ctx.upload_meshlets(blank)
        ctx.upload_geometry(scene["vertices"], scene["meshlet_data"])
        ctx.compute_bounds()
        got = ctx.download_meshlets(len(blank))
        ref = scene["meshlets"]
        for f in ("cone", "coneApex", "radius"):
            a, b = got[f], ref[f]
            rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
            assert np.all((rel <= 1e-5) | (a == b)), f
            assert a.tobytes() == b.tobytes(), "%s not bit-exact" % f
        assert got.tobytes() == ref.tobytes()


def test_errors(mc, kitten, config1):
    g, t = config1["globals"], config1["transforms"]
    ctx = mc.Context(0)  # a fresh context: no pyramids, no ranges
    ctx.upload_meshlets(kitten["meshlets"])
    ctx.upload_transforms(t)
    bad = g.copy(); bad["meshletCount"] = 10 ** 6
    with pytest.raises(mc.McullError) as e:
        ctx.cull(bad, tests=3)
    assert e.value.status == mc.E_INVALID
    with pytest.raises(mc.McullError) as e:
        ctx.cull(g, tests=7, hiz_view=7)
    assert e.value.status == mc.E_STATE
    # capacity overflow is detected, never written past
    ctx.cull(g, tests=2, list_capacity=1000)
    with pytest.raises(mc.McullError) as e:
        ctx.download_list()
    assert e.value.status == mc.E_CAPACITY
    assert ctx.download_count() > 1000
    ctx.close()


@pytest.mark.parametrize("compaction", [0, 1, 2])
def test_empty_and_degenerate_scenes(mc, O, H, kitten, config1, compaction):
    """No instances, no meshlets, one meshlet, nothing visible, everything visible, and an output
    list that is too small (entries beyond the capacity are dropped, nothing is written past it)."""
    import torch

    g, t = config1["globals"], config1["transforms"]
    ml = kitten["meshlets"]
    c = mc.Context(0)
    c.upload_meshlets(ml)
    c.upload_transforms(t[:0])                                   # zero instances
    c.cull(g, tests=3, compaction=compaction)
    assert c.download_count() == 0 and len(c.download_list()) == 0
    c.upload_transforms(t[:5])
    g0 = g.copy(); g0["meshletCount"] = 0                        # zero meshlets per instance
    c.cull(g0, tests=3, compaction=compaction)
    assert c.download_count() == 0
    assert not c.download_commands(5).view(mc.CMD)["taskCount"].any()
    g1 = g.copy(); g1["meshletCount"] = 1                        # a single meshlet per instance
    ref = O.cull_scene(O.View(g1), t[:5], ml, tests=0)
    c.cull(g1, tests=0, compaction=compaction, write_mask=True)
    assert c.download_count() == 5 == ref.total and np.array_equal(c.download_mask(), ref.mask)
    away = H.create_globals(eye=(2.0, 2.0, 2.0), centre=(4.0, 4.0, 4.0), meshlet_count=len(ml))   # looking away
    ref = O.cull_scene(O.View(away), t, ml, tests=1)
    c.upload_transforms(t)
    c.cull(away, tests=1, compaction=compaction)
    assert c.download_count() == ref.total
    ref_all = O.cull_scene(O.View(g), t, ml, tests=0)            # no test: every pair survives
    c.cull(g, tests=0, compaction=compaction)
    assert c.download_count() == ref_all.total == len(t) * len(ml)
    if compaction == 0:
        assert c.download_list().tobytes() == ref_all.list.tobytes()
    # caller-owned list with a canary behind the capacity
    cap = 1000
    buf = torch.full((cap + 64, 2), -1, dtype=torch.int32, device="cuda")
    c.cull(g, tests=2, compaction=compaction, list_ptr=buf.data_ptr(), list_capacity=cap)
    total = c.download_count()
    assert total > cap
    assert bool((buf[cap:] == -1).all()), "entries were written past the capacity"
    if compaction == 0:
        ref2 = O.cull_scene(O.View(g), t, ml, tests=2)
        want = ref2.list.view("<u4").reshape(-1, 2)[:cap]
        assert np.array_equal(buf[:cap].cpu().numpy().view("<u4"), want)
    with pytest.raises(mc.McullError) as e:
        c.download_list()
    assert e.value.status == mc.E_CAPACITY
    c.close()


@pytest.mark.parametrize("n_inst", [700, 66000])
def test_caller_list_at_any_entry_offset_inside_a_cache_line(mc, H, n_inst):
    """The scatter pass deals its 16-byte stores from 128-byte line boundaries of the LIST's address: a caller's list may
    start at any 8-byte entry inside a line.  Both store forms (plain for small lists, non-temporal above 64 MB of
    pairs), with and without a capacity that cuts the list, canaries on both sides."""
    import torch

    per = 128
    cone, sphere = H.synthetic_bounds(n_inst * per, seed=3)
    t = H.synthetic_transforms(n_inst, extent=30.0, seed=4)
    g = H.create_globals(width=1920, height=1080)
    c = mc.Context(0)
    c.upload_bounds_soa(cone, sphere)
    c.upload_transforms(t)
    c.set_instance_ranges(np.arange(n_inst, dtype=np.uint32) * per, np.full(n_inst, per, dtype=np.uint32))
    c.cull(g, tests=3)
    total = c.download_count()
    want = torch.from_numpy(c.download_list().view("<i4").reshape(-1, 2).copy()).cuda()
    cmds_want = c.download_commands(n_inst).copy()
    assert 1000 < total < n_inst * per
    buf = torch.empty((total + 64, 2), dtype=torch.int32, device="cuda")
    for lead in (1, 2, 7, 15):
        for cap in (total, total - 37):
            buf.fill_(-1)
            torch.cuda.synchronize()
            c.cull(g, tests=3, list_ptr=buf.data_ptr() + 8 * lead, list_capacity=cap)
            assert c.download_count() == total
            assert bool((buf[:lead] == -1).all()) and bool((buf[lead + cap:] == -1).all()), "stores outside the list"
            assert torch.equal(buf[lead:lead + cap], want[:cap])
            assert np.array_equal(c.download_commands(n_inst), cmds_want)
    # the same through the other entry points that scatter into caller-owned lists
    bufs = [torch.full((total + 64, 2), -1, dtype=torch.int32, device="cuda") for _ in range(3)]
    cmds = torch.zeros((n_inst, 2), dtype=torch.int32, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    leads = (3, 9, 5)
    c.cull_views([g, g], [dict(tests=3, list_ptr=bufs[v].data_ptr() + 8 * leads[v], list_capacity=total) for v in range(2)])
    c.cull_static(g, bufs[2].data_ptr() + 8 * leads[2], total, cmds.data_ptr(), count.data_ptr(), tests=3)
    c.synchronize()
    for v in range(3):
        assert bool((bufs[v][:leads[v]] == -1).all()) and bool((bufs[v][leads[v] + total:] == -1).all())
        assert torch.equal(bufs[v][leads[v]:leads[v] + total], want)
    assert int(count.item()) == total
    c.close()


def test_task_group_command_form(mc, O, ctx, kitten, config1):
    """MCULL_CMD_NV_TASK_GROUPS: the NV record at the reference's granularity, taskCount = ceil(visible_i / 32) task
    workgroups of 32 (Engine.cpp:1475-1483), firstTask = the draw's offset in the list as in the per-meshlet form."""
    g, t = config1["globals"], config1["transforms"]
    n = len(t)
    ref = O.cull_scene(O.View(g), t, kitten["meshlets"], tests=3)
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(kitten["meshlets"])
    ctx.upload_transforms(t)
    for stride, offset in ((0, 0), (28, 20)):
        r = ctx.cull(g, tests=3, cmd_form=mc.CMD_NV_TASK_GROUPS, cmd_stride=stride, cmd_offset=offset)
        st = stride or 8
        raw = ctx.download_commands(n, st).reshape(n, st)
        cmds = raw[:, offset:offset + 8].copy().view(mc.CMD).reshape(-1)
        assert np.array_equal(cmds["taskCount"], (ref.commands["taskCount"] + 31) // 32)
        assert np.array_equal(cmds["firstTask"], ref.commands["firstTask"])
        assert ctx.download_list().tobytes() == ref.list.tobytes() and int(r.n_pairs) == ref.n_pairs
        # a pass-through task stage: workgroup w of draw i forwards list[firstTask + 32 w + lane], 32 w + lane < visible_i
        first = np.append(cmds["firstTask"].astype(np.int64), ref.total)
        covered = sum(min(32, int(first[i + 1] - first[i]) - 32 * w) for i in range(n) for w in range(int(cmds["taskCount"][i])))
        assert covered == ref.total
    with pytest.raises(mc.McullError) as e:
        ctx.cull(g, tests=3, cmd_form=3)
    assert e.value.status == mc.E_INVALID


def test_unknown_flag_bits_are_refused(mc, ctx, kitten, config1):
    import ctypes

    g, t = config1["globals"], config1["transforms"]
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(kitten["meshlets"])
    ctx.upload_transforms(t)
    gg = np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1)
    for flags, want in ((8, mc.E_INVALID), (0x80000000, mc.E_INVALID), (7, mc.OK), (0, mc.OK)):
        p = mc.CullParams(3, mc.COMPACT_ORDERED, 0, mc.CMD_NV_MESHLETS, 0, 0, 0, flags, None, 0, None, None)
        r = mc.CullResult()
        assert mc.lib.mcull_cull(ctx._h, gg.ctypes.data_as(ctypes.c_void_p), ctypes.byref(p), ctypes.byref(r)) == want
    assert mc.version() >= 110


def test_ext_command_form(mc, O, ctx, kitten, config1):
    """MCULL_CMD_EXT_GROUPS: {groupCountX = visible_i, 1, 1} records (12 bytes, or embedded at a
    caller-chosen stride/offset) plus the per-instance list offsets the EXT record has no room for."""
    import ctypes

    g, t = config1["globals"], config1["transforms"]
    n = len(t)
    ref = O.cull_scene(O.View(g), t, kitten["meshlets"], tests=3)
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(kitten["meshlets"])
    ctx.upload_transforms(t)
    for stride, offset in ((0, 0), (32, 16)):
        r = ctx.cull(g, tests=3, cmd_form=mc.CMD_EXT_GROUPS, cmd_stride=stride, cmd_offset=offset)
        st = stride or 12
        raw = ctx.download_commands(n, st).reshape(n, st)
        cmds = raw[:, offset:offset + 12].copy().view(mc.CMD_EXT).reshape(-1)
        assert np.array_equal(cmds["groupCountX"], ref.commands["taskCount"])
        assert (cmds["groupCountY"] == 1).all() and (cmds["groupCountZ"] == 1).all()
        if st > 12:  # bytes of the record outside the command stay zero
            rest = np.delete(raw, np.s_[offset:offset + 12], axis=1)
            assert not rest.any()
        assert r.first_task
        first = np.zeros(n, dtype="<u4")
        from conftest import loaded_hip_runtime
        hip = loaded_hip_runtime()
        ctx.synchronize()
        assert hip.hipMemcpy(ctypes.c_void_p(first.ctypes.data), ctypes.c_void_p(r.first_task), ctypes.c_size_t(n * 4), 2) == 0
        assert np.array_equal(first, ref.commands["firstTask"])
        assert ctx.download_list().tobytes() == ref.list.tobytes()
    with pytest.raises(mc.McullError):  # nothing to rebase in an EXT record
        ctx.rebase_commands(None, 0)
    with pytest.raises(mc.McullError):  # the 12-byte record does not fit an 8-byte stride
        ctx.cull(g, tests=3, cmd_form=mc.CMD_EXT_GROUPS, cmd_stride=8)
    r = ctx.cull(g, tests=3, compaction=1, cmd_form=mc.CMD_EXT_GROUPS)
    assert not r.first_task
    cmds = ctx.download_commands(n, 12).view(mc.CMD_EXT)
    assert np.array_equal(cmds["groupCountX"], ref.commands["taskCount"])
    ctx.cull(g, tests=3)  # back to the NV form for whoever shares the context


def test_scene_cache_upload(tmp_path, mc, O, ctx, kitten, bunny, H):
    """A scene written to the cache file and uploaded from it culls exactly like the arrays it was
    written from (mixed meshes through ranges), and its bounds can be recomputed on the GPU."""
    meshlets = np.concatenate([kitten["meshlets"], bunny["meshlets"]])
    data = np.concatenate([kitten["meshlet_data"], bunny["meshlet_data"]])
    nk, nb = len(kitten["meshlets"]), len(bunny["meshlets"])
    n = 200
    t = H.update_transforms(n)
    off = np.where(np.arange(n) % 2 == 0, 0, nk).astype("<u4")
    cnt = np.where(np.arange(n) % 2 == 0, nk, nb).astype("<u4")
    path = tmp_path / "scene.mcs"
    mc.save_scene(path, kitten["vertices"], None, kitten["meshlet_data"], kitten["meshlets"])
    ctx.upload_scene(path)  # no ranges in the file: uniform meshlet count, kitten only
    ctx.upload_transforms(t)
    g = H.create_globals(meshlet_count=nk)
    ref = O.cull_scene(O.View(g), t, kitten["meshlets"], tests=3)
    ctx.cull(g, tests=3)
    assert ctx.download_list().tobytes() == ref.list.tobytes()
    ctx.compute_bounds()
    assert ctx.download_meshlets(nk).tobytes() == kitten["meshlets"].tobytes()
    mc.save_scene(path, None, None, data, meshlets, off, cnt)  # mixed scene with ranges, no vertices
    ctx.upload_scene(path)
    g = H.create_globals(meshlet_count=0)
    ref = O.cull_scene(O.View(g), t, meshlets, ranges=(off, cnt), tests=3)
    ctx.cull(g, tests=3, write_mask=True)
    assert np.array_equal(ctx.download_mask(), ref.mask)
    check_against_oracle(mc, ctx, ref, n, 0)
    with pytest.raises(mc.McullError) as e:
        ctx.upload_scene(tmp_path / "missing.mcs")
    assert e.value.status == mc.E_IO
    ctx.set_instance_ranges(None, None)


def test_rebase_commands(mc, O, ctx, kitten, config1):
    """mcull_rebase_commands: firstTask += sum of the lower ranks' visible counts, read from a device
    array of gathered totals (the multi-GPU hand-off); taskCount unchanged; idempotent per call."""
    import torch

    g, t = config1["globals"], config1["transforms"]
    ref = O.cull_scene(O.View(g), t, kitten["meshlets"], tests=3)
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(kitten["meshlets"])
    ctx.upload_transforms(t)
    totals = torch.tensor([5, 7, 11, 13], dtype=torch.int64, device="cuda")
    for compaction in (0, 1):
        ctx.cull(g, tests=3, compaction=compaction)
        for rank, base in ((0, 0), (2, 12), (3, 23), (2, 12)):
            ctx.rebase_commands(totals.data_ptr(), rank)
            cmds = ctx.download_commands(len(t)).view(mc.CMD)
            assert np.array_equal(cmds["taskCount"], ref.commands["taskCount"])
            want = ref.commands["firstTask"] + base if compaction == 0 else np.full(len(t), base, dtype="<u4")
            assert np.array_equal(cmds["firstTask"], want)
    # the ordered cull after the rebases still starts from clean tile counters
    ctx.cull(g, tests=3)
    assert ctx.download_list().tobytes() == ref.list.tobytes()


def test_adversarial_bounds_and_transforms(mc, O, ctx, H):
    """NaN / infinite / denormal bounds, zero and negative radii, zero-length and NaN axes, coneW at
    and beyond its range, singular / NaN / mirrored transforms, spheres at the eye and on the planes:
    every test combination must still agree with the oracle bit for bit."""
    rs = np.random.RandomState(99)
    n_inst, per = 96, 192
    cone, sphere = H.synthetic_bounds(n_inst * per, seed=7)
    cone = cone.copy(); sphere = sphere.copy()
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-40, -1e-40, 3.4e38, -3.4e38, 1.0, -1.0, 0.5], dtype="<f4")
    idx = rs.choice(n_inst * per, 4000, replace=False)
    for k, i in enumerate(idx):
        if k % 2 == 0:
            cone[i, rs.randint(0, 4)] = special[rs.randint(len(special))]
        else:
            sphere[i, rs.randint(0, 4)] = special[rs.randint(len(special))]
    cone[idx[:200], :3] = 0.0                       # zero axis -> NaN after normalisation -> never cone-culled
    sphere[idx[200:300], 3] = -sphere[idx[200:300], 3]
    t = H.synthetic_transforms(n_inst, extent=20.0, seed=5)
    m = t["model"].reshape(n_inst, 4, 4).copy()
    m[3, :3, :3] = 0.0                              # singular
    m[7, 0, 0] = np.nan
    m[11, :3, :3] *= np.float32(-2.5)               # mirrored + scaled
    m[13, 3, :3] = [2.0, 2.0, 2.0]                  # at the eye
    m[17, :3, :3] *= np.float32(1e-30)
    m[19, 3, 0] = np.inf
    t["model"] = m.reshape(n_inst, 16)
    off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
    cnt = np.full(n_inst, per, dtype="<u4")
    w, h = 256, 128
    depth = H.synthetic_depth(w, h, n_discs=12, seed=4)
    depth.flat[rs.randint(0, w * h, 50)] = np.nan
    ctx.upload_bounds_soa(cone, sphere)
    ctx.upload_transforms(t)
    ctx.set_instance_ranges(off, cnt)
    ctx.hiz_build(depth, view=4)
    pyr = O.hiz_build(depth)
    with np.errstate(all="ignore"):
        for flip in (True, False):
            g = H.create_globals(eye=(2.0, 2.0, 2.0), width=w, height=h, flip_y=flip)
            view = O.View(g, pyr, w, h)
            for tests in (7, 5, 6, 3, 4):
                ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=tests)
                for compaction in (0, 1):
                    ctx.cull(g, tests=tests, hiz_view=4, compaction=compaction, write_mask=True)
                    assert np.array_equal(ctx.download_mask(), ref.mask), (flip, tests, compaction)
                    check_against_oracle(mc, ctx, ref, n_inst, compaction)
    ctx.set_instance_ranges(None, None)


def test_hostile_arguments_on_a_live_context(mc, kitten, config1):
    """Bad sizes, flags, indices and orders of calls on a working context: an error code and a message each
    time, and the context keeps working afterwards."""
    import ctypes as C

    g, t = config1["globals"], config1["transforms"]
    ml = np.ascontiguousarray(kitten["meshlets"])
    ctx = mc.Context(0)
    h = ctx._h
    L = mc.lib
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    gg = np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1)
    depth = np.zeros((4, 4), "<f4")
    res = mc.CullResult()
    # before anything is uploaded
    assert L.mcull_cull(h, ptr(gg), None, C.byref(res)) == mc.E_STATE
    assert L.mcull_download_count(h, C.byref(C.c_uint64())) == mc.E_STATE
    assert L.mcull_compute_bounds(h) == mc.E_STATE
    assert L.mcull_depth_rasterize(h, ptr(gg), 64, 64, 1) == mc.E_STATE
    assert L.mcull_select_result(h, 3) == mc.E_STATE and L.mcull_select_result(h, 99) == mc.E_INVALID
    assert L.mcull_rebase_commands(h, None, 0) == mc.E_STATE
    # sizes and flags
    assert L.mcull_upload_meshlets(h, ptr(ml), len(ml), 7) == mc.E_INVALID            # unknown memory kind
    assert L.mcull_upload_meshlets(h, ptr(ml), 1 << 44, mc.MEM_HOST) in (mc.E_NOMEM, mc.E_INVALID, mc.E_HIP)
    assert b"" != L.mcull_last_error(h)
    assert L.mcull_hiz_build(h, 0, ptr(depth), 0, 4, mc.MEM_HOST) == mc.E_INVALID     # empty image
    assert L.mcull_hiz_build(h, mc.MAX_VIEWS, ptr(depth), 4, 4, mc.MEM_HOST) == mc.E_INVALID
    assert L.mcull_hiz_build(h, 0, ptr(depth), 1 << 20, 1 << 20, mc.MEM_HOST) == mc.E_INVALID
    assert L.mcull_hiz_download_level(h, 0, 0, ptr(depth)) == mc.E_STATE              # nothing built in that slot
    # the context still works
    ctx.upload_meshlets(ml)
    ctx.upload_transforms(t)
    p = mc.CullParams()
    p.tests = 8                                                                       # unknown test bit
    assert L.mcull_cull(h, ptr(gg), C.byref(p), C.byref(res)) == mc.E_INVALID
    p.tests, p.compaction = 3, 9
    assert L.mcull_cull(h, ptr(gg), C.byref(p), C.byref(res)) == mc.E_INVALID
    p.compaction, p.cmd_form = 0, 5
    assert L.mcull_cull(h, ptr(gg), C.byref(p), C.byref(res)) == mc.E_INVALID
    p.cmd_form, p.cmd_stride, p.cmd_offset = 0, 8, 4                                   # the record does not fit its stride
    assert L.mcull_cull(h, ptr(gg), C.byref(p), C.byref(res)) == mc.E_INVALID
    p.cmd_stride, p.cmd_offset = 0, 0
    assert L.mcull_cull_views(h, 0, ptr(gg), None, None) == mc.E_INVALID
    assert L.mcull_cull_views(h, mc.MAX_VIEWS + 1, ptr(gg), None, None) == mc.E_INVALID
    assert L.mcull_depth_rasterize(h, ptr(gg), 0, 64, 1) == mc.E_INVALID
    assert L.mcull_cull(h, ptr(gg), C.byref(p), C.byref(res)) == mc.OK
    assert L.mcull_depth_rasterize(h, ptr(gg), 64, 64, 1) == mc.E_STATE                # no geometry uploaded
    assert L.mcull_download_commands(h, ptr(depth), 1 << 40) == mc.E_INVALID          # more than the buffer holds
    p.compaction = mc.COMPACT_WAVE_ATOMIC                                              # a cull that keeps no mask
    assert L.mcull_cull(h, ptr(gg), C.byref(p), C.byref(res)) == mc.OK and not res.mask
    assert L.mcull_download_mask(h, ptr(depth), 1) == mc.E_STATE
    assert L.mcull_emit_task_payloads(h, ptr(depth)) == mc.E_STATE
    assert ctx.download_count() > 0
    ctx.close()


def test_long_runs_of_empty_instances(mc, O, ctx, kitten, H):
    """A ranged scene in which almost every instance owns no meshlets (a sharded or LOD scene): the commands of the
    empty instances carry taskCount 0 and the firstTask of the next instance that owns some."""
    ml = kitten["meshlets"]
    n = 120_000
    t = H.synthetic_transforms(n, extent=3.0, seed=12)
    off = np.zeros(n, dtype="<u4")
    cnt = np.zeros(n, dtype="<u4")
    owners = [0, 1, 40_000, 40_001, 99_999, n - 2]
    for k, i in enumerate(owners):
        off[i], cnt[i] = (7 * k) % 50, 65 + 30 * k
    g = H.create_globals(eye=(6.0, 5.0, 4.0), meshlet_count=0)
    ref = O.cull_scene(O.View(g), t, ml, ranges=(off, cnt), tests=3)
    assert ref.total > 0
    ctx.upload_meshlets(ml)
    ctx.upload_transforms(t)
    ctx.set_instance_ranges(off, cnt)
    ctx.cull(g, tests=3, write_mask=True)
    assert np.array_equal(ctx.download_mask(), ref.mask)
    check_against_oracle(mc, ctx, ref, n, mc.COMPACT_ORDERED)
