"""GPU parity of the depth pre-pass stand-in (mcull_depth_rasterize) and of the two-phase frame it
enables: cull -> rasterise -> Hi-Z pyramid -> cull with the Hi-Z test.  Depth images bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype="<f4").view("<u4")


def test_two_phase_frame_kitten(mc, O, ctx, kitten, config1):
    """Config 1: phase 1 culls without Hi-Z, the survivors are rasterised, the pyramid is built from
    that image on the device and phase 2 culls against it; every stage equals the oracle's."""
    g, t = config1["globals"], config1["transforms"]
    w, h = 800, 600
    ml = kitten["meshlets"]
    ref1 = O.cull_scene(O.View(g), t, ml, tests=3)
    img_ref = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], ml, t, ref1.list, g, w, h)
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(ml)
    ctx.upload_geometry(kitten["vertices"], kitten["meshlet_data"])
    ctx.upload_transforms(t)
    ctx.cull(g, tests=3)
    ctx.depth_rasterize(g, w, h)
    img = ctx.depth_download()
    assert np.array_equal(_bits(img), _bits(img_ref)), "depth image must be bit-exact"
    assert 0.01 < (img > 0).mean() < 0.95
    ptr, dw, dh = ctx.depth_buffer()
    assert (dw, dh) == (w, h)
    ctx.hiz_build(ptr, view=0, width=w, height=h, mem=mc.MEM_DEVICE)
    pyr_ref = O.hiz_build(img_ref)
    for a, b in zip(ctx.hiz_download(view=0), O.hiz_levels(pyr_ref, w, h)):
        assert np.array_equal(_bits(a), _bits(b))
    ref2 = O.cull_scene(O.View(g, pyr_ref, w, h), t, ml, tests=7)
    ctx.cull(g, tests=7, hiz_view=0, write_mask=True)
    assert np.array_equal(ctx.download_mask(), ref2.mask)
    assert ctx.download_list().tobytes() == ref2.list.tobytes()
    assert 0 < ref2.total <= ref1.total


def test_large_triangles_accumulation_and_near_plane(mc, O, ctx, kitten, H):
    """Instances scaled up and pushed towards the camera: boxes of thousands of pixels (rasterised by
    the whole wave), triangles crossing the near plane (clipped), and a second pass accumulated into
    the same image with clear = 0."""
    w, h = 1920, 1080
    ml = kitten["meshlets"]
    n = 24
    t = H.update_transforms(n)
    model = t["model"].reshape(n, 4, 4).copy()      # [c][r]
    rs = np.random.RandomState(4)
    for i in range(n):
        s = np.float32([1.0, 2.0, 4.0, 8.0][i % 4])
        model[i, :3, :3] *= s
        model[i, 3, :3] = (rs.uniform(-1.0, 1.0, 3) * [1.5, 1.5, 1.5] + [0.6, 0.6, 0.6]).astype("<f4")
    t["model"] = model.reshape(n, 16)
    g = H.create_globals(width=w, height=h, meshlet_count=len(ml))
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(ml)
    ctx.upload_geometry(kitten["vertices"], kitten["meshlet_data"])
    ctx.upload_transforms(t)
    ref_a = O.cull_scene(O.View(g), t, ml, tests=1)          # frustum only: back faces reach the rasteriser
    ctx.cull(g, tests=1)
    ctx.depth_rasterize(g, w, h)
    img_a = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], ml, t, ref_a.list, g, w, h)
    got_a = ctx.depth_download()
    assert np.array_equal(_bits(got_a), _bits(img_a))
    assert (img_a > 0).mean() > 0.2                          # the enlarged instances really fill the screen
    # second pass from another camera accumulated into the same image
    g2 = H.create_globals(eye=(3.0, -2.0, 2.5), width=w, height=h, meshlet_count=len(ml))
    ref_b = O.cull_scene(O.View(g2), t, ml, tests=3)
    ctx.cull(g2, tests=3)
    ctx.depth_rasterize(g2, w, h, clear=False)
    img_b = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], ml, t, ref_b.list, g2, w, h, depth=img_a)
    assert np.array_equal(_bits(ctx.depth_download()), _bits(img_b))
    assert not np.array_equal(img_a, img_b)
    # a different size reallocates and clears
    ctx.depth_rasterize(g2, 320, 200, clear=False)
    small = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], ml, t, ref_b.list,
                              H.create_globals(eye=(3.0, -2.0, 2.5), width=w, height=h, meshlet_count=len(ml)), 320, 200)
    assert np.array_equal(_bits(ctx.depth_download()), _bits(small))


def test_mixed_meshes_through_ranges(mc, O, ctx, kitten, bunny, H):
    """The list's meshlet ids are global: kitten and bunny instances through per-instance ranges."""
    w, h = 640, 480
    nk, nb = len(kitten["meshlets"]), len(bunny["meshlets"])
    bun = bunny["meshlets"].copy()
    bun["dataOffset"] += len(kitten["meshlet_data"])
    meshlets = np.concatenate([kitten["meshlets"], bun])
    data = np.concatenate([kitten["meshlet_data"], bunny["meshlet_data"] + 0])
    # bunny's global vertex ids are shifted behind kitten's vertices
    nkv = len(kitten["vertices"])
    data = data.copy()
    for m in bun:
        o, nv = int(m["dataOffset"]), int(m["vertexCount"])
        data[o:o + nv] += nkv
    verts = np.concatenate([kitten["vertices"], bunny["vertices"]])
    n = 40
    t = H.update_transforms(n)
    off = np.where(np.arange(n) % 2 == 0, 0, nk).astype("<u4")
    cnt = np.where(np.arange(n) % 2 == 0, nk, nb).astype("<u4")
    g = H.create_globals(width=w, height=h, meshlet_count=0)
    ctx.upload_meshlets(meshlets)
    ctx.upload_geometry(verts, data)
    ctx.upload_transforms(t)
    ctx.set_instance_ranges(off, cnt)
    ref = O.cull_scene(O.View(g), t, meshlets, ranges=(off, cnt), tests=3)
    ctx.cull(g, tests=3)
    ctx.depth_rasterize(g, w, h)
    want = O.rasterize_depth(verts, data, meshlets, t, ref.list, g, w, h)
    assert np.array_equal(_bits(ctx.depth_download()), _bits(want))
    assert (want > 0).any()
    ctx.set_instance_ranges(None, None)


def test_depth_errors(mc, H, kitten, config1):
    c = mc.Context(0)
    g = config1["globals"]
    with pytest.raises(mc.McullError) as e:
        c.depth_rasterize(g, 64, 64)
    assert e.value.status == mc.E_STATE                       # no cull yet
    with pytest.raises(mc.McullError) as e:
        c.depth_buffer()
    assert e.value.status == mc.E_STATE
    c.upload_meshlets(kitten["meshlets"])
    c.upload_transforms(config1["transforms"][:4])
    c.cull(g, tests=3)
    with pytest.raises(mc.McullError) as e:
        c.depth_rasterize(g, 64, 64)
    assert e.value.status == mc.E_STATE                       # geometry not uploaded
    c.upload_geometry(kitten["vertices"], kitten["meshlet_data"])
    c.cull(g, tests=3)
    with pytest.raises(mc.McullError) as e:
        c.depth_rasterize(g, 0, 64)
    assert e.value.status == mc.E_INVALID
    c.depth_rasterize(g, 64, 64)
    assert c.depth_download().shape == (64, 64)
    c.close()


def test_frame_can_be_captured_as_a_hip_graph(mc, O, kitten, config1):
    """After one ordinary run the two-phase frame allocates nothing and never synchronises, so it can
    be captured on the context's stream and replayed; the replay reproduces the frame (also after the
    transforms in the captured buffer have been overwritten in place)."""
    import torch

    g, t = config1["globals"], config1["transforms"][:300]
    ml = kitten["meshlets"]
    w, h = 640, 480
    stream = torch.cuda.Stream()
    c = mc.Context(0, stream=stream.cuda_stream)
    c.upload_meshlets(ml)
    c.upload_geometry(kitten["vertices"], kitten["meshlet_data"])
    xf = torch.from_numpy(t["model"].copy()).cuda()
    torch.cuda.synchronize()
    c.upload_transforms(xf.data_ptr(), count=len(t), mem=mc.MEM_DEVICE)

    def frame():
        c.cull(g, tests=3)
        c.depth_rasterize(g, w, h)
        ptr, _, _ = c.depth_buffer()
        c.hiz_build(ptr, view=0, width=w, height=h, mem=mc.MEM_DEVICE)
        c.cull(g, tests=7, hiz_view=0)

    frame()
    c.synchronize()
    want_total, want_list, want_depth = c.download_count(), c.download_list().copy(), c.depth_download().copy()
    ref1 = O.cull_scene(O.View(g), t, ml, tests=3)
    img = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], ml, t, ref1.list, g, w, h)
    ref2 = O.cull_scene(O.View(g, O.hiz_build(img), w, h), t, ml, tests=7)
    assert want_total == ref2.total and want_list.tobytes() == ref2.list.tobytes()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        frame()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert c.download_count() == want_total
    assert c.download_list().tobytes() == want_list.tobytes()
    assert np.array_equal(c.depth_download().view("<u4"), want_depth.view("<u4"))
    c.close()


def test_captured_frame_follows_a_moving_camera(mc, O, H, kitten, config1):
    """Globals in device memory: the frame is captured ONCE and replayed for several cameras, the 212
    bytes being refreshed between replays; every replay equals the oracle's frame for that camera, and
    the device-resident form equals the ordinary one bit for bit."""
    import torch

    t = config1["transforms"][:300]
    ml = kitten["meshlets"]
    w, h = 640, 480
    stream = torch.cuda.Stream()
    c = mc.Context(0, stream=stream.cuda_stream)
    c.upload_meshlets(ml)
    c.upload_geometry(kitten["vertices"], kitten["meshlet_data"])
    c.upload_transforms(t)
    cameras = [H.create_globals(eye=eye, centre=centre, fov_deg=fov, width=w, height=h, flip_y=flip, meshlet_count=len(ml))
               for eye, centre, fov, flip in (((2.0, 2.0, 2.0), (0.0, 0.0, 0.0), 45.0, True), ((-6.0, 3.0, 1.0), (1.0, 0.0, 0.5), 60.0, True),
                                              ((0.5, -9.0, 4.0), (0.0, 0.0, 0.0), 35.0, False), ((0.0, 0.0, 0.0), (3.0, 1.0, 0.0), 90.0, True))]
    g_dev = torch.zeros(212, dtype=torch.uint8, device="cuda")
    staged = [torch.from_numpy(np.frombuffer(g.tobytes(), dtype=np.uint8).copy()).cuda() for g in cameras]
    torch.cuda.synchronize()

    def frame():
        c.cull_device_globals(g_dev.data_ptr(), len(ml), tests=3)
        c.depth_rasterize_device_globals(g_dev.data_ptr(), w, h)
        ptr, _, _ = c.depth_buffer()
        c.hiz_build(ptr, view=0, width=w, height=h, mem=mc.MEM_DEVICE)
        c.cull_device_globals(g_dev.data_ptr(), len(ml), tests=7, hiz_view=0, write_mask=True)

    def oracle_frame(g):
        r1 = O.cull_scene(O.View(g), t, ml, tests=3)
        img = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], ml, t, r1.list, g, w, h)
        return img, O.cull_scene(O.View(g, O.hiz_build(img), w, h), t, ml, tests=7)

    with torch.cuda.stream(stream):
        g_dev.copy_(staged[0])
    frame()  # the ordinary run: allocates what the frame needs
    c.synchronize()
    img, ref = oracle_frame(cameras[0])
    assert c.download_count() == ref.total and c.download_list().tobytes() == ref.list.tobytes()
    assert np.array_equal(c.download_mask(), ref.mask)
    assert np.array_equal(c.depth_download().view("<u4"), img.view("<u4"))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        frame()
    totals = []
    for k in (1, 2, 3, 0, 2):
        with torch.cuda.stream(stream):
            g_dev.copy_(staged[k])
        stream.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        img, ref = oracle_frame(cameras[k])
        assert c.download_count() == ref.total, k
        assert c.download_list().tobytes() == ref.list.tobytes(), k
        assert np.array_equal(c.download_mask(), ref.mask), k
        assert np.array_equal(c.depth_download().view("<u4"), img.view("<u4")), k
        cmds = c.download_commands(len(t)).view(mc.CMD).reshape(-1)
        assert np.array_equal(cmds["taskCount"], ref.commands["taskCount"]) and np.array_equal(cmds["firstTask"], ref.commands["firstTask"])
        totals.append(ref.total)
    assert len(set(totals)) >= 3  # the cameras really see different things
    with pytest.raises(mc.McullError):  # the atomic compaction modes take their constants as kernel arguments only
        p = mc.CullParams(3, mc.COMPACT_WAVE_ATOMIC, 0, 0, 0, 0, 0, 0, None, 0, None, None)
        c._chk(mc.lib.mcull_cull_device_globals(c._h, g_dev.data_ptr(), len(ml), p, None))
    c.close()


def test_adversarial_vertices(mc, O, ctx, kitten, H):
    """NaN / infinite / huge vertex positions and singular or NaN transforms: the image still equals
    the oracle's bit for bit (such triangles produce NaN areas or depths and draw nothing, or cover
    the whole screen box and are filtered per pixel)."""
    rs = np.random.RandomState(12)
    ml = kitten["meshlets"]
    verts = kitten["vertices"].copy()
    special = np.array([np.nan, np.inf, -np.inf, 3.0e38, -3.0e38, 1e-40, 0.0, -0.0], dtype="<f4")
    for i in rs.choice(len(verts), 300, replace=False):
        verts["pos"][i, rs.randint(0, 3)] = special[rs.randint(len(special))]
    n = 40
    t = H.update_transforms(n)
    m = t["model"].reshape(n, 4, 4).copy()
    m[2, :3, :3] = 0.0
    m[5, 1, 1] = np.nan
    m[9, :3, :3] *= np.float32(30.0)
    m[9, 3, :3] = [2.0, 2.0, 2.0]
    m[12, 3, 2] = np.inf
    t["model"] = m.reshape(n, 16)
    w, h = 96, 64
    g = H.create_globals(width=w, height=h, meshlet_count=len(ml))
    ctx.set_instance_ranges(None, None)
    ctx.upload_meshlets(ml)
    ctx.upload_geometry(verts, kitten["meshlet_data"])
    ctx.upload_transforms(t)
    with np.errstate(all="ignore"):
        ref = O.cull_scene(O.View(g), t, ml, tests=0)
        ctx.cull(g, tests=0)
        ctx.depth_rasterize(g, w, h)
        want = O.rasterize_depth(verts, kitten["meshlet_data"], ml, t, ref.list, g, w, h)
    got = ctx.depth_download()
    assert np.array_equal(_bits(got), _bits(want))
    assert (want > 0).any() and not np.isnan(want).any()


def test_captured_multi_view_cull_follows_moving_cameras(mc, O, H, kitten, config1):
    """mcull_cull_views_device_globals captured once: three views per replay, their Globals refreshed in the
    device array between replays; each view of each replay equals the oracle's cull for that camera."""
    import torch

    t = config1["transforms"][:200]
    ml = kitten["meshlets"]
    w, h, nv = 320, 240, 3
    stream = torch.cuda.Stream()
    c = mc.Context(0, stream=stream.cuda_stream)
    c.upload_meshlets(ml)
    c.upload_transforms(t)
    pyrs = []
    for v in range(nv):
        depth = H.synthetic_depth(w, h, n_discs=10, seed=30 + v)
        c.hiz_build(depth, view=v)
        pyrs.append(O.hiz_build(depth))
    params = [dict(tests=7, hiz_view=v, write_mask=True) for v in range(nv)]

    def cameras(frame):
        return [H.create_globals(eye=(6.0 * np.cos(0.7 * frame + v), 6.0 * np.sin(0.7 * frame + v), 2.0 + v), fov_deg=50.0, width=w,
                                 height=h, flip_y=bool(v & 1), meshlet_count=len(ml)) for v in range(nv)]

    def stage(views):
        packed = np.concatenate([np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1) for g in views])
        return torch.from_numpy(packed.view("u1").reshape(-1).copy()).cuda()

    g_dev = torch.zeros(nv * 212, dtype=torch.uint8, device="cuda")
    staged = [stage(cameras(f)) for f in range(3)]
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        g_dev.copy_(staged[0])
    c.cull_views(nv, params, device_globals_ptr=g_dev.data_ptr(), meshlet_count=len(ml))  # the allocating run
    c.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        c.cull_views(nv, params, device_globals_ptr=g_dev.data_ptr(), meshlet_count=len(ml))
    for f in (1, 2, 0):
        with torch.cuda.stream(stream):
            g_dev.copy_(staged[f])
        stream.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        for v, g in enumerate(cameras(f)):
            ref = O.cull_scene(O.View(g, pyrs[v], w, h), t, ml, tests=7)
            c.select_result(v)
            assert c.download_count() == ref.total, (f, v)
            assert c.download_list().tobytes() == ref.list.tobytes(), (f, v)
            assert np.array_equal(c.download_mask(), ref.mask), (f, v)
    c.close()


def _screen_filling_scene(mc, rs, n_quads):
    """Quads far larger than the screen, at different depths and tilts, both windings (one of each pair faces the camera)."""
    verts = np.zeros(4 * n_quads, dtype=mc.VERTEX)
    meshlets = np.zeros(n_quads, dtype=mc.MESHLET)
    data = []
    for q in range(n_quads):
        z = rs.uniform(-3.0, 1.0)
        tilt = rs.uniform(-0.02, 0.02, 4)
        verts["pos"][4 * q:4 * q + 4, :3] = [[-40, -40, z + tilt[0] * 40], [40, -40, z + tilt[1] * 40], [40, 40, z + tilt[2] * 40], [-40, 40, z + tilt[3] * 40]]
        meshlets[q]["dataOffset"] = sum(len(x) for x in data)
        meshlets[q]["vertexCount"], meshlets[q]["triangleCount"] = 4, 4
        data += [np.arange(4 * q, 4 * q + 4, dtype="<u4"), np.array([0, 1, 2, 0, 2, 3, 0, 2, 1, 0, 3, 2], dtype="u1").view("<u4")]
    return verts, meshlets, np.concatenate(data).astype("<u4")


@pytest.mark.parametrize("queue", [None, "3,40", "1,1", "0,0", "1000,7"])
def test_huge_triangles_go_through_the_job_queue(mc, O, monkeypatch, queue):
    """Boxes above 512 pixels are cut into 64x64 pieces for a second kernel.  Same image as the oracle's, with the
    default queue and with queues so small that some or all of the triangles are walked in place instead."""
    if queue is not None:
        monkeypatch.setenv("MCULL_RASTER_QUEUE", queue)
    import subprocess, sys, os, textwrap

    # the queue size is read once per process: every setting runs in a process of its own
    code = textwrap.dedent('''
        import importlib, sys
        import numpy as np
        sys.path.insert(0, sys.argv[1])
        sys.path.insert(0, sys.argv[1] + "/tests")
        mc = importlib.import_module("vulkan-renderer_amd")
        import oracle as O
        from test_gpu_depth import _screen_filling_scene
        rs = np.random.RandomState(3)
        verts, meshlets, data = _screen_filling_scene(mc, rs, 6)
        t = np.zeros(2, dtype=mc.TRANSFORM)
        t["model"][0] = np.eye(4, dtype="<f4").reshape(-1)
        m = np.eye(4, dtype="<f4"); m[3, :3] = [3.0, -2.0, 0.5]; t["model"][1] = m.reshape(-1)
        with mc.Context(0) as ctx:
            ctx.upload_meshlets(meshlets)
            ctx.upload_geometry(verts, data)
            ctx.compute_bounds()
            ctx.upload_transforms(t)
            for (w, h), eye in (((800, 600), (0.0, 0.0, 5.0)), ((333, 517), (1.0, 2.0, 4.0)), ((1024, 1024), (0.0, 0.5, 2.0))):
                g = mc.create_globals(w, h, len(meshlets), eye=eye, up=(0.0, 1.0, 0.0))
                for rep in range(2):  # the second call starts from the counters the first one left
                    ctx.cull(g, tests=0)
                    ctx.depth_rasterize(g, w, h)
                    got = ctx.depth_download()
                want = O.rasterize_depth(verts, data, meshlets, t, ctx.download_list(), g, w, h)
                assert (want > 0).mean() > 0.9
                assert np.array_equal(got.view("<u4"), want.view("<u4")), (w, h, int((got.view("<u4") != want.view("<u4")).sum()))
        print("huge ok")
    ''')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0 and "huge ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_transforms_replaced_since_the_cull_are_reported(mc, H, kitten):
    """The list's drawIDs index the transforms: a rasteriser call after an upload of a different number of them is
    refused (MCULL_E_STATE) instead of reading past the new array."""
    ml = kitten["meshlets"]
    g = H.create_globals(width=64, height=64, meshlet_count=len(ml))
    with mc.Context(0) as c:
        c.upload_meshlets(ml)
        c.upload_geometry(kitten["vertices"], kitten["meshlet_data"])
        c.upload_transforms(mc.reference_transforms(8))
        c.cull(g, tests=3)
        c.depth_rasterize(g, 64, 64)
        c.upload_transforms(mc.reference_transforms(4))
        with pytest.raises(mc.McullError) as e:
            c.depth_rasterize(g, 64, 64)
        assert e.value.status == mc.E_STATE
        c.cull(g, tests=3)
        c.depth_rasterize(g, 64, 64)
