"""mcull_cull_static, the entry point that allocates nothing, against mcull_cull on the same inputs.

Called through the C ABI twice in a row on one GPU, the second time with a smaller scene, so that what the
first call left in the GPU's static scratch array (ballots, tile counters, per-instance offsets) must not
leak into the second.  Bar: list, commands, count and mask byte-identical to mcull_cull's (which the other
GPU tests hold against the oracle).  The property the entry point exists for - no allocation - is checked by
calling it under stream capture, where an allocation invalidates the capture, and replaying the graph.
Collected last (file name) and self-contained (own contexts), so it passes alone in a fresh process as well
as after the other tests.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def hip_runtime():
    """The HIP runtime this process has already loaded (torch's), found through the process's own mappings."""
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError("no HIP runtime loaded")


def run_both(mc, ctx, torch, g, n_inst, n_pairs, n_groups, tests, stride=8, offset=0):
    dev = "cuda:%d" % ctx.device
    out = {}
    for name in ("ref", "static"):
        lst = torch.full((max(n_pairs, 1), 2), -1, dtype=torch.int32, device=dev)
        cmds = torch.full((max(n_inst, 1) * stride,), 0x5A, dtype=torch.uint8, device=dev)
        total = torch.full((1,), -1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        if name == "ref":
            ctx.cull(g, tests=tests, cmd_stride=stride, cmd_offset=offset, write_mask=True, list_ptr=lst.data_ptr(),
                     list_capacity=n_pairs, commands_ptr=cmds.data_ptr(), total_ptr=total.data_ptr())
            ctx.synchronize()
            mask = ctx.download_mask()
        else:
            r = ctx.cull_static(g, lst.data_ptr(), n_pairs, cmds.data_ptr(), total.data_ptr(), tests=tests,
                                cmd_stride=stride, cmd_offset=offset)
            ctx.synchronize()
            assert int(r.n_groups) == n_groups and int(r.n_pairs) == n_pairs
            mask = np.zeros(n_groups, dtype="<u8")
            if n_groups:
                assert r.mask, "a cull with groups reports where its ballots are"
                hip = hip_runtime()
                assert hip.hipMemcpy(mask.ctypes.data_as(C.c_void_p), C.c_void_p(r.mask), C.c_size_t(mask.nbytes), 2) == 0
        n = int(total.item())
        out[name] = (n, lst[:n].cpu().numpy().tobytes(), cmds.cpu().numpy().tobytes(), mask.tobytes())
    assert out["ref"][0] == out["static"][0], "visible count differs"
    assert out["ref"][1] == out["static"][1], "visible list differs"
    assert out["ref"][2] == out["static"][2], "indirect commands differ"
    assert out["ref"][3] == out["static"][3], "visibility mask differs"
    return out["ref"][0]


def test_static_cull_equals_cull_on_two_calls_in_a_row(mc, H):
    torch = pytest.importorskip("torch")
    with mc.Context(0) as ctx:
        # first call: 3000 instances x 200 unique meshlets (ragged groups: 200 = 3 x 64 + 8), Hi-Z on
        n_inst, per = 3000, 200
        cone, sphere = H.synthetic_bounds(n_inst * per, seed=21)
        t = H.synthetic_transforms(n_inst, extent=50.0, seed=22)
        off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
        cnt = np.full(n_inst, per, dtype="<u4")
        cnt[7] = 0  # an instance without meshlets
        w, h = 512, 256
        g = H.create_globals(eye=(150.0, 140.0, 130.0), fov_deg=50.0, width=w, height=h, flip_y=False)
        ctx.upload_bounds_soa(cone, sphere)
        ctx.upload_transforms(t)
        ctx.set_instance_ranges(off, cnt)
        ctx.hiz_build(H.synthetic_depth(w, h, n_discs=30, seed=23), view=0)
        n_groups = int(((cnt.astype(np.int64) + 63) // 64).sum())
        vis1 = run_both(mc, ctx, torch, g, n_inst, int(cnt.sum()), n_groups, mc.TEST_ALL, stride=28, offset=20)
        assert 0 < vis1 < int(cnt.sum())

        # second call, smaller input: 500 instances sharing meshlets [0, 70) (uniform count, no ranges)
        n2, per2 = 500, 70
        ctx.set_instance_ranges(None, None)
        ctx.upload_transforms(t[:n2])
        g2 = H.create_globals(eye=(150.0, 140.0, 130.0), fov_deg=50.0, width=w, height=h, flip_y=True, meshlet_count=per2)
        vis2 = run_both(mc, ctx, torch, g2, n2, n2 * per2, n2 * 2, mc.TEST_FRUSTUM | mc.TEST_CONE)
        assert 0 < vis2 < n2 * per2

        # third: nothing to cull at all (meshletCount 0), then the limits
        g3 = H.create_globals(width=w, height=h, meshlet_count=0)
        assert run_both(mc, ctx, torch, g3, n2, 0, 0, mc.TEST_ALL) == 0


def test_static_cull_rejects_what_does_not_fit(mc, H):
    torch = pytest.importorskip("torch")
    with mc.Context(0) as ctx:
        cone, sphere = H.synthetic_bounds(64 * 40, seed=3)
        ctx.upload_bounds_soa(cone, sphere)
        buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0")
        g = H.create_globals(meshlet_count=64 * 40)
        # 60 000 instances x 40 groups = 2.4 M groups > MCULL_STATIC_MAX_GROUPS
        ctx.upload_transforms(H.synthetic_transforms(60000, extent=5.0, seed=1))
        with pytest.raises(mc.McullError) as e:
            ctx.cull_static(g, buf.data_ptr(), 16, buf.data_ptr(), buf.data_ptr(), tests=3)
        assert e.value.status == mc.E_CAPACITY and "static scratch" in str(e.value)
        # caller-owned outputs are mandatory, ordered compaction only
        ctx.upload_transforms(H.synthetic_transforms(10, extent=5.0, seed=1))
        with pytest.raises(mc.McullError) as e:
            ctx.cull_static(g, None, 16, buf.data_ptr(), buf.data_ptr())
        assert e.value.status == mc.E_INVALID
        g1 = H.create_globals(meshlet_count=1)
        ctx.upload_transforms(H.synthetic_transforms(mc.STATIC_MAX_INSTANCES + 1, extent=5.0, seed=1))
        with pytest.raises(mc.McullError) as e:
            ctx.cull_static(g1, buf.data_ptr(), 16, buf.data_ptr(), buf.data_ptr(), tests=1)
        assert e.value.status == mc.E_CAPACITY


def test_static_cull_is_capturable_and_the_replayed_graph_equals_cull(mc, H):
    """An allocation (hipMalloc / hipFree, torch's allocator) inside a stream capture invalidates the capture, so a
    call that captures cleanly allocated nothing; the replayed graph must write what mcull_cull writes."""
    torch = pytest.importorskip("torch")
    stream = torch.cuda.Stream(device="cuda:0")
    with mc.Context(0, stream=stream.cuda_stream) as ctx:
        n_inst, per = 2000, 96
        cone, sphere = H.synthetic_bounds(n_inst * per, seed=31)
        t = H.synthetic_transforms(n_inst, extent=40.0, seed=32)
        off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
        cnt = np.full(n_inst, per, dtype="<u4")
        g = H.create_globals(eye=(120.0, 110.0, 100.0), fov_deg=50.0, width=640, height=360, flip_y=False)
        ctx.upload_bounds_soa(cone, sphere)
        ctx.upload_transforms(t)
        ctx.set_instance_ranges(off, cnt)
        n_pairs = n_inst * per
        tests = mc.TEST_FRUSTUM | mc.TEST_CONE
        dev = "cuda:0"
        bufs = {}
        for name in ("ref", "graph"):
            bufs[name] = (torch.full((n_pairs, 2), -1, dtype=torch.int32, device=dev), torch.zeros((n_inst, 2), dtype=torch.int32, device=dev),
                          torch.full((1,), -1, dtype=torch.int64, device=dev))
        lst, cmds, total = bufs["ref"]
        ctx.cull(g, tests=tests, list_ptr=lst.data_ptr(), list_capacity=n_pairs, commands_ptr=cmds.data_ptr(), total_ptr=total.data_ptr())
        ctx.synchronize()
        lst, cmds, total = bufs["graph"]
        ptrs = (lst.data_ptr(), cmds.data_ptr(), total.data_ptr())
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=stream):  # global capture mode: an allocating call would raise / invalidate
            ctx.cull_static(g, ptrs[0], n_pairs, ptrs[1], ptrs[2], tests=tests)
        assert int(total.item()) == -1, "capture records the launches, it does not run them"
        for _ in range(2):
            total.fill_(-1)
            graph.replay()
            torch.cuda.synchronize()
            n = int(total.item())
            assert n == int(bufs["ref"][2].item()) and 0 < n < n_pairs
            assert torch.equal(lst[:n], bufs["ref"][0][:n]) and torch.equal(cmds, bufs["ref"][1])


def test_static_cull_refuses_overlapping_use_and_unsupported_options(mc, H):
    torch = pytest.importorskip("torch")
    s1, s2 = torch.cuda.Stream(device="cuda:0"), torch.cuda.Stream(device="cuda:0")
    with mc.Context(0, stream=s1.cuda_stream) as a, mc.Context(0, stream=s2.cuda_stream) as b:
        n_inst, per = 8000, 128  # a millisecond or so of kernels: still running when the second call arrives
        cone, sphere = H.synthetic_bounds(n_inst * per, seed=41)
        t = H.synthetic_transforms(n_inst, extent=40.0, seed=42)
        off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
        cnt = np.full(n_inst, per, dtype="<u4")
        g = H.create_globals(eye=(120.0, 110.0, 100.0), fov_deg=50.0, width=640, height=360, flip_y=False)
        outs = []
        for ctx in (a, b):
            ctx.upload_bounds_soa(cone, sphere)
            ctx.upload_transforms(t)
            ctx.set_instance_ranges(off, cnt)
            ctx.synchronize()
            outs.append((torch.empty((n_inst * per, 2), dtype=torch.int32, device="cuda:0"), torch.zeros((n_inst, 2), dtype=torch.int32, device="cuda:0"),
                         torch.zeros((1,), dtype=torch.int64, device="cuda:0")))
        torch.cuda.synchronize()
        args = [(o[0].data_ptr(), n_inst * per, o[1].data_ptr(), o[2].data_ptr()) for o in outs]
        # a long-running kernel on stream 1 in front of the cull keeps the first call's event pending for certain
        with torch.cuda.stream(s1):
            torch.cuda._sleep(50_000_000)
        a.cull_static(g, *args[0], tests=3)
        with pytest.raises(mc.McullError) as e:
            b.cull_static(g, *args[1], tests=3)
        assert e.value.status == mc.E_STATE and "in use" in str(e.value)
        a.cull_static(g, *args[0], tests=3)  # the same stream is ordered behind itself: allowed
        torch.cuda.synchronize()
        b.cull_static(g, *args[1], tests=3)  # the first user has finished
        b.synchronize()
        assert int(outs[1][2].item()) == int(outs[0][2].item()) > 0
        # options the entry point does not honour are refused, not ignored
        p = mc.CullParams(3, mc.COMPACT_ORDERED, 0, mc.CMD_NV_MESHLETS, 0, 0, 1, 0, *args[1])
        r = mc.CullResult()
        gg = np.ascontiguousarray(g, dtype=mc.GLOBALS).reshape(1)
        assert mc.lib.mcull_cull_static(b._h, gg.ctypes.data_as(C.c_void_p), C.byref(p), C.byref(r)) == mc.E_INVALID
        p = mc.CullParams(3, mc.COMPACT_ORDERED, 0, mc.CMD_NV_MESHLETS, 0, 0, 0, 4, *args[1])
        assert mc.lib.mcull_cull_static(b._h, gg.ctypes.data_as(C.c_void_p), C.byref(p), C.byref(r)) == mc.E_INVALID
