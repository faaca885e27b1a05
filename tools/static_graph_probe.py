"""mcull_cull_static under stream capture: what a replayed graph writes, against a direct call (debugging aid)."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mc = importlib.import_module("vulkan-renderer_amd")
H = importlib.import_module("vulkan-renderer_amd.host")

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
    lst = torch.full((n_pairs, 2), -1, dtype=torch.int32, device="cuda:0")
    cmds = torch.zeros((n_inst, 2), dtype=torch.int32, device="cuda:0")
    total = torch.full((1,), -1, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    r = ctx.cull_static(g, lst.data_ptr(), n_pairs, cmds.data_ptr(), total.data_ptr(), tests=3)
    ctx.synchronize()
    print("direct:", int(total.item()))
    import ctypes as C
    hip = None
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            hip = C.CDLL(line.split()[-1]); break
    def peek(offset, n):
        a = np.zeros(n, dtype="<u4")
        assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), C.c_void_p(r.mask + offset), C.c_size_t(a.nbytes), 2) == 0
        return a
    print("tile counters after direct:", peek(16777216, 16))
    MODE = os.environ.get("PROBE_MODE", "torch")
    if MODE == "torch":
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            ctx.cull_static(g, lst.data_ptr(), n_pairs, cmds.data_ptr(), total.data_ptr(), tests=3)
        replay = graph.replay
    else:
        st = C.c_void_p(stream.cuda_stream)
        gr, ex = C.c_void_p(), C.c_void_p()
        assert hip.hipStreamBeginCapture(st, 0) == 0
        ctx.cull_static(g, lst.data_ptr(), n_pairs, cmds.data_ptr(), total.data_ptr(), tests=3)
        assert hip.hipStreamEndCapture(st, C.byref(gr)) == 0
        assert hip.hipGraphInstantiate(C.byref(ex), gr, None, None, 0) == 0
        def replay():
            assert hip.hipGraphLaunch(ex, st) == 0
    for k in range(3):
        total.fill_(-1)
        torch.cuda.synchronize()
        replay()
        torch.cuda.synchronize()
        print("replay", k, int(total.item()), int(cmds[:, 0].sum().item()), "tile counters:", peek(16777216, 16), "mask words:", peek(0, 4))
