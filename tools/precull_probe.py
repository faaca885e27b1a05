#!/usr/bin/env python3
"""Where does the mask pass's time go when the instance pre-cull removes work?  The 100 M-pair scene culled
with the bench's camera and with cameras that see nothing / everything, pre-cull on and off."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

os.environ.setdefault("MCULL_PRECULL_ADAPTIVE", "0")
mc = importlib.import_module("vulkan-renderer_amd")
host = importlib.import_module("vulkan-renderer_amd.host")
dev = torch.device("cuda", 0)
n_inst, per = 781250, 128
cone, sphere, xf = bench.make_scene_range(dev, 0, n_inst, per, seed=42)
if "--sorted" in sys.argv:
    # the same instances, those outside the bench camera's frustum first: the pre-cull then drops one contiguous
    # third of the scene instead of scattered instances
    g0 = host.create_globals(fov_deg=45.0, width=3840, height=2160, flip_y=True, eye=(51.0, 54.4, 40.8))
    pl = torch.from_numpy(np.asarray(g0["frustum"], dtype=np.float32).reshape(5, 4)).to(dev)
    centre = xf[:, 12:15] + 0.5
    dist = (centre @ pl[:, :3].T + pl[:, 3]).min(dim=1).values
    order = torch.argsort((dist > -1.0).to(torch.int32), stable=True)
    xf = xf[order].contiguous()
torch.cuda.synchronize()
stream = torch.cuda.Stream(device=dev)
ctx = mc.Context(0, stream=stream.cuda_stream)
ctx.upload_bounds_soa(cone.data_ptr(), sphere.data_ptr(), count=n_inst * per, mem=mc.MEM_DEVICE, adopt=True)
ctx.upload_transforms(xf.data_ptr(), count=n_inst, mem=mc.MEM_DEVICE)
off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
ctx.set_instance_ranges(off, np.full(n_inst, per, dtype="<u4"))
lst = torch.empty((n_inst * per, 2), device=dev, dtype=torch.int32)
cmds = torch.zeros((n_inst, 2), device=dev, dtype=torch.int32)
tot = torch.zeros(1, device=dev, dtype=torch.int64)
cams = {
    "bench camera": dict(eye=(51.0, 54.4, 40.8)),
    "sees nothing (looks away)": dict(eye=(60.0, 60.0, 60.0), centre=(120.0, 120.0, 120.0)),
    "sees everything (far away)": dict(eye=(400.0, 400.0, 400.0)),
}
for name, kw in cams.items():
    g = host.create_globals(fov_deg=45.0, width=3840, height=2160, flip_y=True, **kw)
    for pre in (True, False):
        def cull():
            ctx.cull(g, tests=3, list_ptr=lst.data_ptr(), list_capacity=n_inst * per, commands_ptr=cmds.data_ptr(),
                     total_ptr=tot.data_ptr(), precull=pre)
        for _ in range(60):
            cull()
        ctx.enable_timing(True)
        ms = []
        for _ in range(20):
            cull()
            ms.append(ctx.last_kernel_ms())
        ctx.enable_timing(False)
        marked, tested, _ = ctx.precull_stats()
        print("%-28s precull %-5s  %.4f ms per cull  visible %9d  groups marked %d / %d"
              % (name, pre, float(np.median(ms)), int(tot.item()), marked, tested), flush=True)
