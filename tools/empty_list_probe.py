#!/usr/bin/env python3
"""What a mask pass costs when the instance level leaves it nothing: 1 M instances of one 13 916-meshlet mesh, camera
looking away from the scene, device time of the whole cull (pre-pass, mask pass over an empty list inside a grid
sized for the scene, scan, scatter, commands).  Usage: tools/empty_list_probe.py [instances]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mc = importlib.import_module("vulkan-renderer_amd")
H = importlib.import_module("vulkan-renderer_amd.host")
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
per = 13916
dev = torch.device("cuda:0")
c_np, s_np = H.synthetic_bounds(per, seed=5)
cone, sphere = torch.from_numpy(c_np).to(dev), torch.from_numpy(s_np).to(dev)
xf = bench.make_transforms_range(dev, 0, n, 70.0, 77)
torch.cuda.synchronize()
ctx = mc.Context(0)
ctx.upload_bounds_soa(cone.data_ptr(), sphere.data_ptr(), count=per, mem=mc.MEM_DEVICE, adopt=True)
ctx.upload_transforms(xf.data_ptr(), count=n, mem=mc.MEM_DEVICE)
lst = torch.empty((1 << 20, 2), device=dev, dtype=torch.int32)
for name, centre in (("looking away", (400.0, 400.0, 400.0)), ("looking at the scene", (0.0, 0.0, 0.0))):
    g = H.create_globals(eye=(200.0, 200.0, 200.0), centre=centre, width=3840, height=2160, flip_y=False, meshlet_count=per)
    ctx.enable_timing(True)
    ms = []
    for k in range(6):
        ctx.cull(g, tests=mc.TEST_FRUSTUM | mc.TEST_CONE, list_ptr=lst.data_ptr(), list_capacity=1 << 20)
        ms.append(ctx.last_kernel_ms())
    ctx.enable_timing(False)
    marked, tested, _ = ctx.precull_stats()
    print("%-22s instances %d dropped %d: cull %.3f ms (best of %s)" % (name, tested, marked, min(ms[1:]), ["%.3f" % x for x in ms[1:]]))
ctx.close()
