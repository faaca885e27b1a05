#!/usr/bin/env python3
"""Mask pass of the 100 M-pair scene, three launches back to back, then three with 2 GiB of unrelated data streamed
through the caches before each: under rocprofv3 --pmc this shows how many of the scene's lines a repeated launch
finds on the die (Infinity Cache) instead of fetching them."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

mc = importlib.import_module("vulkan-renderer_amd")
host = importlib.import_module("vulkan-renderer_amd.host")
dev = torch.device("cuda", 0)
n_inst, per = 781250, 128
cone, sphere, xf = bench.make_scene_range(dev, 0, n_inst, per, seed=42)
junk = torch.zeros(1 << 29, dtype=torch.float32, device=dev)  # 2 GiB
torch.cuda.synchronize()
ctx = mc.Context(0)
ctx.upload_bounds_soa(cone.data_ptr(), sphere.data_ptr(), count=n_inst * per, mem=mc.MEM_DEVICE, adopt=True)
ctx.upload_transforms(xf.data_ptr(), count=n_inst, mem=mc.MEM_DEVICE)
off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
ctx.set_instance_ranges(off, np.full(n_inst, per, dtype="<u4"))
g = host.create_globals(eye=(51.0, 54.4, 40.8), fov_deg=45.0, width=3840, height=2160, flip_y=True)
for cold in (False, False, False, True, True, True):
    if cold:
        junk.add_(1.0)
        torch.cuda.synchronize()
    ctx.cull(g, tests=3, list_capacity=1000, precull=False)
    ctx.synchronize()
print("done")
