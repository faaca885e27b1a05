"""Host cost of one mcull_group_cull (one thread driving n shards) on one GPU: the reference scene split n ways.
The shards share the GPU here, so the figure is the per-frame cost of the control flow (launches, copies,
event waits), not a scaling number."""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
mc = importlib.import_module("vulkan-renderer_amd")
d = np.load("tests/golden/kitten_scene.npz")
meshlets = d["meshlets"].view(mc.MESHLET).reshape(-1)
t = mc.reference_transforms(1000)
g = mc.create_globals(800, 600, len(meshlets))
for n in (1, 2, 4, 8):
    with mc.Group([0] * n) as grp:
        for r, s in enumerate(grp.shards):
            s.upload_meshlets(meshlets)
            s.upload_transforms(t[len(t) * r // n:len(t) * (r + 1) // n])
        for _ in range(20):
            grp.cull(g, tests=3, cmd_stride=28, cmd_offset=20)
        grp.synchronize()
        t0 = time.perf_counter()
        k = 200
        for _ in range(k):
            grp.cull(g, tests=3, cmd_stride=28, cmd_offset=20)
        t1 = time.perf_counter()
        grp.synchronize()
        t2 = time.perf_counter()
        print("shards %d: %.1f us per frame issued, %.1f us per frame completed" % (n, (t1 - t0) / k * 1e6, (t2 - t0) / k * 1e6), flush=True)
