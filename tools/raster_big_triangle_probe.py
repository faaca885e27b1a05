"""How long does the rasteriser take on its worst input, one screen-filling quad (two triangles walked by a single
wave)?  Prints the time per image size."""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
mc = importlib.import_module("vulkan-renderer_amd")
verts = np.zeros(4, dtype=mc.VERTEX)
verts["pos"][:, :3] = [[-50, -50, 0], [50, -50, 0], [50, 50, 0], [-50, 50, 0]]
meshlets = np.zeros(1, dtype=mc.MESHLET)
meshlets[0]["vertexCount"], meshlets[0]["triangleCount"] = 4, 4
local = np.array([0, 1, 2, 0, 2, 3, 0, 2, 1, 0, 3, 2], dtype="u1")  # both windings: one pair faces the camera
data = np.concatenate([np.arange(4, dtype="<u4"), local.view("<u4")])
t = np.zeros(1, dtype=mc.TRANSFORM)
t["model"][0] = np.eye(4, dtype="<f4").reshape(-1)
with mc.Context(0) as ctx:
    ctx.upload_meshlets(meshlets)
    ctx.upload_geometry(verts, data)
    ctx.compute_bounds()
    ctx.upload_transforms(t)
    for w, h in ((800, 600), (3840, 2160), (8192, 8192)):
        g = mc.create_globals(w, h, 1, eye=(0.0, 0.0, 5.0), up=(0.0, 1.0, 0.0))
        ctx.cull(g, tests=0)
        ctx.depth_rasterize(g, w, h)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.depth_rasterize(g, w, h)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        d = ctx.depth_download()
        print("%dx%d: %.3f ms per image, %.1f %% of the pixels covered" % (w, h, ms, 100.0 * (d > 0).mean()), flush=True)
