"""Rasteriser time when the camera is close to the mesh (triangles of hundreds to thousands of pixels): a few kitten
instances filling a 3840x2160 image, and the reference scene at 800x600 for comparison."""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
mc = importlib.import_module("vulkan-renderer_amd")
d = np.load("tests/golden/kitten_scene.npz")
ml, md, vt = d["meshlets"].view(mc.MESHLET).reshape(-1), d["meshlet_data"], d["vertices"].view(mc.VERTEX).reshape(-1)
centre = ml["coneApex"][:, :3].mean(0)
span = float(np.abs(vt["pos"][:, :3] - centre).max())
with mc.Context(0) as ctx:
    ctx.upload_meshlets(ml)
    ctx.upload_geometry(vt, md)
    for name, n, (w, h), dist in (("close-up, 1 instance", 1, (3840, 2160), 1.6), ("close-up, 8 instances", 8, (3840, 2160), 2.5),
                                   ("very close, 1 instance", 1, (3840, 2160), 0.7), ("reference scene", 1000, (800, 600), None)):
        if dist is None:
            t = mc.reference_transforms(n)
            g = mc.create_globals(w, h, len(ml))
        else:
            t = np.zeros(n, dtype=mc.TRANSFORM)
            for i in range(n):
                m = np.eye(4, dtype="<f4"); m[3, :3] = [0.0, (i - (n - 1) / 2) * span * 1.2, 0.0]; t["model"][i] = m.reshape(-1)
            eye = centre + np.array([dist * span, 0.0, 0.2 * span])
            g = mc.create_globals(w, h, len(ml), eye=eye.astype("<f4"), centre=centre.astype("<f4"))
        ctx.upload_transforms(t)
        ctx.cull(g, tests=3)
        ctx.depth_rasterize(g, w, h)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.depth_rasterize(g, w, h)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print("%-24s %4dx%-4d %8.3f ms  coverage %.2f" % (name, w, h, ms, float((ctx.depth_download() > 0).mean())), flush=True)
