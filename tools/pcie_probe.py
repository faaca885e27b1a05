"""Measures the PCIe-inclusive rate of the cull path: the bounds of a 10 M-pair scene re-uploaded from
(pageable) host memory before every cull.  Not how the path is meant to be used - scene arrays stay
resident - but the figure belongs next to the HBM-resident one.  Run on a GPU box: python tools/pcie_probe.py"""
import importlib, time, numpy as np, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
mc = importlib.import_module("vulkan-renderer_amd")
H = importlib.import_module("vulkan-renderer_amd.host")
n_inst, per = 78125, 128
cone, sphere = H.synthetic_bounds(n_inst * per, seed=1)
t = H.synthetic_transforms(n_inst, seed=2)
off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4"); cnt = np.full(n_inst, per, "<u4")
g = H.create_globals(eye=(51.0, 54.4, 40.8), width=3840, height=2160)
with mc.Context(0) as ctx:
    ctx.upload_transforms(t); ctx.upload_bounds_soa(cone, sphere); ctx.set_instance_ranges(off, cnt)
    ctx.cull(g, tests=3); ctx.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.upload_bounds_soa(cone, sphere)
        ctx.cull(g, tests=3)
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    nbytes = cone.nbytes + sphere.nbytes
    print("pairs %d, bounds %.0f MB from pageable host memory: upload+cull %.2f ms = %.2f G pairs/s, %.1f GB/s over the link"
          % (n_inst * per, nbytes / 1e6, best * 1e3, n_inst * per / best / 1e9, nbytes / best / 1e9))
