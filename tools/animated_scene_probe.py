"""A frame that uploads new matrices for the same instances before it culls (an animated scene), against the cull of a
static scene: 100 M pairs, matrices already in device memory.
    python tools/animated_scene_probe.py"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mc = importlib.import_module("vulkan-renderer_amd")
host = importlib.import_module("vulkan-renderer_amd.host")


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n_inst, per = 781250, 128
    ctx = mc.Context(0)
    axis, centre = host.synthetic_bounds(n_inst * per, seed=42)
    ctx.upload_bounds_soa(axis, centre)
    t = host.synthetic_transforms(n_inst, seed=43)
    t_dev = torch.from_numpy(t.view("u1").reshape(n_inst, 64).copy()).to(dev)
    ctx.upload_transforms(t)
    ctx.set_instance_ranges(np.arange(n_inst, dtype=np.uint32) * per, np.full(n_inst, per, dtype=np.uint32))
    g = host.create_globals(eye=(51.0, 54.4, 40.8), width=3840, height=2160)
    torch.cuda.synchronize()

    def run(animated, n=100):
        for _ in range(20):
            if animated:
                ctx.upload_transforms(t_dev.data_ptr(), count=n_inst, mem=mc.MEM_DEVICE)
            ctx.cull(g, tests=3)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if animated:
                ctx.upload_transforms(t_dev.data_ptr(), count=n_inst, mem=mc.MEM_DEVICE)
            ctx.cull(g, tests=3)
        ctx.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    print("static %.4f ms per frame, with the matrices uploaded every frame %.4f ms" % (run(False), run(True)))


if __name__ == "__main__":
    main()
