"""What two culls with different tests cost when they alternate on one context, as the two phases of a frame do
(frustum + cone, then all three tests), against the same culls run in blocks of their own: the work-list estimate of one
phase must not size the other's mask pass.  100 M pairs, one GPU:
    python tools/alternating_culls_probe.py"""
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
    ctx.upload_transforms(host.synthetic_transforms(n_inst, seed=43))
    ctx.set_instance_ranges(np.arange(n_inst, dtype=np.uint32) * per, np.full(n_inst, per, dtype=np.uint32))
    w, h = 3840, 2160
    g = host.create_globals(eye=(51.0, 54.4, 40.8), width=w, height=h)  # the headline's camera: half the pairs pass frustum + cone, 4 % all tests
    ctx.hiz_build(host.synthetic_depth(w, h, 64, 1234), view=0)

    def run(seq, n=60):
        for t in seq * 6:
            ctx.cull(g, tests=t, hiz_view=0)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for t in seq:
                ctx.cull(g, tests=t, hiz_view=0)
        ctx.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    a, b = run([3]), run([7])
    print("frustum+cone alone %.4f ms, all tests alone %.4f ms, sum %.4f ms" % (a, b, a + b))
    print("alternating         %.4f ms per pair" % run([3, 7]))


if __name__ == "__main__":
    main()
