"""Host microseconds per enqueue of the pieces of one sharded step (small shard: 97 656 instances x 128 meshlets),
measured on one GPU with a communicator of world size 1:
    python tools/host_costs.py
Each piece is enqueued 300 times back to back (the queue is drained in between, outside the timing)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

mc = importlib.import_module("vulkan-renderer_amd")
host = importlib.import_module("vulkan-renderer_amd.host")
mdist = importlib.import_module("vulkan-renderer_amd.dist")


def per_call(fn, n=300):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize()
    return dt


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n_inst, per = 97656, 128
    stream = torch.cuda.Stream(device=dev)
    side = torch.cuda.Stream(device=dev)
    ctx = mc.Context(0, stream=stream.cuda_stream)
    n_m = n_inst * per
    axis, centre = host.synthetic_bounds(n_m, seed=42)
    ctx.upload_bounds_soa(axis, centre)
    ctx.upload_transforms(host.synthetic_transforms(n_inst, seed=43))
    ctx.set_instance_ranges(np.arange(n_inst, dtype=np.uint32) * per, np.full(n_inst, per, dtype=np.uint32))
    g = host.create_globals(width=1920, height=1080)
    cmds = torch.zeros((n_inst + 1, 2), device=dev, dtype=torch.int32)
    lst = torch.empty((n_m, 2), device=dev, dtype=torch.int32)
    kw = dict(tests=3, list_ptr=lst.data_ptr(), list_capacity=n_m, commands_ptr=cmds.data_ptr(), total_ptr=cmds[n_inst:].data_ptr())
    for _ in range(5):
        ctx.cull(g, **kw)
    print("ctx.cull (host globals)            %6.1f us" % per_call(lambda: ctx.cull(g, **kw)))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        ctx.cull(g, **kw)
    with torch.cuda.stream(stream):
        print("replay of the captured cull        %6.1f us" % per_call(graph.replay))
    ev = torch.cuda.Event()
    print("event record + wait on side stream %6.1f us" % per_call(lambda: (ev.record(stream), side.wait_event(ev))))
    import ctypes

    hip = None
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            hip = ctypes.CDLL(line.split()[-1])
            break
    e = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), 2) == 0  # hipEventDisableTiming
    s0, s1 = ctypes.c_void_p(stream.cuda_stream), ctypes.c_void_p(side.cuda_stream)
    print("raw hipEventRecord                 %6.1f us" % per_call(lambda: hip.hipEventRecord(e, s0)))
    hip.hipEventRecord(e, s0)
    print("raw hipStreamWaitEvent             %6.1f us" % per_call(lambda: hip.hipStreamWaitEvent(s1, e, 0)))
    print("raw record + wait                  %6.1f us" % per_call(lambda: (hip.hipEventRecord(e, s0), hip.hipStreamWaitEvent(s1, e, 0))))
    print("torch record only                  %6.1f us" % per_call(lambda: ev.record(stream)))
    print("torch wait only                    %6.1f us" % per_call(lambda: side.wait_event(ev)))
    comm = mc.RcclComm(mc.rccl_unique_id(), 1, 0, 0)
    gathered = torch.empty((n_inst + 1, 2), device=dev, dtype=torch.int32)
    out = torch.empty((2, 1), device=dev, dtype=torch.int64)
    print("gather_draw_list_native            %6.1f us" % per_call(lambda: mdist.gather_draw_list_native(comm, side, cmds, gathered, out)))
    a = (comm.handle, side.cuda_stream, cmds.data_ptr(), gathered.data_ptr(), 1, n_inst, out[0].data_ptr(), out[1].data_ptr())
    print("mcull_allgather_commands alone     %6.1f us" % per_call(lambda: mc.lib.mcull_allgather_commands(*a)))
    r = (side.cuda_stream, gathered.data_ptr(), 1, n_inst, out[0].data_ptr(), out[1].data_ptr())
    print("mcull_rebase_gathered_commands     %6.1f us" % per_call(lambda: mc.lib.mcull_rebase_gathered_commands(*r)))
    comm.close()


if __name__ == "__main__":
    main()
