#!/usr/bin/env python3
"""Benchmark of the GPU-driven culling path: meshlet-instances culled + compacted per second.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (per GPU, weak scaling): the synthetic 100 M meshlet-instance scene — 781 250 instances x
128 unique meshlets each (3.2 GB of SoA bounds resident in HBM), frustum + backface-cone tests,
ordered compaction into one visible list + one VkDrawMeshTasksIndirectCommandNV per instance.  One
step = one pass of the hot path (cull mask -> compact -> emit commands) over that scene; with N > 1
every rank culls its own shard and the per-rank visible counts and per-instance commands are
all-gathered over RCCL into one global draw list description (firstTask rebased by the gathered
counts on the device, no host round trip).

Rank 0 prints ONE JSON line; `roofline` is measured live with HIP events around the kernels of each
step on the stream they run on, `cpu_baseline` times the CPU oracle (a port of the reference shader,
the reference has no CPU culling path) on this host's cores.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def make_scene(device, n_inst, per, seed):
    """Synthetic scene on the device: unique meshlets per instance, random rigid transforms."""
    gen = torch.Generator(device=device).manual_seed(seed)
    n = n_inst * per
    cone = torch.empty((n, 4), device=device, dtype=torch.float32)
    chunk = 1 << 24
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        v = torch.randn((b - a, 3), device=device, generator=gen)
        cone[a:b, :3] = v / v.norm(dim=1, keepdim=True)
        cw = torch.rand(b - a, device=device, generator=gen)
        cw[torch.rand(b - a, device=device, generator=gen) < 0.1] = 1.0  # never cone-culled
        cone[a:b, 3] = cw
    sphere = torch.empty((n, 4), device=device, dtype=torch.float32)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        sphere[a:b, :3] = torch.rand((b - a, 3), device=device, generator=gen)
        sphere[a:b, 3] = 0.01 + 0.04 * torch.rand(b - a, device=device, generator=gen)
    q = torch.randn((n_inst, 4), device=device, generator=gen)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    m = torch.zeros((n_inst, 4, 4), device=device, dtype=torch.float32)  # [c][r]
    m[:, 0, 0] = 1 - 2 * (y * y + z * z); m[:, 1, 0] = 2 * (x * y - z * w); m[:, 2, 0] = 2 * (x * z + y * w)
    m[:, 0, 1] = 2 * (x * y + z * w); m[:, 1, 1] = 1 - 2 * (x * x + z * z); m[:, 2, 1] = 2 * (y * z - x * w)
    m[:, 0, 2] = 2 * (x * z - y * w); m[:, 1, 2] = 2 * (y * z + x * w); m[:, 2, 2] = 1 - 2 * (x * x + y * y)
    m[:, 3, :3] = (torch.rand((n_inst, 3), device=device, generator=gen) * 2 - 1) * 50.0
    m[:, 3, 3] = 1.0
    return cone, sphere, m.reshape(n_inst, 16).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--instances", type=int, default=781250)
    ap.add_argument("--meshlets-per-instance", type=int, default=128)
    ap.add_argument("--compaction", type=int, default=0, help="0 ordered, 1 wave atomic, 2 block atomic")
    ap.add_argument("--hiz", action="store_true", help="also run the Hi-Z test against a 4K pyramid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank gather path even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist

        if "MASTER_ADDR" not in os.environ:  # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)

    mc = importlib.import_module("vulkan-renderer_amd")  # raises if libmcull.so is missing
    host = importlib.import_module("vulkan-renderer_amd.host")
    mdist = importlib.import_module("vulkan-renderer_amd.dist")
    ctx = mc.Context(local_rank)

    n_inst, per = args.instances, args.meshlets_per_instance
    n_pairs = n_inst * per
    cone, sphere, xf = make_scene(device, n_inst, per, seed=42 + rank)
    off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
    cnt = np.full(n_inst, per, dtype="<u4")
    ctx.upload_bounds_soa(cone.data_ptr(), sphere.data_ptr(), count=n_pairs, mem=mc.MEM_DEVICE, adopt=True)
    ctx.upload_transforms(xf.data_ptr(), count=n_inst, mem=mc.MEM_DEVICE)
    ctx.set_instance_ranges(off, cnt)
    g = host.create_globals(eye=(51.0, 54.4, 40.8), fov_deg=45.0, width=3840, height=2160, flip_y=True)
    tests = mc.TEST_FRUSTUM | mc.TEST_CONE
    if args.hiz:
        depth = torch.from_numpy(host.synthetic_depth(3840, 2160, 64, 1234)).to(device)
        ctx.hiz_build(depth.data_ptr(), view=0, width=3840, height=2160, mem=mc.MEM_DEVICE)
        tests |= mc.TEST_HIZ

    # caller-owned outputs (torch owns what the collectives touch); commands and counts are double
    # buffered so that the gather of step k overlaps the cull of step k+1
    lst = torch.empty((n_pairs, 2), device=device, dtype=torch.int32)
    cmds = [torch.zeros((n_inst, 2), device=device, dtype=torch.int32) for _ in range(2)]
    total = [torch.zeros(1, device=device, dtype=torch.int64) for _ in range(2)]
    kernel_ms = []
    gathered = [None, None]
    if multi:
        comm = torch.cuda.Stream(device=device)
        ev_cull = [torch.cuda.Event() for _ in range(2)]
        ev_comm = [torch.cuda.Event() for _ in range(2)]
        main = torch.cuda.current_stream(device)

    def step(k, record=False):
        b = k & 1
        if multi:
            main.wait_event(ev_comm[b])  # the gather that last read buffer b has finished
        ctx.cull(g, tests=tests, compaction=args.compaction, list_ptr=lst.data_ptr(), list_capacity=n_pairs,
                 commands_ptr=cmds[b].data_ptr(), total_ptr=total[b].data_ptr())
        if multi:
            # per-rank counts -> every rank's list offset -> rebased per-instance commands, gathered
            # into one global draw-list description; runs on a side stream beside the next cull
            ev_cull[b].record(main)
            with torch.cuda.stream(comm):
                comm.wait_event(ev_cull[b])

                def rebase(totals_all, rk, cb=cmds[b]):
                    cb[:, 1] += totals_all[:rk].sum().to(torch.int32)

                gathered[b] = mdist.gather_draw_list(total[b], cmds[b], rebase)
                ev_comm[b].record(comm)
        if record:
            kernel_ms.append(ctx.last_kernel_ms())

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events on the stream the kernels are launched on (the context's stream = the null stream),
    # bracketing exactly the timed region
    hip = ctypes.CDLL("libamdhip64.so.7")
    ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipEventCreate(ctypes.byref(ev0)) == 0 and hip.hipEventCreate(ctypes.byref(ev1)) == 0

    for k in range(args.warmup):
        step(k)
    fence()
    t0 = time.perf_counter()
    assert hip.hipEventRecord(ev0, None) == 0
    for k in range(args.steps):
        step(k)
    assert hip.hipEventRecord(ev1, None) == 0
    fence()
    elapsed = time.perf_counter() - t0
    dev_ms = ctypes.c_float()
    assert hip.hipEventElapsedTime(ctypes.byref(dev_ms), ev0, ev1) == 0
    region_ms_per_step = dev_ms.value / args.steps
    if multi:
        te = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    # separate, untimed steps for the per-call device time (the event read-back synchronises)
    ctx.enable_timing(True)
    for k in range(min(args.steps, 10)):
        step(k, record=True)
    fence()
    visible = int(total[0].item())
    if multi:  # the gathered description must be consistent on every rank
        totals_all, cmds_all = gathered[0]
        assert totals_all.numel() == world and cmds_all.shape[0] == world * n_inst
        mine = cmds_all[rank * n_inst:(rank + 1) * n_inst].to(torch.int64)
        assert int(mine[:, 0].sum()) == int(totals_all[rank]) and int(mine[0, 1]) == int(totals_all[:rank].sum())
    ms_per_step = elapsed / args.steps * 1e3
    value = n_pairs * world / (elapsed / args.steps) / 1e9

    if rank == 0:
        k_ms = region_ms_per_step if world == 1 else float(np.mean(kernel_ms))
        n_hz = 0
        algo = 32.0 * n_pairs + 64.0 * n_inst + 8.0 * n_inst + 8.0 * visible + 4.0 * n_hz
        achieved = algo / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args, n_inst, per)
        out = {
            "metric": "meshlet-instances culled+compacted/sec", "value": round(value, 3), "unit": "G/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "config 4: synthetic scene, %d instances x %d unique meshlets = %d meshlet-instances per GPU, "
                            "frustum+cone%s, %s compaction" % (n_inst, per, n_pairs, "+hiz(4K)" if args.hiz else "",
                                                               ["ordered", "wave-atomic", "block-atomic"][args.compaction]),
                "meshlet_instances_per_gpu": n_pairs, "visible_fraction": round(visible / n_pairs, 4),
                "parallelism": ("instances sharded x%d, RCCL all-gather of counts + rebased commands overlapped with the next cull" % world)
                if multi else "single GPU",
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "cull_mask_kernel + compact_kernel + emit_commands_kernel (one step)",
                "kernel_ms": round(k_ms, 4), "kernel_ms_single_call": round(float(np.mean(kernel_ms)), 4),
                "algorithmic_bytes": int(algo),
                "bytes_per_pair": round(algo / n_pairs, 2),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cone, sphere, xf, off, cnt, g, tests, n_inst, per)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def pmc_traffic(args, n_inst, per):
    """HBM bytes per step from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE
    doubled as the gfx950 correction prescribes, WRITE_SIZE as is; separate passes), or None when the
    workload differs from the profiled one.  Counters cannot be collected from inside this process."""
    path = os.path.join(ROOT, "profiles", "r01_bench_config4_summary.json")
    if (n_inst, per, args.compaction, args.hiz) != (781250, 128, 0, False) or not os.path.exists(path):
        return None, None
    with open(path) as f:
        summ = json.load(f)
    total = 0
    for name, row in summ.items():
        if any(k in name for k in ("cull_mask_kernel", "compact_kernel", "emit_commands_kernel")):
            total += row.get("fetch_bytes_corrected", 0) + row.get("write_bytes_corrected", 0)
    return (total or None), "profiles/r01_bench_config4_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"


def cpu_baseline(cone, sphere, xf, off, cnt, g, tests, n_inst, per):
    """The CPU oracle (a port of shaders/shader.task, test infrastructure) timed on this host."""
    import oracle as O

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    threads = max(1, min(threads, O.max_threads()))
    sample = min(n_inst, 262144)  # 33.5 M pairs, ~1 GB of bounds on the host
    c = cone[: sample * per].cpu().numpy()
    s = sphere[: sample * per].cpu().numpy()
    t = np.zeros(sample, dtype=O.TRANSFORM)
    t["model"] = xf[:sample].cpu().numpy()
    view = O.View(g)
    rng = (off[:sample], cnt[:sample])
    O.cull_scene(view, t[:1024], cone_axis_w=c, centre_radius=s, ranges=(off[:1024], cnt[:1024]), tests=tests, threads=threads)
    reps, t_used = 0, 0.0
    while t_used < 10.0 and reps < 400:
        t0 = time.perf_counter()
        r = O.cull_scene(view, t, cone_axis_w=c, centre_radius=s, ranges=rng, tests=tests, threads=threads)
        t_used += time.perf_counter() - t0
        reps += 1
    return {"value": round(sample * per * reps / t_used / 1e9, 4), "unit": "G/s", "cores": threads, "kind": "port",
            "sample": "first %d instances (%d meshlet-instances) of the same scene, %d passes, OpenMP over instances"
                      % (sample, sample * per, reps)}


if __name__ == "__main__":
    main()
