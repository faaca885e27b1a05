#!/usr/bin/env python3
"""Benchmark of the GPU-driven culling path: meshlet-instances culled + compacted per second.

    python bench.py --gpus N --steps K --warmup W [--workload config4|config3|config5|hiz|bounds]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (per GPU, weak scaling): the synthetic 100 M meshlet-instance scene — 781 250
instances x 128 unique meshlets each (3.2 GB of SoA bounds resident in HBM), frustum + backface-cone
tests, ordered compaction into one visible list + one VkDrawMeshTasksIndirectCommandNV per instance.
One step = one pass of the hot path (cull mask -> compact -> emit commands) over that scene; with
N > 1 every rank culls its own shard and the per-rank visible counts and per-instance commands are
all-gathered over RCCL into one global draw list description (firstTask rebased by the gathered
counts on the device, no host round trip).

Other workloads (each prints its own single JSON line):
  config1  the reference's own scene (1000 kitten instances, 800x600) as a two-phase frame
  config3  1 M instances of one 13 916-meshlet mesh (13.9 G meshlet-instances) with Hi-Z against a
           3840x2160 pyramid; shared bounds, so the bound is the per-pair arithmetic and the list
  config5  the default scene culled against 8 views, each with its own 8192^2 pyramid rebuilt in the
           step (8 x (pyramid build + cull))
  hiz      the 8192^2 pyramid build alone (9.33 algorithmic bytes per pixel)
  bounds   cone/sphere computation for 64 copies of the kitten fixture's meshlets
  depth    the depth pre-pass stand-in: visible meshlets of 4 096 kitten instances rasterised at 4K
  meshbuild  vertex dedup + meshlet building + bounds of an 871 200-triangle mesh on the GPU

Rank 0 prints ONE JSON line; `roofline` is measured live with HIP events around the kernels of the
timed region on the stream they run on, `cpu_baseline` times the CPU oracle (a port of the
reference shader / CPU code; the reference has no CPU culling path) on this host's cores.
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


SCENE_CHUNK = 1250  # instances per generator chunk (781 250 = 625 chunks)


def _chunk_gen(device, seed, chunk):
    """The generator of one chunk of a synthetic scene: seeded by (scene seed, chunk index) alone, so any
    rank can generate any contiguous range of instances locally and the shards of a scene concatenate to
    exactly the scene a single rank generates."""
    return torch.Generator(device=device).manual_seed(seed * 1_000_003 + chunk)


def make_transforms_range(device, first, n_inst, extent, seed):
    """Random rigid transforms of instances [first, first + n_inst) of the scene `seed`."""
    out = torch.empty((n_inst, 16), device=device, dtype=torch.float32)
    c0, c1 = first // SCENE_CHUNK, (first + n_inst + SCENE_CHUNK - 1) // SCENE_CHUNK
    for ch in range(c0, c1):
        gen = _chunk_gen(device, seed ^ 0x5bd1e995, ch)
        m = make_transforms(device, SCENE_CHUNK, extent, gen)
        a, b = max(first, ch * SCENE_CHUNK), min(first + n_inst, (ch + 1) * SCENE_CHUNK)
        out[a - first:b - first] = m[a - ch * SCENE_CHUNK:b - ch * SCENE_CHUNK]
    return out


def make_scene_range(device, first, n_inst, per, seed):
    """Instances [first, first + n_inst) of the synthetic scene `seed`: `per` unique meshlets per instance
    (unit-cube centres, radius U(0.01, 0.05), axis uniform on the sphere, coneW U(0,1) with 10 % set to 1),
    random rigid transforms over +-50.  Returns (cone_axis_w, centre_radius, transforms) on the device."""
    n = n_inst * per
    cone = torch.empty((n, 4), device=device, dtype=torch.float32)
    sphere = torch.empty((n, 4), device=device, dtype=torch.float32)
    c0, c1 = first // SCENE_CHUNK, (first + n_inst + SCENE_CHUNK - 1) // SCENE_CHUNK
    rows = SCENE_CHUNK * per
    for ch in range(c0, c1):
        gen = _chunk_gen(device, seed, ch)
        v = torch.randn((rows, 3), device=device, generator=gen)
        cw = torch.rand(rows, device=device, generator=gen)
        cw[torch.rand(rows, device=device, generator=gen) < 0.1] = 1.0  # never cone-culled
        ctr = torch.rand((rows, 3), device=device, generator=gen)
        rad = 0.01 + 0.04 * torch.rand(rows, device=device, generator=gen)
        a, b = max(first, ch * SCENE_CHUNK), min(first + n_inst, (ch + 1) * SCENE_CHUNK)
        src = slice((a - ch * SCENE_CHUNK) * per, (b - ch * SCENE_CHUNK) * per)
        dst = slice((a - first) * per, (b - first) * per)
        cone[dst, :3] = (v / v.norm(dim=1, keepdim=True))[src]
        cone[dst, 3] = cw[src]
        sphere[dst, :3] = ctr[src]
        sphere[dst, 3] = rad[src]
    return cone, sphere, make_transforms_range(device, first, n_inst, 50.0, seed)


def make_transforms(device, n_inst, extent, gen):
    q = torch.randn((n_inst, 4), device=device, generator=gen)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    m = torch.zeros((n_inst, 4, 4), device=device, dtype=torch.float32)  # [c][r]
    m[:, 0, 0] = 1 - 2 * (y * y + z * z); m[:, 1, 0] = 2 * (x * y - z * w); m[:, 2, 0] = 2 * (x * z + y * w)
    m[:, 0, 1] = 2 * (x * y + z * w); m[:, 1, 1] = 1 - 2 * (x * x + z * z); m[:, 2, 1] = 2 * (y * z - x * w)
    m[:, 0, 2] = 2 * (x * z - y * w); m[:, 1, 2] = 2 * (y * z + x * w); m[:, 2, 2] = 1 - 2 * (x * x + y * y)
    m[:, 3, :3] = (torch.rand((n_inst, 3), device=device, generator=gen) * 2 - 1) * extent
    m[:, 3, 3] = 1.0
    return m.reshape(n_inst, 16).contiguous()


def shard_range(n_total, world, rank):
    """Contiguous instance range [first, first + n) of `rank` when n_total instances are split over `world` ranks."""
    a, b = (n_total * rank) // world, (n_total * (rank + 1)) // world
    return a, b - a


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes, one rank each, with the
    rendezvous variables torch.distributed.run would set.  This process never touches the GPU (it only
    starts the children, relays rank 0's stdout - the one JSON line - and returns the first non-zero exit
    code); the other ranks' stdout goes to stderr."""
    import subprocess
    import threading

    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))

    def relay():
        for raw in procs[0].stdout:
            sys.stdout.write(raw.decode(errors="replace"))
            sys.stdout.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    code = 0
    live = set(range(n))
    while live and code == 0:
        time.sleep(0.1)
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0:
                    code = rc
                    print("bench.py: rank %d exited with status %d" % (r, rc), file=sys.stderr)
    if code != 0:  # stop exactly the children started here
        for r in live:
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    th.join(timeout=10)
    return code if code >= 0 else 128 - code


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block (banners of the collective libraries)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def loaded_hip_runtime():
    """The HIP runtime this process already uses (torch's), found through the process's own mappings."""
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    return ctypes.CDLL("libamdhip64.so")


class Env:
    """Process-wide state shared by the workloads: ranks, the library, the timing primitives."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:  # main() starts the ranks itself when no launcher did
            raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, self.world))
        if args.rehearse_shared_gpu:
            # control-flow rehearsal of the multi-rank step on a one-GPU box: every rank uses GPU 0 and the
            # collectives run over gloo on host copies (RCCL refuses two ranks on one device); its numbers
            # mean nothing
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.multi = self.world > 1 or args.force_dist
        self.dist = None
        if self.multi:
            import torch.distributed as dist

            if "MASTER_ADDR" not in os.environ:  # --force-dist without a launcher
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
            # RCCL prints a version banner on stdout when its communicator comes up; stdout carries the
            # one JSON line only, so the banner is sent to stderr (communicator creation forced here).
            with stdout_to_stderr():
                if args.rehearse_shared_gpu:
                    dist.init_process_group("gloo")
                else:
                    dist.init_process_group("nccl", device_id=self.device)
                dist.barrier()
                torch.cuda.synchronize()
            self.dist = dist
        self.mc = importlib.import_module("vulkan-renderer_amd")  # raises if libmcull.so is missing
        self.host = importlib.import_module("vulkan-renderer_amd.host")
        self.mdist = importlib.import_module("vulkan-renderer_amd.dist")
        # The library runs on a stream of its own, not on the null stream: work on the null stream waits for
        # everything enqueued earlier on the other streams of the process, which would serialise each
        # cull behind the previous step's collective instead of overlapping them.
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx = self.mc.Context(self.local_rank, stream=self.stream.cuda_stream)
        # the step's collective goes through the library (mcull_allgather_commands on a communicator of its own, made
        # from an id that torch.distributed carries to the ranks) unless --torch-collective asks for all_gather_into_tensor
        self.first_steps_ms = None
        self.gather_mode = None
        self.native_comm = None
        self.collective = "torch.distributed all_gather_into_tensor + mcull_rebase_gathered_commands" if self.multi else None
        if self.multi and not args.rehearse_shared_gpu and not args.torch_collective:
            try:
                with stdout_to_stderr():
                    self.native_comm = self.mdist.create_native_comm(self.local_rank)
                self.collective = "mcull_allgather_commands (ncclAllGather + rebase enqueued by the library on its own RCCL communicator)"
            except Exception as e:  # noqa: BLE001 - RCCL not found at run time, or the communicator refused: the torch path still works
                print("bench: native collective unavailable (%s); using torch.distributed" % e, file=sys.stderr)
        # HIP events on the stream the kernels are launched on, bracketing exactly the timed region
        self.hip = loaded_hip_runtime()
        self.ev0, self.ev1 = ctypes.c_void_p(), ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(self.ev0)) == 0 and self.hip.hipEventCreate(ctypes.byref(self.ev1)) == 0

    def fence(self):
        if self.multi:
            # the barrier runs on torch.distributed's communicator, the steps' gathers on the library's: drain the device
            # first, so that kernels of the two communicators are never in flight together
            torch.cuda.synchronize()
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, step, steps=None, warmup=None):  # noqa: C901
        """W untimed steps, then K steps between barrier+synchronize fences.  Returns (wall seconds,
        max over ranks; device milliseconds between HIP events on the kernels' stream)."""
        steps = self.args.steps if steps is None else steps
        warmup = self.args.warmup if warmup is None else warmup
        torch.cuda.synchronize()  # tensors prepared on torch's stream are complete before the library's stream reads them
        if self.first_steps_ms is None:
            # What a caller sees who starts from an idle GPU: the first 20 steps of the process's first timed leg, before
            # any ramp (clocks still rising, first-use allocations of the library inside); reported beside the sustained figure.
            self.fence()
            t_first = time.perf_counter()
            for k in range(20):
                step(k)
            self.fence()
            self.first_steps_ms = (time.perf_counter() - t_first) * 1e3 / 20
        if self.args.ramp_ms > 0:
            # After the idle set-up phase the GPU needs ~30 ms of continuous load before its clocks reach
            # their sustained state (measured: 0.66 ms per step over the first 20 steps, 0.60 ms from
            # about step 40 on).  Run the step untimed for ramp_ms before the W warm-up steps, so that
            # K timed steps measure sustained throughput whatever W and K are.
            t_ramp = time.perf_counter()
            k = 0
            while True:
                for _ in range(8):
                    step(k)
                    k += 1
                torch.cuda.synchronize()
                done = (time.perf_counter() - t_ramp) * 1e3 >= self.args.ramp_ms
                if self.multi:
                    # every step issues a collective, so all ranks must run the same number of ramp
                    # steps: they stop together, once the slowest rank has had its ramp time
                    flag = torch.tensor([1 if done else 0], device=self.device, dtype=torch.int32)
                    self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
                    done = bool(flag.item())
                if done:
                    break
        for k in range(warmup):
            step(k)
        self.fence()
        t0 = time.perf_counter()
        assert self.hip.hipEventRecord(self.ev0, ctypes.c_void_p(self.stream.cuda_stream)) == 0
        for k in range(steps):
            step(k)
        assert self.hip.hipEventRecord(self.ev1, ctypes.c_void_p(self.stream.cuda_stream)) == 0
        self.fence()
        elapsed = time.perf_counter() - t0
        dev_ms = ctypes.c_float()
        assert self.hip.hipEventElapsedTime(ctypes.byref(dev_ms), self.ev0, self.ev1) == 0
        if self.multi:
            te = torch.tensor([elapsed], device=self.device, dtype=torch.float64)
            self.dist.all_reduce(te, op=self.dist.ReduceOp.MAX)
            elapsed = float(te.item())
        return elapsed, dev_ms.value

    def line(self, out):
        if self.rank == 0:
            print(json.dumps(out), flush=True)

    def close(self):
        if self.native_comm is not None:
            torch.cuda.synchronize()
            self.native_comm.close()
        if self.multi:
            self.dist.barrier()
            self.dist.destroy_process_group()
        self.ctx.close()


def base_line(env, metric, unit, value, elapsed, steps, workload, extra_config=None):
    a = env.args
    cfg = {"workload": workload}
    cfg.update(extra_config or {})
    return {
        "metric": metric, "value": round(value, 3), "unit": unit, "n_gpus": env.world, "steps": steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "config": cfg, "ramp_ms": a.ramp_ms,
        "first_steps_ms": None if env.first_steps_ms is None else round(env.first_steps_ms, 4),
    }


def roofline(algo_bytes, kernel_ms, kernel_name, traffic=None, traffic_src=None, **extra):
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
         "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes": int(algo_bytes)}
    r.update(extra)
    return r


def measured_copy_gbs(device, mib=1024, reps=10):
    """The device-to-device copy rate of this GPU, measured now (read + write bytes per second of a 1 GiB
    hipMemcpy): the ceiling a plain copy reaches on this box, reported next to the 8 TB/s datasheet peak."""
    n = mib << 20
    src = torch.empty(n, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    src.fill_(1)
    for _ in range(3):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del src, dst
    return round(2.0 * n / (ms * 1e-3) / 1e9, 1)


def host_threads():
    import oracle as O

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    return max(1, min(threads, O.max_threads()))


def time_cpu(fn, budget_s, max_reps=400):
    reps, used = 0, 0.0
    while used < budget_s and reps < max_reps:
        t0 = time.perf_counter()
        fn()
        used += time.perf_counter() - t0
        reps += 1
    return reps, used


# ------------------------------------------------------------------------------------------------
# cull workloads (config 4 = default, config 3, config 5)

def cull_leg(env, workload, scaling, steps=None, warmup=None, gather_lists=False):
    """One measured leg of a cull workload.  scaling "weak": every rank culls a whole scene of its own
    (seed 42 + rank); "strong": ONE scene (seed 42) whose instances are split into `world` contiguous
    shards, rank r generating and culling instances [r*I/N, (r+1)*I/N) - configuration 4 as its statement words it (100 M pairs sharded over the GPUs).
    Returns a dict of measurements (same on every rank where it matters) plus the scene tensors."""
    args, mc, ctx, device = env.args, env.mc, env.ctx, env.device
    rank, world, multi, dist = env.rank, env.world, env.multi, env.dist
    steps = args.steps if steps is None else steps
    n_views = 1
    strong = scaling == "strong"
    if workload == "config3":
        n_scene, per = (args.instances or 1_000_000), (args.meshlets_per_instance or 13916)
        first, n_inst = shard_range(n_scene, world, rank) if strong else (0, n_scene)
        xf = make_transforms_range(device, first, n_inst, 70.0, 77 if strong else 77 + rank)
        ctx.set_instance_ranges(None, None)
        if args.c3_mesh == "proxy" and not args.meshlets_per_instance:
            # the stand-in for dragon.obj (871 200 triangles) through the product's own load-time pipeline: dedup,
            # reorder, meshlets, bounds; consecutive meshlets are neighbours on the surface, as in any real mesh
            pos, tris = env.host.proxy_mesh(660, 660, seed=12)
            cor = np.full((tris.size, 3), -1, dtype="<i4")
            cor[:, 0] = tris.reshape(-1)
            mesh = ctx.mesh_from_arrays(pos, cor).optimize()
            ctx.build_meshlets(mesh)  # leaves meshlets, geometry and bounds uploaded
            per = len(mesh.meshlets)
            ml = ctx.download_meshlets(per)
            c_np = np.ascontiguousarray(ml["cone"])
            s_np = np.ascontiguousarray(np.concatenate([ml["coneApex"][:, :3], ml["radius"][:, None]], axis=1))
            cone, sphere = torch.from_numpy(c_np).to(device), torch.from_numpy(s_np).to(device)
            torch.cuda.synchronize()
        else:
            c_np, s_np = env.host.synthetic_bounds(per, seed=5)  # one shared "mesh" of unrelated meshlets in a unit cube
            cone, sphere = torch.from_numpy(c_np).to(device), torch.from_numpy(s_np).to(device)
            torch.cuda.synchronize()
            ctx.upload_bounds_soa(cone.data_ptr(), sphere.data_ptr(), count=per, mem=mc.MEM_DEVICE, adopt=True)
        ctx.upload_transforms(xf.data_ptr(), count=n_inst, mem=mc.MEM_DEVICE)
        off = cnt = None
        g = env.host.create_globals(eye=(90.0, 85.0, 60.0), width=3840, height=2160, flip_y=False, meshlet_count=per)
        use_hiz, hiz_size = True, (3840, 2160)
        unique = per
    else:
        n_scene, per = (args.instances or 781250), (args.meshlets_per_instance or 128)
        first, n_inst = shard_range(n_scene, world, rank) if strong else (0, n_scene)
        cone, sphere, xf = make_scene_range(device, first, n_inst, per, seed=42 if strong else 42 + rank)
        if args.spatial_order:
            # experiment: the same instances stored cell by cell of a 16^3 grid over the scene (Morton order of the cells),
            # as a scene graph or a streaming system would store them, instead of in random order
            cell = ((xf[:, 12:15] + 50.0) * (16.0 / 100.0)).clamp(0, 15).to(torch.int64)
            key = torch.zeros(n_inst, dtype=torch.int64, device=device)
            for bit in range(4):
                for axis in range(3):
                    key |= ((cell[:, axis] >> bit) & 1) << (3 * bit + axis)
            xf = xf[torch.argsort(key, stable=True)].contiguous()
        torch.cuda.synchronize()
        off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
        cnt = np.full(n_inst, per, dtype="<u4")
        ctx.upload_bounds_soa(cone.data_ptr(), sphere.data_ptr(), count=n_inst * per, mem=mc.MEM_DEVICE,
                              adopt=not args.copy_bounds)
        ctx.upload_transforms(xf.data_ptr(), count=n_inst, mem=mc.MEM_DEVICE)
        ctx.set_instance_ranges(off, cnt)
        g = env.host.create_globals(eye=(51.0, 54.4, 40.8), fov_deg=args.fov, width=3840, height=2160, flip_y=not args.unflipped)
        use_hiz, hiz_size = args.hiz, (3840, 2160)
        unique = n_inst * per
        if workload == "config5":
            n_views, use_hiz, hiz_size = 8, True, (8192, 8192)
    n_pairs = n_inst * per                      # this rank's meshlet-instances
    job_pairs = n_scene * per * (1 if strong else world)  # the whole job's
    # every rank's command buffer has the rows of the largest shard, so that the all-gather is regular
    n_rows = shard_range(n_scene, world, 0)[1] if strong else n_inst
    if strong:
        n_rows = max(shard_range(n_scene, world, r)[1] for r in range(world))
    tests = mc.TEST_FRUSTUM | mc.TEST_CONE
    views = [g]
    def orbit():
        # a different Globals every step: the eye on a slow orbit (0.2 degrees per step, back and forth over 64 steps), so the
        # list an earlier cull reported is off by a few per cent and the mask pass's second launch has real work
        cams = []
        for k in range(64):
            a = np.deg2rad(0.2 * (k if k < 32 else 64 - k))
            ex, ey = 51.0 * np.cos(a) - 54.4 * np.sin(a), 51.0 * np.sin(a) + 54.4 * np.cos(a)
            cams.append(env.host.create_globals(eye=(ex, ey, 40.8), fov_deg=args.fov, width=3840, height=2160, flip_y=not args.unflipped))
        return cams

    cameras = orbit() if args.moving_camera and workload == "config4" else None
    depth = None
    if use_hiz:
        tests |= mc.TEST_HIZ
        w, h = hiz_size
        if workload == "config5":
            gen = torch.Generator(device=device).manual_seed(3)
            coarse = torch.rand((64, 64), device=device, generator=gen)
            d = torch.where(coarse > 0.6, coarse * 0.05, torch.zeros_like(coarse))
            depth = d.repeat_interleave(h // 64, 0).repeat_interleave(w // 64, 1).contiguous()
            views = []
            for v in range(8):
                ang = 2 * np.pi * v / 8
                views.append(env.host.create_globals(eye=(120 * np.cos(ang), 120 * np.sin(ang), 40.0 + 10 * v),
                                                     fov_deg=30.0 + 5 * v, width=w, height=h, flip_y=bool(v & 1)))
        else:
            depth = torch.from_numpy(env.host.synthetic_depth(w, h, 64, 1234)).to(device)
        torch.cuda.synchronize()
        for v in range(n_views):
            ctx.hiz_build(depth.data_ptr(), view=v, width=w, height=h, mem=mc.MEM_DEVICE)

    if args.tests is not None:  # experiments: a subset of the workload's tests (the pyramids are still built)
        tests &= args.tests
    # caller-owned outputs (torch owns what the collectives touch); commands and counts are double
    # buffered so that the gather of step k overlaps the cull of step k+1
    cap = n_pairs if workload != "config3" else min(n_pairs, 1 << 30)
    lst = torch.empty((max(cap, 1), 2), device=device, dtype=torch.int32)
    # one extra row per command buffer: the visible count (uint64) is written there, so that one
    # all-gather per step carries both the commands and the counts
    cmds_packed = [torch.zeros((n_rows + 1, 2), device=device, dtype=torch.int32) for _ in range(2)]
    cmds = [x[:n_inst] for x in cmds_packed]
    total = [x[n_rows:].view(torch.int64) for x in cmds_packed]
    # multi-view: the last view writes the buffers the collectives read; the others stay context-owned
    view_params = [[dict(tests=tests, hiz_view=v, precull=not args.no_precull, exact=args.exact) for v in range(n_views)] for _ in range(2)]
    for bb in range(2):
        view_params[bb][-1].update(list_ptr=lst.data_ptr(), list_capacity=cap, commands_ptr=cmds[bb].data_ptr(),
                                   total_ptr=total[bb].data_ptr())
    kernel_ms = []
    gathered = [None, None]
    if multi:
        gather_buf = [torch.empty((world * (n_rows + 1), 2), device=device, dtype=torch.int32) for _ in range(2)]
        gather_out = [torch.empty((2, world), device=device, dtype=torch.int64) for _ in range(2)]
        comm = torch.cuda.Stream(device=device)
        ev_cull = [torch.cuda.Event() for _ in range(2)]
        ev_comm = [torch.cuda.Event() for _ in range(2)]
        main = env.stream

    native = [env.native_comm]  # set to None below if its first result is not what torch.distributed's gather gives

    # shards whose cull is shorter than the host needs to enqueue two cross-stream edges (below 16 M pairs) gather on the
    # cull's own stream; larger ones overlap the gather with the next cull
    want_serial = args.serial_gather or (n_pairs * n_views < (16 << 20) and not args.overlap_gather)
    serial_gather = bool(multi and native[0] is not None and want_serial and not gather_lists)
    env.gather_mode = "on the cull's stream (serial)" if serial_gather else "on a side stream, overlapped with the next cull"

    host_s = [0.0, 0.0, 0]  # MCULL_BENCH_HOST_TIMES=1: host seconds spent enqueuing the culls / the gathers, steps

    def step(k, record=False, gather_lists=False):
        b = k & 1
        t0 = time.perf_counter()
        if multi and not (serial_gather and native[0] is not None):
            main.wait_event(ev_comm[b])  # the gather that last read buffer b has finished
        ms = 0.0
        if workload == "config5" and not args.separate_views:
            # the pyramid of every view is rebuilt from its depth each step, then ONE pass over the
            # bounds culls all views (each into its own context-owned list / commands / count)
            ctx.hiz_build_views([depth.data_ptr()] * n_views, width=hiz_size[0], height=hiz_size[1], mem=mc.MEM_DEVICE)
            if record:
                ms += ctx.last_kernel_ms()
            ctx.cull_views(views, view_params[b])
            if record:
                ms += ctx.last_kernel_ms()
        else:
            if cameras is not None:
                views[0] = cameras[k % len(cameras)]
            for v in range(n_views):
                if workload == "config5":
                    ctx.hiz_build(depth.data_ptr(), view=v, width=hiz_size[0], height=hiz_size[1], mem=mc.MEM_DEVICE)
                    if record:
                        ms += ctx.last_kernel_ms()
                if args.ctx_outputs:
                    ctx.cull(views[v], tests=tests, hiz_view=v, compaction=args.compaction, total_ptr=total[b].data_ptr(),
                             precull=not args.no_precull, exact=args.exact)
                else:
                    ctx.cull(views[v], tests=tests, hiz_view=v, compaction=args.compaction, list_ptr=lst.data_ptr(),
                             list_capacity=cap, commands_ptr=cmds[b].data_ptr(), total_ptr=total[b].data_ptr(),
                             precull=not args.no_precull, exact=args.exact)
                if record:
                    ms += ctx.last_kernel_ms()
        t1 = time.perf_counter()
        if multi and serial_gather and native[0] is not None:
            # small shards: the gather on the cull's own stream, behind it - no cross-stream edge to enqueue (two of them cost
            # the host more than the collective), at the price of not overlapping the next cull
            gathered[b] = env.mdist.gather_draw_list_native(native[0], main, cmds_packed[b], gather_buf[b], gather_out[b])
        elif multi:
            # per-rank counts -> every rank's list offset -> rebased per-instance commands, gathered
            # into one global draw-list description; runs on a side stream beside the next cull
            ev_cull[b].record(main)
            with torch.cuda.stream(comm):
                comm.wait_event(ev_cull[b])

                if args.rehearse_shared_gpu:
                    gathered[b] = env.mdist.gather_draw_list_packed(cmds_packed[b].cpu())
                elif native[0] is not None:
                    gathered[b] = env.mdist.gather_draw_list_native(native[0], comm, cmds_packed[b], gather_buf[b], gather_out[b])
                else:
                    gathered[b] = env.mdist.gather_draw_list_packed(cmds_packed[b], gathered=gather_buf[b], out=gather_out[b])
                if gather_lists:
                    env.mdist.gather_visible_lists(lst, gathered[b][0])
                ev_comm[b].record(comm)
        host_s[0] += t1 - t0
        host_s[1] += time.perf_counter() - t1
        host_s[2] += 1
        if record:
            kernel_ms.append(ms)

    if multi and native[0] is not None:
        # once, untimed: the library's collective against torch.distributed's on the same commands; a rank that sees a
        # difference takes every rank back to the torch path (and the line says which one ran)
        torch.cuda.synchronize()
        step(0)
        torch.cuda.synchronize()
        ref_totals, ref_cmds = env.mdist.gather_draw_list_packed(cmds_packed[0])
        same = torch.equal(ref_totals, gathered[0][0]) and torch.equal(ref_cmds, gathered[0][1])
        flag = torch.tensor([1 if same else 0], device=device, dtype=torch.int32)
        env.dist.all_reduce(flag, op=env.dist.ReduceOp.MIN)
        if not bool(flag.item()):
            print("bench: mcull_allgather_commands disagreed with torch.distributed on rank %d; using torch.distributed" % rank, file=sys.stderr)
            native[0] = None
            env.gather_mode = "on a side stream, overlapped with the next cull"
            env.collective = "torch.distributed all_gather_into_tensor + mcull_rebase_gathered_commands (the native collective's first result differed)"
        torch.cuda.synchronize()
    elapsed, dev_ms = env.timed(step, steps=steps, warmup=warmup)
    region_ms_per_step = dev_ms / steps
    if os.environ.get("MCULL_BENCH_HOST_TIMES") and rank == 0:
        print("host time per step: cull enqueue %.1f us, gather enqueue %.1f us (%d steps incl. warm-up)"
              % (host_s[0] / host_s[2] * 1e6, host_s[1] / host_s[2] * 1e6, host_s[2]), file=sys.stderr)
    moving_ms = None
    if workload == "config4" and cameras is None and not multi and not args.instances and not args.meshlets_per_instance:
        # the same step under a camera that moves (a short leg beside the headline figure, which repeats one camera)
        cameras = orbit()
        n_m = min(steps, 64)
        el_m, _ = env.timed(step, steps=n_m, warmup=2)
        moving_ms = el_m / n_m * 1e3
        cameras = None
        views[0] = g
    two_phase_ms = None
    if workload == "config4" and args.hiz and not multi and not args.instances and not args.meshlets_per_instance and args.tests is None:
        # the two phases of a frame taking turns on this context: frustum + cone, then all three tests (each keeps its own
        # list-length estimate and back-off; a short leg beside the figure of the all-tests cull alone)
        all_tests = tests

        def pair(k):
            nonlocal tests
            tests = all_tests & 3
            step(2 * k)
            tests = all_tests
            step(2 * k + 1)

        n_p = min(steps, 50)
        el_p, _ = env.timed(pair, steps=n_p, warmup=4)
        two_phase_ms = el_p / n_p * 1e3
        tests = all_tests
    # separate, untimed steps for the per-call device time (the event read-back synchronises)
    ctx.enable_timing(True)
    for k in range(min(steps, 10)):
        step(k, record=True)
    ctx.enable_timing(False)
    env.fence()
    visible = int(total[0].item())
    assert visible <= cap, "the visible list (%d entries) does not fit its buffer (%d): the timed steps wrote a truncated list" % (visible, cap)
    with_lists = None
    job_visible = visible
    if multi:  # the gathered description must be consistent on every rank
        totals_all, cmds_all = gathered[0]
        assert totals_all.numel() == world and tuple(cmds_all.shape) == (world, n_rows, 2)
        mine = cmds_all[rank].to(torch.int64)
        assert int(mine[:, 0].sum()) == int(totals_all[rank]) and (int(mine[0, 1]) & 0xffffffff) == (int(totals_all[:rank].sum()) & 0xffffffff)
        job_visible = int(totals_all.sum())
        if gather_lists:
            # the same step including the optional all-gather of the visible lists themselves
            # (padded to the largest shard; link-bound over xGMI), timed separately
            n2 = min(steps, 5)
            el2, _ = env.timed(lambda k: step(k, gather_lists=True), steps=n2, warmup=1)
            with_lists = {"ms_per_step": round(el2 / n2 * 1e3, 4), "value": round(job_pairs * n_views / (el2 / n2) / 1e9, 3),
                          "list_bytes_per_rank": int(totals_all.max()) * 8 * world}
    pre_marked, pre_tested, pre_per_group = ctx.precull_stats()  # of the last timed cull (view slot 0)
    grp_dropped, grp_tested = ctx.precull_group_stats()
    if os.environ.get("MCULL_BENCH_GROUP_CENSUS") and workload == "config3" and rank == 0:
        # experiment: of the groups the mask pass was handed, how many hold a meshlet that passes frustum + occlusion?
        for t in (tests & ~mc.TEST_CONE, tests):
            ctx.cull(views[0], tests=t, write_mask=True, list_ptr=lst.data_ptr(), list_capacity=cap, commands_ptr=cmds[1].data_ptr(),
                     total_ptr=total[1].data_ptr())
            env.fence()
            m = ctx.download_mask()
            gd, gt = ctx.precull_group_stats()
            pm, pt, _ = ctx.precull_stats()
            full = int((m == np.uint64(0xffffffffffffffff)).sum())
            per_inst = (m.reshape(n_inst, -1) != 0)
            print("census tests=%d: instances with a visible meshlet %d, with every group holding one %d"
                  % (t, int(per_inst.any(1).sum()), int(per_inst.all(1).sum())), file=sys.stderr)
            print("census tests=%d: groups %d, instance level kept %d of %d instances, group level handed %d, listed %d, "
                  "groups with a visible meshlet %d (all 64 visible: %d), visible pairs %d"
                  % (t, m.size, pt - pm, pt, gt, gt - gd, int((m != 0).sum()), full, int(total[1].item())), file=sys.stderr)
    hiz_pairs = 0
    if use_hiz and workload == "config4":
        # pairs that reach the occlusion test (one 4-byte texel each in the algorithmic bytes): the survivors
        # of the two arithmetic tests, counted by one untimed cull without the occlusion test
        ctx.cull(views[0], tests=mc.TEST_FRUSTUM | mc.TEST_CONE, list_ptr=lst.data_ptr(), list_capacity=cap,
                 commands_ptr=cmds[1].data_ptr(), total_ptr=total[1].data_ptr())
        env.fence()
        hiz_pairs = int(total[1].item())
    vis_views = None
    if workload == "config5" and not args.separate_views:
        vis_views = []
        for v in range(n_views):
            ctx.select_result(v)
            vis_views.append(ctx.download_count())
    return dict(scaling=scaling, elapsed=elapsed, steps=steps, region_ms_per_step=region_ms_per_step,
                kernel_ms=float(np.mean(kernel_ms)), visible=visible, job_visible=job_visible, with_lists=with_lists,
                n_inst=n_inst, n_scene=n_scene, per=per, n_pairs=n_pairs, job_pairs=job_pairs, n_views=n_views, unique=unique,
                tests=tests, views=views, depth=depth, hiz_size=hiz_size, vis_views=vis_views, hiz_pairs=hiz_pairs,
                precull=dict(marked=pre_marked, tested=pre_tested, unit="groups" if pre_per_group else "instances",
                             **({"group_level": {"dropped": grp_dropped, "tested": grp_tested}} if grp_tested else {})),
                value=job_pairs * n_views / (elapsed / steps) / 1e9, gather_mode=env.gather_mode, moving_ms=moving_ms, two_phase_ms=two_phase_ms,
                scene=(cone, sphere, xf, off, cnt))


def run_cull(env, workload):
    args, mc, device = env.args, env.mc, env.device
    rank, world, multi = env.rank, env.world, env.multi
    r = cull_leg(env, workload, args.scaling, gather_lists=args.gather_lists)
    n_inst, per, n_pairs, n_views, unique, visible = r["n_inst"], r["per"], r["n_pairs"], r["n_views"], r["unique"], r["visible"]
    out = None
    if rank == 0:
        k_ms = r["region_ms_per_step"] if world == 1 else r["kernel_ms"]
        # Algorithmic bytes: 32 B per unique meshlet bound TOUCHED + 64 B per transform + 8 B per command + 8 B per
        # visible pair (+ 4 B per pair that reaches the occlusion test).  The instance pre-cull drops whole
        # instances before any of their bounds is touched, so those bounds are not part of the figure the rate is
        # computed from; the figure with every bound counted (the definition before the pre-cull existed) is
        # reported beside it as `all_bounds`.
        pre = r["precull"]
        dropped = 0.0
        if pre["tested"] and workload != "config3":  # unique bounds per instance (config 3's bounds are one shared mesh)
            dropped = 32.0 * (pre["marked"] * 64.0 if pre["unit"] == "groups" else pre["marked"] * float(per))
        if workload == "config5" and not args.separate_views:
            # bounds and transforms are read once for all views; each view writes its own list + commands
            algo_all = 32.0 * unique + 64.0 * n_inst + sum(8.0 * n_inst + 8.0 * x for x in r["vis_views"])
        else:
            algo_all = n_views * (32.0 * unique + 64.0 * n_inst + 8.0 * n_inst + 8.0 * visible)
        algo_all += 4.0 * r["hiz_pairs"]  # config 4 with the occlusion test on: one texel per pair that reaches it
        algo = algo_all - n_views * dropped
        if dropped and r["hiz_pairs"]:
            # with the instance pre-cull most of those pairs never reach the test; how many do is not counted, so the
            # figure of the bytes touched leaves the texel reads out altogether (a lower bound of the bytes)
            algo -= 4.0 * r["hiz_pairs"]
        kernels = "instance_precull_kernel + cull_mask_kernel + scan_tiles_kernel + compact_kernel + emit_commands_kernel (one step)"
        shard_note = ("one %d-instance scene split into %d contiguous shards" % (r["n_scene"], world)) if r["scaling"] == "strong" \
            else ("one whole scene per GPU")
        extra_cfg = {"meshlet_instances_per_gpu": n_pairs * n_views, "visible_fraction": round(visible / max(n_pairs, 1), 4),
                     **({"pairs_reaching_hiz": r["hiz_pairs"]} if r["hiz_pairs"] else {}),
                     "instance_precull": r["precull"],
                     **({"collective": env.collective, "gather": env.gather_mode} if multi else {}),
                     **({"camera": "a different Globals every step: 0.2 degrees of orbit per step, 64-step cycle"} if args.moving_camera and workload == "config4" else {}),
                     "parallelism": ("%s, one RCCL all-gather of commands + counts per step, rebased after the gather" % shard_note)
                     if multi else "single GPU"}
        if workload == "config5":
            if r["vis_views"]:
                extra_cfg["visible_per_view"] = [int(x) for x in r["vis_views"]]
            w, h = r["hiz_size"]
            algo += n_views * (4.0 * w * h + 4.0 * mc.hiz_layout(w, h)[4])
            algo_all += n_views * (4.0 * w * h + 4.0 * mc.hiz_layout(w, h)[4])
            kernels = ("hiz_rows_views_kernel + hiz_tile_views_kernel (8 pyramids) + classify_views_kernel + cull_mask_views_kernel + scan_tiles_views_kernel + compact_views_kernel + emit_commands_views_kernel"
                       if not args.separate_views else
                       "8 x (hiz_rows_kernel + hiz_tile_kernel + cull_mask_kernel + scan_tiles_kernel + compact_kernel + emit_commands_kernel)")
            name = ("config 5: %d instances x %d unique meshlets against 8 views, one 8192^2 Hi-Z pyramid per view rebuilt "
                    "every step, frustum+cone+hiz, ordered compaction, %s (visible fraction is the last view's)"
                    % (n_inst, per, "one pass over the bounds for all views" if not args.separate_views else "one cull per view"))
        elif workload == "config3":
            name = ("config 3: %d instances of one %d-meshlet mesh (%s) = %d meshlet-instances per GPU, frustum+cone+hiz against a "
                    "3840x2160 pyramid (unflipped projection), ordered compaction; bounds are shared, so the HBM figure "
                    "counts little more than the visible list and excludes the Hi-Z texel reads"
                    % (n_inst, per, "the 871 200-triangle stand-in for dragon.obj, meshletised by the library"
                       if args.c3_mesh == "proxy" and not args.meshlets_per_instance else "unrelated random meshlets in a unit cube", n_pairs))
        else:
            name = ("config 4: synthetic scene, %d instances x %d unique meshlets = %d meshlet-instances per GPU, frustum+cone%s, "
                    "%s compaction" % (n_inst, per, n_pairs,
                                       ("+hiz(4K, %s projection)" % ("unflipped" if args.unflipped else "reference (flipped)")) if args.hiz else "",
                                       ["ordered", "wave-atomic", "block-atomic"][args.compaction]))
        out = base_line(env, "meshlet-instances culled+compacted/sec", "G/s", r["value"], r["elapsed"], r["steps"], name, extra_cfg)
        out["scaling"] = r["scaling"]
        if r["two_phase_ms"] is not None:
            out["two_phase_ms_per_pair"] = round(r["two_phase_ms"], 4)  # a frustum + cone cull and an all-tests cull, alternating
        if r["moving_ms"] is not None:
            out["moving_camera_ms_per_step"] = round(r["moving_ms"], 4)  # 0.2 degrees of orbit per step: every cull's list differs from the last one's
        traffic, traffic_src = pmc_traffic(args, workload, n_inst, per)
        out["roofline"] = roofline(algo, k_ms, kernels, traffic, traffic_src,
                                   kernel_ms_single_call=round(r["kernel_ms"], 4),
                                   bytes_per_pair=round(algo / max(n_pairs * n_views, 1), 2))
        if dropped:
            rate = algo_all / (k_ms * 1e-3) / 1e9
            out["roofline"]["all_bounds"] = {"algorithmic_bytes": int(algo_all), "achieved": round(rate, 1),
                                             "frac": round(rate / HBM_PEAK_GBS, 4),
                                             "bounds_bytes_not_touched": int(n_views * dropped),
                                             "note": "every meshlet bound counted as read, as before the instance pre-cull existed"}
        default_scene = not (args.no_precull or args.exact or args.tests is not None or args.meshlets_per_instance)
        if workload == "config3" and world == 1 and default_scene and (n_inst, args.c3_mesh) == (1_000_000, "proxy"):
            # the pass is bound by vector-instruction issue, not by bytes: that bound leads, the byte figure stays beside it
            step_kernels = (("instance_precull_kernel", 1), ("group_precull_kernel", 1), ("cull_mask_kernel", 1), ("scan_tiles_kernel", 1),
                            ("compact_kernel", 1), ("emit_commands_kernel", 1), ("zero_words_kernel", 1))
            counted = pmc_valu("r03_bench_config3", step_kernels)
            if counted:
                hbm = out["roofline"]
                if hbm.get("traffic"):
                    hbm["traffic_note"] = ("beside the lists the step zeroes the ballot word of every group (1.6 GB) and writes the two "
                                           "work lists and the listed groups' ballots: bytes the algorithmic figure leaves out")
                out["roofline"] = valu_roofline(counted, k_ms, kernels)
                out["roofline"]["hbm"] = hbm
        if workload == "config5" and world == 1 and default_scene and not args.separate_views and (n_inst, per) == (781250, 128):
            # the mask pass of the eight views alone is issue-bound; its duration comes from the committed kernel trace
            counted = pmc_valu("r03_bench_config5", (("cull_mask_views_kernel", 1),))
            rel = os.path.join("profiles", "r03_bench_config5_summary.json")
            if counted and os.path.exists(os.path.join(ROOT, rel)):
                with open(os.path.join(ROOT, rel)) as f:
                    us = [row.get("avg_us") for name, row in json.load(f).items() if "cull_mask_views_kernel" in name]
                if us and us[0]:
                    out["roofline"]["mask_pass_valu"] = valu_roofline(counted, us[0] * 1e-3, "cull_mask_views_kernel (duration from %s)" % rel)
        if world == 1:
            out["roofline"]["measured_device_copy"] = measured_copy_gbs(device)
            hbm_part = out["roofline"].get("hbm", out["roofline"])
            out["roofline"]["frac_of_measured_copy"] = round(hbm_part["achieved"] / out["roofline"]["measured_device_copy"], 4)
        if r["with_lists"]:
            out["with_list_gather"] = r["with_lists"]
        if not args.no_cpu_baseline and world == 1:
            cone, sphere, xf, off, cnt = r["scene"]
            out["cpu_baseline"] = cpu_baseline_cull(cone, sphere, xf, off, cnt, r["views"][-1], r["tests"], n_inst, per,
                                                    r["depth"], r["hiz_size"], workload)
    if world > 1 and args.scaling == "weak" and not args.no_strong_leg:
        # the same job as the statement of configuration 4 words it: ONE scene split over the ranks.  Reported beside the
        # weak-scaling value in the same line (the default `value` stays weak: fixed work per GPU).
        r.pop("scene")  # release the weak leg's scene before the second one is generated
        r2 = cull_leg(env, workload, "strong", steps=min(args.steps, 50), warmup=min(args.warmup, 5), gather_lists=args.gather_lists)
        if rank == 0:
            out["strong_scaling"] = {"value": round(r2["value"], 3), "unit": "G/s", "ms_per_step": round(r2["elapsed"] / r2["steps"] * 1e3, 4),
                                     "steps": r2["steps"], "meshlet_instances_total": r2["job_pairs"] * r2["n_views"],
                                     "meshlet_instances_per_gpu": r2["n_pairs"] * r2["n_views"],
                                     "kernel_ms_single_call": round(r2["kernel_ms"], 4),
                                     "visible_fraction": round(r2["job_visible"] / max(r2["job_pairs"], 1), 4),
                                     "gather": r2["gather_mode"], "with_list_gather": r2["with_lists"]}
    if rank == 0:
        env.line(out)


def pmc_summary_bytes(tag, kernels):
    """HBM bytes per launch summed over `kernels` from the committed rocprofv3 PMC passes of the same
    command (profiles/<tag>_summary.json: FETCH_SIZE doubled as the gfx950 correction prescribes,
    WRITE_SIZE as is; separate passes).  Counters cannot be collected from inside this process."""
    rel = os.path.join("profiles", "%s_summary.json" % tag)
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        summ = json.load(f)
    total = 0
    for name, row in summ.items():
        if any(k in name for k in kernels):
            total += row.get("fetch_bytes_corrected", 0) + row.get("write_bytes_corrected", 0)
    return (total or None), rel + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"


SIMDS = 1024  # 256 CUs x 4 SIMDs; a SIMD issues one vector instruction of a wave per 4 cycles


def pmc_valu(tag, kernels):
    """Vector instructions (per wave, summed over the waves) of one launch of each of `kernels`, summed, and the clock
    while they ran, from the committed rocprofv3 pass of the same command (profiles/<tag>_summary.json:
    SQ_INSTS_VALU, and GRBM_GUI_ACTIVE over the dispatch's duration).  `kernels`: (substring, launches per step)."""
    rel = os.path.join("profiles", "%s_summary.json" % tag)
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        summ = json.load(f)
    insts, cycles, weight = 0.0, 0.0, 0.0
    for name, row in summ.items():
        for key, per_step in kernels:
            if key in name and "valu_insts_per_launch" in row:
                insts += row["valu_insts_per_launch"] * per_step
                if "clock_mhz" in row:
                    cycles += row["clock_mhz"] * row["valu_insts_per_launch"] * per_step
                    weight += row["valu_insts_per_launch"] * per_step
    if not insts or not weight:
        return None
    return {"valu_insts": insts, "clock_mhz": cycles / weight, "source": rel + " (rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE)"}


def valu_roofline(counted, kernel_ms, kernel_name):
    """The issue-rate bound: `counted` vector instructions need counted x 4 cycles of the 1 024 SIMDs."""
    peak = SIMDS * counted["clock_mhz"] * 1e6 / 4.0 / 1e9  # G wave-instructions per second
    achieved = counted["valu_insts"] / (kernel_ms * 1e-3) / 1e9
    return {"bound": "valu", "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "G wave-instr/s",
            "frac": round(achieved / peak, 4), "traffic": None, "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4),
            "valu_insts_per_step": int(counted["valu_insts"]), "clock_mhz": round(counted["clock_mhz"], 1),
            "counters_source": counted["source"]}


def pmc_traffic(args, workload, n_inst, per):
    """Traffic of one step of a profiled default workload, or None when the workload differs from the profiled one."""
    if args.no_precull or args.exact or args.compaction or args.tests is not None or args.meshlets_per_instance:
        return None, None
    if workload == "config4" and (n_inst, per, args.hiz) == (781250, 128, False):
        return pmc_summary_bytes("r03_bench_config4", ("instance_precull_kernel", "cull_mask_kernel", "scan_tiles_kernel", "compact_kernel",
                                                       "emit_commands_kernel"))
    if workload == "config5" and (n_inst, per) == (781250, 128) and not args.separate_views:
        return pmc_summary_bytes("r03_bench_config5", ("hiz_rows_views_kernel", "hiz_tile_views_kernel", "hiz_top_views_kernel",
                                                       "classify_views_kernel", "cull_mask_views_kernel", "scan_tiles_views_kernel",
                                                       "compact_views_kernel", "emit_commands_views_kernel"))
    if workload == "config3" and (n_inst, args.c3_mesh) == (1_000_000, "proxy"):
        return pmc_summary_bytes("r03_bench_config3", ("instance_precull_kernel", "group_precull_kernel", "cull_mask_kernel", "scan_tiles_kernel",
                                                       "compact_kernel", "emit_commands_kernel", "zero_words_kernel"))
    return None, None


def cpu_baseline_cull(cone, sphere, xf, off, cnt, g, tests, n_inst, per, depth, hiz_size, workload):
    """The CPU oracle (a port of shaders/shader.task, test infrastructure) timed on this host: all
    host threads (OpenMP over instances) and one thread, on a bounded sample of the same scene."""
    import oracle as O

    threads = host_threads()
    shared = off is None
    sample = min(n_inst, 262144 if not shared else 2048)  # ~30 M pairs either way
    rows = sample * per if not shared else per
    c = cone[:rows].cpu().numpy()
    s = sphere[:rows].cpu().numpy()
    t = np.zeros(sample, dtype=O.TRANSFORM)
    t["model"] = xf[:sample].cpu().numpy()
    if tests & 4:
        w, h = hiz_size
        view = O.View(g, O.hiz_build(depth.cpu().numpy(), threads=threads), w, h)
    else:
        view = O.View(g)
    rng = None if shared else (off[:sample], cnt[:sample])

    def run(n, th):
        return O.cull_scene(view, t[:n], cone_axis_w=c, centre_radius=s, ranges=None if shared else (off[:n], cnt[:n]),
                            tests=tests, threads=th)

    run(min(sample, 1024), threads)
    reps, used = time_cpu(lambda: run(sample, threads), 10.0)
    one = max(1, sample // 8)
    reps1, used1 = time_cpu(lambda: run(one, 1), 4.0)
    return {"value": round(sample * per * reps / used / 1e9, 4), "unit": "G/s", "cores": threads, "kind": "port",
            "value_1_thread": round(one * per * reps1 / used1 / 1e9, 4),
            "sample": "first %d instances (%d meshlet-instances) of the same scene, %d passes, OpenMP over instances; "
                      "1 thread: first %d instances, %d passes" % (sample, sample * per, reps, one, reps1)}


# ------------------------------------------------------------------------------------------------
# Hi-Z pyramid build alone

def run_hiz(env):
    args, mc, ctx, device = env.args, env.mc, env.ctx, env.device
    w, h = args.hiz_size
    gen = torch.Generator(device=device).manual_seed(11 + env.rank)
    depth = torch.rand((h, w), device=device, dtype=torch.float32, generator=gen)
    torch.cuda.synchronize()

    def step(k):
        ctx.hiz_build(depth.data_ptr(), view=0, width=w, height=h, mem=mc.MEM_DEVICE)

    elapsed, dev_ms = env.timed(step)
    levels, _, _, _, total = mc.hiz_layout(w, h)
    algo = 4.0 * w * h + 4.0 * total
    value = w * h * env.world / (elapsed / args.steps) / 1e9
    if env.rank == 0:
        out = base_line(env, "Hi-Z pyramid pixels reduced/sec", "Gpixel/s", value, elapsed, args.steps,
                        "Hi-Z min pyramid of a %dx%d depth image, %d levels%s"
                        % (w, h, levels, " (replicas: every GPU builds its own)" if env.world > 1 else ""))
        traffic, src = pmc_summary_bytes("r03_bench_hiz", ("hiz_rows_kernel", "hiz_tile_kernel")) if (w, h) == (8192, 8192) else (None, None)
        out["roofline"] = roofline(algo, dev_ms / args.steps, "hiz_rows_kernel + hiz_tile_kernel", traffic, src,
                                   bytes_per_pixel=round(algo / (w * h), 3))
        if not args.no_cpu_baseline and env.world == 1:
            import oracle as O

            threads = host_threads()
            d = depth.cpu().numpy()
            O.hiz_build(d, threads=threads)
            reps, used = time_cpu(lambda: O.hiz_build(d, threads=threads), 8.0)
            reps1, used1 = time_cpu(lambda: O.hiz_build(d, threads=1), 6.0)
            out["cpu_baseline"] = {"value": round(w * h * reps / used / 1e9, 4), "unit": "Gpixel/s", "cores": threads, "kind": "port",
                                   "value_1_thread": round(w * h * reps1 / used1 / 1e9, 4),
                                   "sample": "the same %dx%d image, level by level as shader.comp, %d passes on %d threads, %d on one"
                                             % (w, h, reps, threads, reps1)}
        env.line(out)


# ------------------------------------------------------------------------------------------------
# meshlet bounds (cone + sphere) — load-time work of Engine::buildMeshletCon

def run_bounds(env):
    args, mc, ctx = env.args, env.mc, env.ctx
    copies = args.bounds_copies
    z = np.load(os.path.join(ROOT, "tests", "golden", "kitten_scene.npz"))
    ml, md, vt = z["meshlets"].view(mc.MESHLET), z["meshlet_data"], z["vertices"].view(mc.VERTEX)
    mls = []
    for c in range(copies):
        m2 = ml.copy()
        m2["dataOffset"] += c * len(md)
        mls.append(m2)
    mls = np.concatenate(mls)
    mds = np.tile(md, copies)
    ctx.upload_meshlets(mls)
    ctx.upload_geometry(vt, mds)

    def step(k):
        ctx.compute_bounds()

    elapsed, dev_ms = env.timed(step)
    v = mls["vertexCount"].astype(np.int64)
    t = mls["triangleCount"].astype(np.int64)
    algo = float((4 * v + 4 * ((3 * t + 3) // 4) + 12 * v + 48 + 36).sum())
    value = len(mls) * env.world / (elapsed / args.steps) / 1e6
    if env.rank == 0:
        out = base_line(env, "meshlet bounds (cone+sphere) computed/sec", "M/s", value, elapsed, args.steps,
                        "buildMeshletCon over %d copies of the kitten fixture's %d meshlets (%d meshlets, cache-resident vertices)"
                        % (copies, len(ml), len(mls)))
        out["roofline"] = roofline(algo, dev_ms / args.steps, "split_meshlets_kernel + bounds_kernel",
                                   bytes_per_meshlet=round(algo / len(mls), 1))
        if not args.no_cpu_baseline and env.world == 1:
            import oracle as O

            reps, used = time_cpu(lambda: O.build_meshlet_cones(vt, md, ml), 5.0)
            pos = np.ascontiguousarray(vt["pos"][:, :3])
            tris = []  # the fixture's triangles, recovered from the packed meshlet data
            for m in ml:
                o, nv, nt = int(m["dataOffset"]), int(m["vertexCount"]), int(m["triangleCount"])
                local = md[o + nv: o + nv + (3 * nt + 3) // 4].view("u1")[: 3 * nt]
                tris.append(md[o: o + nv][local])
            corners = np.full((sum(len(x) for x in tris), 3), -1, dtype="<i4")
            corners[:, 0] = np.concatenate(tris)
            t0 = time.perf_counter()
            O.build_mesh(pos, corners)
            whole = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(len(ml) * reps / used / 1e6, 4), "unit": "M/s", "cores": 1, "kind": "port",
                                   "sample": "buildMeshletCon over the fixture's %d meshlets, %d passes, one thread as in the reference; "
                                             "the whole load-time pipeline on the fixture (dedup, Forsyth reorder, meshlets, bounds) "
                                             "takes %.3f s on one thread" % (len(ml), reps, whole)}
        env.line(out)


# ------------------------------------------------------------------------------------------------
# config 1: the reference's own scene as a two-phase frame

def run_config1(env):
    """1000 kitten instances at 800x600 (Engine.cpp:2353-2379 transforms, :2323-2352 camera): phase-1
    cull (frustum + cone), depth pre-pass stand-in over the survivors, Hi-Z pyramid, phase-2 cull with
    all three tests - the frame of Engine.cpp:2381-2551 without the raster hardware."""
    args, mc, ctx = env.args, env.mc, env.ctx
    z = np.load(os.path.join(ROOT, "tests", "golden", "kitten_scene.npz"))
    ml, md, vt = z["meshlets"].view(mc.MESHLET), z["meshlet_data"], z["vertices"].view(mc.VERTEX)
    n_inst, w, h = 1000, 800, 600
    t = mc.reference_transforms(n_inst)  # the library's own updateTransforms / createGlobals (host code)
    g = mc.create_globals(w, h, len(ml))
    ctx.upload_meshlets(ml)
    ctx.upload_geometry(vt, md)
    ctx.upload_transforms(t)
    ms_parts = []

    def step(k, record=False):
        parts = []
        ctx.cull(g, tests=mc.TEST_FRUSTUM | mc.TEST_CONE)
        if record: parts.append(ctx.last_kernel_ms())
        ctx.depth_rasterize(g, w, h, clear=True)
        if record: parts.append(ctx.last_kernel_ms())
        ptr, _, _ = ctx.depth_buffer()
        ctx.hiz_build(ptr, view=0, width=w, height=h, mem=mc.MEM_DEVICE)
        if record: parts.append(ctx.last_kernel_ms())
        ctx.cull(g, tests=mc.TEST_ALL, hiz_view=0)
        if record:
            parts.append(ctx.last_kernel_ms())
            ms_parts.append(parts)

    elapsed, dev_ms = env.timed(step)
    graph_ms = None
    if args.graph:
        # The frame allocates nothing and never synchronises after its first run, so the caller can
        # capture its 11 launches into one HIP graph on the library's stream and replay it per frame.
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=env.stream):
            step(0)
        el_g, dev_g = env.timed(lambda k: graph.replay())
        graph_ms = dev_g / args.steps
        graph_visible = ctx.download_count()
        graph_depth = ctx.depth_download()
        # The same with the Globals in device memory, refreshed before every replay (a moving camera): the
        # kernels read them when the graph executes, so one capture serves every frame.
        cams = [mc.create_globals(w, h, len(ml), eye=(2.0 + 0.01 * i, 2.0, 2.0)) for i in range(8)]
        staged = [torch.from_numpy(np.frombuffer(c.tobytes(), dtype=np.uint8).copy()).to(env.device) for c in cams]
        g_dev = torch.zeros(212, dtype=torch.uint8, device=env.device)
        torch.cuda.synchronize()

        def dev_frame():
            ctx.cull_device_globals(g_dev.data_ptr(), len(ml), tests=mc.TEST_FRUSTUM | mc.TEST_CONE)
            ctx.depth_rasterize_device_globals(g_dev.data_ptr(), w, h)
            ptr, _, _ = ctx.depth_buffer()
            ctx.hiz_build(ptr, view=0, width=w, height=h, mem=mc.MEM_DEVICE)
            ctx.cull_device_globals(g_dev.data_ptr(), len(ml), tests=mc.TEST_ALL, hiz_view=0)

        with torch.cuda.stream(env.stream):
            g_dev.copy_(staged[0])
            dev_frame()
        torch.cuda.synchronize()
        graph_dev = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph_dev, stream=env.stream):
            dev_frame()

        def moving(k):
            with torch.cuda.stream(env.stream):
                g_dev.copy_(staged[k % len(staged)], non_blocking=True)
            graph_dev.replay()

        el_m, dev_m = env.timed(moving)
        graph_moving_ms = dev_m / args.steps
        with torch.cuda.stream(env.stream):
            g_dev.copy_(staged[0])
        graph_dev.replay()
        torch.cuda.synchronize()
        assert ctx.download_count() == graph_visible and np.array_equal(ctx.depth_download(), graph_depth)
    ctx.enable_timing(True)
    for k in range(10):
        step(k, record=True)
    ctx.enable_timing(False)
    env.fence()
    visible2 = ctx.download_count()
    if graph_ms is not None:  # the replayed frames must have produced the same frame
        assert graph_visible == visible2 and np.array_equal(graph_depth, ctx.depth_download())
    n_pairs = n_inst * len(ml)
    parts = np.mean(np.array(ms_parts), axis=0)
    value = 2 * n_pairs * env.world / (elapsed / args.steps) / 1e9
    if env.rank == 0:
        out = base_line(env, "meshlet-instances culled+compacted/sec", "G/s", value, elapsed, args.steps,
                        "config 1: the reference scene, %d kitten instances x %d meshlets at %dx%d as a two-phase frame "
                        "(cull frustum+cone -> depth pre-pass stand-in -> Hi-Z pyramid -> cull with all tests); both culls counted"
                        % (n_inst, len(ml), w, h),
                        {"meshlet_instances_per_frame": 2 * n_pairs, "visible_after_phase2": visible2,
                         "ms_cull1_raster_hiz_cull2": [round(float(x), 4) for x in parts],
                         "ms_per_frame_replayed_as_hip_graph": None if graph_ms is None else round(graph_ms, 4),
                         "ms_per_frame_replayed_with_moving_camera": None if graph_ms is None else round(graph_moving_ms, 4)})
        algo = 2 * (32.0 * len(ml) + 72.0 * n_inst) + 8.0 * visible2 + 9.33 * w * h
        out["roofline"] = roofline(algo, dev_ms / args.steps, "cull_mask + compact + emit, depth_raster, hiz_tile + hiz_generic, cull_mask + compact + emit",
                                   note="launch- and latency-bound at this size (11 launches per frame)")
        if not args.no_cpu_baseline and env.world == 1:
            import oracle as O

            tt = np.zeros(n_inst, dtype=O.TRANSFORM)
            tt["model"] = t["model"]

            def frame(threads):
                r1 = O.cull_scene(O.View(g), tt, ml, tests=3, threads=threads)
                img = O.rasterize_depth(vt, md, ml, tt, r1.list, g, w, h)
                pyr = O.hiz_build(img, threads=threads)
                return O.cull_scene(O.View(g, pyr, w, h), tt, ml, tests=7, threads=threads)

            threads = host_threads()
            ref = frame(threads)
            assert ref.total == visible2, "GPU frame and oracle frame disagree"
            reps, used = time_cpu(lambda: frame(threads), 8.0, 200)
            reps1, used1 = time_cpu(lambda: frame(1), 6.0, 50)
            out["cpu_baseline"] = {"value": round(2 * n_pairs * reps / used / 1e9, 4), "unit": "G/s", "cores": threads, "kind": "port",
                                   "value_1_thread": round(2 * n_pairs * reps1 / used1 / 1e9, 4),
                                   "sample": "the same frame (culls on %d threads, rasteriser on one), %d frames; one thread: %d frames"
                                             % (threads, reps, reps1)}
        env.line(out)


# ------------------------------------------------------------------------------------------------
# depth pre-pass stand-in: software rasterisation of the visible meshlets

def run_depth(env):
    args, mc, ctx = env.args, env.mc, env.ctx
    w, h = args.depth_size
    z = np.load(os.path.join(ROOT, "tests", "golden", "kitten_scene.npz"))
    ml, md, vt = z["meshlets"].view(mc.MESHLET), z["meshlet_data"], z["vertices"].view(mc.VERTEX)
    n_inst = args.instances or 4096
    t = env.host.synthetic_transforms(n_inst, extent=25.0, seed=9 + env.rank)
    g = env.host.create_globals(eye=(40.0, 38.0, 30.0), width=w, height=h, meshlet_count=len(ml))
    ctx.upload_meshlets(ml)
    ctx.upload_geometry(vt, md)
    ctx.upload_transforms(t)
    ctx.cull(g, tests=mc.TEST_FRUSTUM | mc.TEST_CONE)
    visible = ctx.download_count()
    lst = ctx.download_list()

    def step(k):
        ctx.depth_rasterize(g, w, h, clear=True)

    elapsed, dev_ms = env.timed(step)
    img = ctx.depth_download()
    covered = int((img > 0).sum())
    tri = ml["triangleCount"].astype(np.int64)[lst["meshletID"]]
    vc = ml["vertexCount"].astype(np.int64)[lst["meshletID"]]
    n_tri = int(tri.sum())
    algo = float((8 + 48 + 64 + 4 * vc + 4 * ((3 * tri + 3) // 4) + 24 * vc).sum()) + 4.0 * w * h + 4.0 * covered
    value = n_tri * env.world / (elapsed / args.steps) / 1e9
    if env.rank == 0:
        out = base_line(env, "triangles depth-rasterised/sec", "G/s", value, elapsed, args.steps,
                        "depth pre-pass stand-in: %d visible meshlets (%d triangles) of %d kitten instances into a %dx%d "
                        "reverse-Z image, clear included" % (visible, n_tri, n_inst, w, h),
                        {"visible_meshlets": visible, "triangles": n_tri, "covered_pixels": covered})
        out["roofline"] = roofline(algo, dev_ms / args.steps, "memset + depth_raster_kernel",
                                   note="per-pixel atomic max and per-triangle set-up bound, not a streaming kernel")
        if env.world == 1 and (n_inst, w, h) == (4096, 3840, 2160):
            # bound by vector-instruction issue (triangle set-up): that bound leads, the byte figure stays beside it
            counted = pmc_valu("r03_bench_depth", (("raster_prepare_kernel", 1), ("depth_raster_kernel", 1), ("depth_raster_jobs_kernel", 1)))
            if counted:
                hbm = out["roofline"]
                out["roofline"] = valu_roofline(counted, dev_ms / args.steps, "raster_prepare_kernel + depth_raster_kernel + depth_raster_jobs_kernel")
                out["roofline"]["hbm"] = hbm
        if not args.no_cpu_baseline and env.world == 1:
            import oracle as O

            sample = min(len(lst), 20000)
            tt = np.zeros(n_inst, dtype=O.TRANSFORM)
            tt["model"] = t["model"]
            sub = lst[:sample].view(O.VISIBLE)
            reps, used = time_cpu(lambda: O.rasterize_depth(vt, md, ml, tt, sub, g, w, h), 8.0, 50)
            tri_s = int(ml["triangleCount"].astype(np.int64)[sub["meshletID"]].sum())
            out["cpu_baseline"] = {"value": round(tri_s * reps / used / 1e9, 5), "unit": "G/s", "cores": 1, "kind": "port",
                                   "sample": "the first %d list entries (%d triangles) of the same frame, %d passes, one thread"
                                             % (sample, tri_s, reps)}
        env.line(out)


# ------------------------------------------------------------------------------------------------
# load-time front half on the GPU: vertex dedup + meshlet building (the reorderer is host code)

def run_meshbuild(env):
    args, mc, ctx = env.args, env.mc, env.ctx
    pos, tris = env.host.proxy_mesh(660, 660, seed=12)
    cor = np.full((tris.size, 3), -1, dtype="<i4")
    cor[:, 0] = tris.reshape(-1)
    n_tri = len(tris)
    state = {}

    def step(k):
        mesh = ctx.mesh_from_arrays(pos, cor)     # GPU dedup (host arrays in, host arrays out)
        ctx.build_meshlets(mesh)                  # GPU meshlets + bounds, mesh left uploaded
        state["mesh"] = mesh

    steps = min(args.steps, 10)
    elapsed, dev_ms = env.timed(step, steps=steps, warmup=min(args.warmup, 2))
    mesh = state["mesh"]
    n_meshlets, n_vert = len(mesh.meshlets), len(mesh.vertices)
    algo = 12.0 * cor.shape[0] + 12.0 * len(pos) + 4.0 * cor.shape[0] + 24.0 * n_vert + 4.0 * len(mesh.meshlet_data) + 48.0 * n_meshlets
    value = n_tri * env.world / (elapsed / steps) / 1e6
    if env.rank == 0:
        out = base_line(env, "triangles deduplicated + meshletised/sec", "M/s", value, elapsed, steps,
                        "load-time front half of a %d-triangle mesh (index order as given, the reorderer is host code on both "
                        "sides): vertex dedup, meshlet building and bounds on the GPU; host arrays in and out, so transfers and "
                        "allocations are inside the step" % n_tri, {"meshlets": n_meshlets, "vertices": n_vert})
        out["roofline"] = roofline(algo, dev_ms / steps, "corner_vertex/dedup_*/scan_*/meshlet_span/meshlet_emit/bounds kernels + copies",
                                   note="latency- and transfer-bound at this size, not a streaming kernel")
        if not args.no_cpu_baseline and env.world == 1:
            import oracle as O

            reps, used = time_cpu(lambda: O.build_mesh(pos, cor, optimize=False), 8.0, 50)
            t0 = time.perf_counter()
            host_mesh = mc.Mesh.from_arrays(pos, cor).build_meshlets()
            host_s = time.perf_counter() - t0
            assert np.array_equal(host_mesh.meshlet_data, mesh.meshlet_data)
            out["cpu_baseline"] = {"value": round(n_tri * reps / used / 1e6, 3), "unit": "M/s", "cores": 1, "kind": "port",
                                   "sample": "the same mesh through the oracle's dedup + createMeshlets + buildMeshletCon, %d passes, "
                                             "one thread; the product's own host path (hash-map dedup + meshlets, no bounds) takes "
                                             "%.3f s" % (reps, host_s)}
        env.line(out)


def run_rendezvous(args):
    """The multi-rank control flow alone, on the CPU over gloo (no GPU, no library): rendezvous, one
    all-reduce, rank 0 prints one JSON line.  Exists so that the launcher path of `--gpus N` can be tested
    where there is no GPU.  MCULL_BENCH_FAIL_RANK=r makes rank r exit with status 3 after the rendezvous."""
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    first, n = shard_range(781250, world, rank)
    t = torch.tensor([n, 1 << rank], dtype=torch.int64)
    if world > 1:
        with stdout_to_stderr():
            dist.init_process_group("gloo")
            dist.all_reduce(t)
    if os.environ.get("MCULL_BENCH_FAIL_RANK") == str(rank):
        os._exit(3)
    if rank == 0:
        print(json.dumps({"metric": "rendezvous", "n_gpus": world, "instances": int(t[0]), "ranks_seen": int(t[1])}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--c3-mesh", default="proxy", choices=["proxy", "random"],
                    help="config3: the mesh every instance draws - the procedural stand-in for dragon.obj built by the library's "
                         "mesh pipeline (default), or 13 916 unrelated random meshlets (the scene of the earlier measurements)")
    ap.add_argument("--workload", default="config4", choices=["config4", "config1", "config3", "config5", "hiz", "bounds", "depth", "meshbuild", "rendezvous"])
    ap.add_argument("--instances", type=int, default=0, help="override the workload's instance count")
    ap.add_argument("--meshlets-per-instance", type=int, default=0)
    ap.add_argument("--compaction", type=int, default=0, help="0 ordered, 1 wave atomic, 2 block atomic")
    ap.add_argument("--hiz", action="store_true", help="config4: also run the Hi-Z test against a 4K pyramid")
    ap.add_argument("--unflipped", action="store_true",
                    help="config4: projection without the reference's y flip, so that the Hi-Z test selects real mip levels")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="cull workloads with N ranks: weak = one whole scene per GPU (value = N scenes per step); strong = one "
                         "scene split into N contiguous instance shards (configuration 4 as its statement words it)")
    ap.add_argument("--no-strong-leg", action="store_true",
                    help="multi-rank weak runs also measure the strong-scaling form and report it as `strong_scaling`; this skips it")
    ap.add_argument("--hiz-size", type=int, nargs=2, default=(8192, 8192), metavar=("W", "H"))
    ap.add_argument("--depth-size", type=int, nargs=2, default=(3840, 2160), metavar=("W", "H"))
    ap.add_argument("--separate-views", action="store_true", help="config5: one cull call per view instead of one fused pass")
    ap.add_argument("--ctx-outputs", action="store_true", help="experiment: context-owned list/commands instead of torch tensors")
    ap.add_argument("--copy-bounds", action="store_true", help="experiment: copy the bounds into library-owned memory instead of adopting")
    ap.add_argument("--serial-gather", action="store_true",
                    help="multi-rank cull workloads: the all-gather on the cull's own stream instead of overlapped on a side stream "
                         "(the default below 16 M pairs per rank)")
    ap.add_argument("--overlap-gather", action="store_true", help="... always on the side stream")
    ap.add_argument("--moving-camera", action="store_true",
                    help="config4: a different camera every step (a slow orbit), so that each cull's list differs from the last one's")
    ap.add_argument("--ramp-ms", type=float, default=60.0,
                    help="milliseconds of untimed load before the warm-up steps (clock ramp after the idle set-up); 0 = off")
    ap.add_argument("--graph", action="store_true", help="config1: also replay the frame as one captured HIP graph")
    ap.add_argument("--gather-lists", action="store_true", help="multi-rank: also time the step with the visible lists all-gathered")
    ap.add_argument("--no-precull", action="store_true", help="cull workloads: switch the instance-level pre-cull off")
    ap.add_argument("--tests", type=int, default=None, help="cull workloads, experiments: MCULL_TEST_* bits to run instead of the workload's")
    ap.add_argument("--spatial-order", action="store_true",
                    help="config4/5, experiment: instances stored cell by cell of a 16^3 grid instead of in random order")
    ap.add_argument("--fov", type=float, default=45.0,
                    help="config4/5, experiments: vertical field of view of the camera in degrees (45 = the workload's; a wide one leaves "
                         "the instance level nothing to drop)")
    ap.add_argument("--exact", action="store_true",
                    help="cull workloads: every pair through the IEEE expressions instead of the guarded fast forms (same results)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bounds-copies", type=int, default=64, help="bounds workload: copies of the kitten fixture")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank gather path even with one rank")
    ap.add_argument("--torch-collective", action="store_true",
                    help="multi-rank step: torch.distributed.all_gather_into_tensor + the library's rebase instead of mcull_allgather_commands")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="control-flow rehearsal only: all ranks share GPU 0, collectives over gloo on host copies")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks here, as fresh child processes, before anything touches the GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.workload == "rendezvous":
        sys.exit(run_rendezvous(args))
    env = Env(args)
    if args.workload == "hiz":
        run_hiz(env)
    elif args.workload == "bounds":
        run_bounds(env)
    elif args.workload == "config1":
        run_config1(env)
    elif args.workload == "depth":
        run_depth(env)
    elif args.workload == "meshbuild":
        run_meshbuild(env)
    else:
        run_cull(env, args.workload)
    env.close()


if __name__ == "__main__":
    main()
