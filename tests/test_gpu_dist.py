"""vulkan-renderer_amd.dist on the GPU: the sharded two-phase frame with its collective on RCCL.  A one-GPU box
holds one rank, so the reduction itself is trivial here (the multi-rank merge is covered over gloo on the
CPU); what this checks is the plumbing a real run depends on: the library's depth image wrapped as a torch
tensor without a copy, and the collective ordered between the rasteriser and the pyramid on the library's
stream.  Runs in a child process: a process group belongs to its process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK = r'''
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
import torch
import torch.distributed as dist
mc = importlib.import_module("vulkan-renderer_amd")
D = importlib.import_module("vulkan-renderer_amd.dist")
import oracle as O

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
z = np.load(os.path.join(sys.argv[1], "tests", "golden", "kitten_scene.npz"))
ml, md, vt = z["meshlets"].view(mc.MESHLET).reshape(-1), z["meshlet_data"], z["vertices"].view(mc.VERTEX).reshape(-1)
t = mc.reference_transforms(1000)
w, h = 800, 600
stream = torch.cuda.Stream(device=0)
ctx = mc.Context(0, stream=stream.cuda_stream)
ctx.upload_meshlets(ml)
ctx.upload_geometry(vt, md)
ctx.upload_transforms(t)
for eye in (None, (3.0, 1.5, 2.5)):
    g = mc.create_globals(w, h, len(ml), eye=eye)
    D.sharded_frame(ctx, stream, g, w, h, tests=mc.TEST_ALL)
    total = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    r1 = O.cull_scene(O.View(g), t, ml, tests=3)
    img = O.rasterize_depth(vt, md, ml, t, r1.list, g, w, h)
    r2 = O.cull_scene(O.View(g, O.hiz_build(img), w, h), t, ml, tests=7)
    assert ctx.hiz_download(view=0)[0].view("<u4").tobytes() == img.view("<u4").tobytes()
    assert ctx.download_count() == r2.total and ctx.download_list().tobytes() == r2.list.tobytes()
    cmds = ctx.download_commands(len(t)).view(mc.CMD).reshape(-1)
    assert np.array_equal(cmds["firstTask"], r2.commands["firstTask"])
ctx.close()
dist.destroy_process_group()
print("sharded frame ok")
'''


def test_sharded_frame_over_rccl_world_size_one(tmp_path):
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rank.py"
    script.write_text(RANK)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script), ROOT, str(port)], capture_output=True, text=True, timeout=110, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "sharded frame ok" in out.stdout


@pytest.mark.parametrize("world,rows", [(1, 5), (4, 1001), (8, 1), (3, 200_000)])
def test_one_launch_rebase_equals_the_tensor_ops(mc, world, rows):
    """mcull_rebase_gathered_commands against the same arithmetic in tensor operations, on a made-up gathered
    buffer: counts above 2^32 included, so firstTask wraps and the 64-bit bases do not."""
    import importlib

    import torch

    D = importlib.import_module("vulkan-renderer_amd.dist")
    gen = torch.Generator().manual_seed(world * 1000 + rows)
    g = torch.randint(-(2**31), 2**31 - 1, (world * rows, 2), dtype=torch.int32, generator=gen)
    counts = torch.randint(0, 2**34, (world,), dtype=torch.int64, generator=gen)
    gv = g.view(world, rows, 2)
    gv[:, rows - 1, 0] = ((counts & 0xFFFFFFFF) - ((counts & 0x80000000) << 1)).to(torch.int32)
    gv[:, rows - 1, 1] = (counts >> 32).to(torch.int32)
    want_totals, want_cmds = D.rebase_gathered_tensor_ops(g.clone(), world, rows)
    assert torch.equal(want_totals, counts)
    dev = g.to("cuda:0")
    totals, cmds = D.rebase_gathered(dev, world, rows)
    torch.cuda.synchronize()
    assert torch.equal(totals.cpu(), want_totals) and torch.equal(cmds.cpu(), want_cmds)
    assert torch.equal(dev.view(world, rows, 2)[:, rows - 1].cpu(), gv[:, rows - 1]), "the count records stay as gathered"


NATIVE = r'''
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
with_torch_rccl = sys.argv[2] == "torch"
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], RANK="0", WORLD_SIZE="1")
import torch
mc = importlib.import_module("vulkan-renderer_amd")
D = importlib.import_module("vulkan-renderer_amd.dist")
H = importlib.import_module("vulkan-renderer_amd.host")
import oracle as O

torch.cuda.set_device(0)
if with_torch_rccl:  # PyTorch's own RCCL is in the process (and in use): the library must share it
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    comm = D.create_native_comm(0)
else:                # no process group at all: the id goes nowhere, the library loads RCCL itself
    comm = mc.RcclComm(mc.rccl_unique_id(), 1, 0, 0)
n_inst, per = 3000, 70
cone, sphere = H.synthetic_bounds(n_inst * per, seed=3)
t = H.synthetic_transforms(n_inst, extent=40.0, seed=4)
off = (np.arange(n_inst, dtype=np.uint64) * per).astype("<u4")
cnt = np.full(n_inst, per, dtype="<u4")
g = H.create_globals(eye=(100.0, 95.0, 80.0), fov_deg=50.0, width=640, height=360, flip_y=False)
stream, side = torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)
ctx = mc.Context(0, stream=stream.cuda_stream)
ctx.upload_bounds_soa(cone, sphere)
ctx.upload_transforms(t)
ctx.set_instance_ranges(off, cnt)
ref = O.cull_scene(O.View(g), t, cone_axis_w=cone, centre_radius=sphere, ranges=(off, cnt), tests=3)
packed = torch.zeros((n_inst + 1, 2), dtype=torch.int32, device="cuda:0")
lst = torch.empty((n_inst * per, 2), dtype=torch.int32, device="cuda:0")
gathered = torch.full((n_inst + 1, 2), -1, dtype=torch.int32, device="cuda:0")
out = torch.zeros((2, 1), dtype=torch.int64, device="cuda:0")
for frame in range(3):  # a frame loop: the gather of frame k on the side stream, ordered behind the cull by an event
    ctx.cull(g, tests=3, list_ptr=lst.data_ptr(), list_capacity=n_inst * per, commands_ptr=packed.data_ptr(),
             total_ptr=packed[n_inst:].data_ptr())
    ev = torch.cuda.Event()
    ev.record(stream)
    side.wait_event(ev)
    totals, cmds = D.gather_draw_list_native(comm, side, packed, gathered, out)
    side.synchronize()
    assert int(totals[0]) == ref.total and int(out[1, 0]) == 0
    got = cmds[0].cpu().numpy()
    assert np.array_equal(got[:, 0].astype(np.uint32), ref.commands["taskCount"]) and np.array_equal(got[:, 1].astype(np.uint32), ref.commands["firstTask"])
# a world that does not match the communicator is refused, as is a null buffer
for bad in (lambda: mc.allgather_commands(comm, side.cuda_stream, packed.data_ptr(), gathered.data_ptr(), 2, n_inst, out[0].data_ptr(), out[1].data_ptr()),
            lambda: mc.allgather_commands(comm, side.cuda_stream, None, gathered.data_ptr(), 1, n_inst, out[0].data_ptr(), out[1].data_ptr())):
    try:
        bad()
        raise SystemExit("a bad call went through")
    except mc.McullError as e:
        assert e.status == mc.E_INVALID
ctx.close()
comm.close()
if with_torch_rccl:
    dist.destroy_process_group()
print("native gather ok")
'''


@pytest.mark.parametrize("mode", ["torch", "alone"])
def test_native_allgather_of_commands_world_size_one(tmp_path, mode):
    """mcull_allgather_commands (ncclAllGather + rebase enqueued by the library on the caller's communicator and
    stream) against the oracle's commands, with PyTorch's RCCL already at work in the process and without."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "native.py"
    script.write_text(NATIVE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script), ROOT, mode, str(port)], capture_output=True, text=True, timeout=110, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "native gather ok" in out.stdout
