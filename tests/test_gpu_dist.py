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
