"""CPU tests of the boundary (the shared library loads and exports the whole ABI; no compute calls
without a GPU) and of the multi-rank draw-list gather over gloo with world sizes 2 and 4."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(mc):
    header = open(os.path.join(ROOT, "include", "mcull.h")).read()
    declared = sorted(set(re.findall(r"\b(mcull_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found"
    assert sorted(mc.ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(mc.lib, name), name
    assert mc.version() >= 100
    assert mc.lib.mcull_status_string(0) == b"MCULL_OK" and mc.lib.mcull_status_string(-3) == b"MCULL_E_CAPACITY"


def test_no_gpu_means_loud_failure_not_fallback(mc):
    """Without a device the product must refuse to run (there is no CPU fallback)."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(mc.McullError) as e:
        mc.Context(0)
    assert e.value.status in (mc.E_HIP, mc.E_INVALID)
    assert mc.lib.mcull_cull(None, None, None, None) == mc.E_INVALID


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "vulkan-renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "mcull_oracle" not in text and "libmcull_oracle" not in text, f


def test_scene_cache_round_trip(tmp_path, mc, kitten, bunny):
    """The scene cache file: every section read back bit for bit, header counts, corruption and
    truncation detected, empty sections allowed (host code, no GPU)."""
    path = tmp_path / "kitten.mcs"
    n = 37
    off = (np.arange(n, dtype="<u4") * 3) % 200
    cnt = np.arange(n, dtype="<u4") % 65
    idx = np.arange(299, dtype="<u4")[::-1].copy()  # odd length: exercises the section padding
    mc.save_scene(path, kitten["vertices"], idx, kitten["meshlet_data"], kitten["meshlets"], off, cnt)
    info = mc.scene_info(path)
    assert (info.vertex_count, info.index_count, info.meshlet_data_words, info.meshlet_count, info.range_count) == (
        len(kitten["vertices"]), 299, len(kitten["meshlet_data"]), len(kitten["meshlets"]), n)
    got = mc.load_scene(path)
    assert got["vertices"].tobytes() == kitten["vertices"].tobytes()
    assert got["meshlets"].tobytes() == kitten["meshlets"].tobytes()
    assert np.array_equal(got["meshlet_data"], kitten["meshlet_data"]) and np.array_equal(got["indices"], idx)
    assert np.array_equal(got["range_offset"], off) and np.array_equal(got["range_count"], cnt)
    assert os.path.getsize(path) % 16 == 0
    # same content -> same checksum; different content -> different checksum
    mc.save_scene(tmp_path / "again.mcs", kitten["vertices"], idx, kitten["meshlet_data"], kitten["meshlets"], off, cnt)
    assert mc.scene_info(tmp_path / "again.mcs").checksum == info.checksum
    mc.save_scene(tmp_path / "bunny.mcs", bunny["vertices"], None, bunny["meshlet_data"], bunny["meshlets"])
    b = mc.load_scene(tmp_path / "bunny.mcs")
    assert b["meshlets"].tobytes() == bunny["meshlets"].tobytes() and len(b["indices"]) == 0 and len(b["range_count"]) == 0
    assert mc.scene_info(tmp_path / "bunny.mcs").checksum != info.checksum
    # one flipped payload byte -> checksum mismatch
    raw = bytearray(open(path, "rb").read())
    raw[96 + 1000] ^= 0x40
    (tmp_path / "corrupt.mcs").write_bytes(bytes(raw))
    with pytest.raises(mc.McullError) as e:
        mc.load_scene(tmp_path / "corrupt.mcs")
    assert e.value.status == mc.E_INVALID
    # truncated payload -> header no longer matches the file; wrong magic; missing file
    (tmp_path / "short.mcs").write_bytes(bytes(raw[:5000]))
    with pytest.raises(mc.McullError) as e:
        mc.load_scene(tmp_path / "short.mcs")
    assert e.value.status == mc.E_INVALID  # refused from its length, before anything is read
    (tmp_path / "junk.mcs").write_bytes(b"not a scene file" * 10)
    with pytest.raises(mc.McullError) as e:
        mc.scene_info(tmp_path / "junk.mcs")
    assert e.value.status == mc.E_INVALID
    with pytest.raises(mc.McullError) as e:
        mc.scene_info(tmp_path / "missing.mcs")
    assert e.value.status == mc.E_IO
    mc.save_scene(tmp_path / "empty.mcs")
    assert mc.scene_info(tmp_path / "empty.mcs").meshlet_count == 0 and os.path.getsize(tmp_path / "empty.mcs") == 96


def test_shard_instances(mc):
    D = __import__("importlib").import_module("vulkan-renderer_amd.dist")
    for n, w in ((1000, 8), (7, 3), (781250, 8), (5, 8)):
        cuts = [D.shard_instances(n, w, r) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
    weights = np.array([100] * 10 + [1] * 990)
    cuts = [D.shard_instances(1000, 2, r, weights) for r in range(2)]
    assert cuts[0] == (0, 10) or abs(weights[: cuts[0][1]].sum() - weights.sum() / 2) <= 100
    assert cuts[0][1] == cuts[1][0] and cuts[1][1] == 1000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_dir):
    import importlib
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = importlib.import_module("vulkan-renderer_amd.dist")
    H = importlib.import_module("vulkan-renderer_amd.host")
    import oracle as O

    z = np.load(os.path.join(ROOT, "tests", "golden", "kitten_scene.npz"))
    ml = z["meshlets"].view(O.MESHLET)
    n_inst = 64
    t = H.update_transforms(n_inst)
    g = H.create_globals(meshlet_count=len(ml))
    b, e = D.shard_instances(n_inst, world, rank)
    # the rank's shard, culled by the CPU oracle standing in for the GPU kernels
    r = O.cull_scene(O.View(g), t[b:e], ml, tests=3)
    total = torch.tensor([r.total], dtype=torch.int64)
    cmds = torch.from_numpy(r.commands.view("<i4").reshape(-1, 2).copy())

    def rebase(totals_all, rk):
        cmds[:, 1] += int(totals_all[:rk].sum())

    totals_all, cmds_all = D.gather_draw_list(total, cmds, rebase)
    np.save(os.path.join(out_dir, "cmds_%d.npy" % rank), cmds_all.numpy())
    np.save(os.path.join(out_dir, "totals_%d.npy" % rank), totals_all.numpy())
    np.save(os.path.join(out_dir, "list_%d.npy" % rank), r.list.view("<u4").reshape(-1, 2) + np.array([b, 0], dtype="<u4"))
    # the one-collective form: commands + count row gathered at once, rebased after the gather
    packed = torch.zeros((len(r.commands) + 1, 2), dtype=torch.int32)
    packed[:-1] = torch.from_numpy(r.commands.view("<i4").reshape(-1, 2).copy())
    packed[-1] = torch.from_numpy(np.array([r.total], dtype="<u8").view("<i4").copy())
    totals_p, cmds_p = D.gather_draw_list_packed(packed)
    assert torch.equal(totals_p, totals_all) and torch.equal(cmds_p.reshape(-1, 2), cmds_all)
    # the same into a buffer the caller keeps from frame to frame
    kept = torch.full((world * packed.shape[0], 2), -1, dtype=torch.int32)
    totals_k, cmds_k = D.gather_draw_list_packed(packed, gathered=kept)
    assert torch.equal(totals_k, totals_all) and torch.equal(cmds_k.reshape(-1, 2), cmds_all) and cmds_k.data_ptr() == kept.data_ptr()
    # optional third step: the packed global list (drawIDs made global before the gather)
    cap = len(ml) * (n_inst // world + 1)
    lst = torch.zeros((cap, 2), dtype=torch.int32)
    lst[: r.total] = torch.from_numpy((r.list.view("<u4").reshape(-1, 2) + np.array([b, 0], dtype="<u4")).view("<i4"))
    packed = D.gather_visible_lists(lst, totals_all)
    np.save(os.path.join(out_dir, "packed_%d.npy" % rank), packed.numpy().view("<u4"))
    # the same from the lists as the culls write them (drawIDs counted from the shard's first instance)
    lst_local = torch.zeros((cap, 2), dtype=torch.int32)
    lst_local[: r.total] = torch.from_numpy(r.list.view("<i4").reshape(-1, 2).copy())
    firsts = [D.shard_instances(n_inst, world, q)[0] for q in range(world)]
    packed2 = D.gather_visible_lists(lst_local, totals_all, first_instances=firsts)
    assert torch.equal(packed2, packed)
    if world > 1:
        assert not torch.equal(D.gather_visible_lists(lst_local, totals_all), packed), "rank-local drawIDs overlap"
    # the two-phase frame's exchange: every rank rasterises its own survivors, the images are merged
    w, h = 320, 240
    g2 = H.create_globals(width=w, height=h, meshlet_count=len(ml))
    r1 = O.cull_scene(O.View(g2), t[b:e], ml, tests=3)
    own = O.rasterize_depth(z["vertices"].view(O.VERTEX), z["meshlet_data"], ml, t[b:e], r1.list, g2, w, h)
    merged = D.merge_depth(torch.from_numpy(own.copy()))
    np.save(os.path.join(out_dir, "depth_%d.npy" % rank), merged.numpy())
    r2 = O.cull_scene(O.View(g2, O.hiz_build(merged.numpy()), w, h), t[b:e], ml, tests=7)
    np.save(os.path.join(out_dir, "list2_%d.npy" % rank), r2.list.view("<u4").reshape(-1, 2) + np.array([b, 0], dtype="<u4"))
    dist.barrier()
    dist.destroy_process_group()


def _native_comm_rank(rank, world, port, out_dir):
    import importlib
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = importlib.import_module("vulkan-renderer_amd.dist")
    try:
        D.create_native_comm(0)
        verdict = "made a communicator without a GPU"
    except Exception as e:  # noqa: BLE001
        verdict = "raised: %s" % e
    with open(os.path.join(out_dir, "native_%d.txt" % rank), "w") as f:
        f.write(verdict)
    dist.barrier()
    dist.destroy_process_group()


def test_native_communicator_failure_reaches_every_rank(tmp_path):
    """Without a GPU no rank can join a communicator (and where RCCL itself is missing, rank 0 cannot even make the id:
    every rank must then hear of it).  Each rank raises and goes on - to the barrier here, to the torch.distributed path in
    the bench - instead of waiting in the broadcast for an id that never comes."""
    import torch.multiprocessing as mp

    mp.spawn(_native_comm_rank, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        text = (tmp_path / ("native_%d.txt" % rank)).read_text()
        assert text.startswith("raised:"), text


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_draw_list_gather_equals_single_rank(tmp_path, O, H, kitten, world):
    """world_size 2 and 4 over gloo: concatenated lists == the single-rank list, gathered commands ==
    the single-rank commands (firstTask = exclusive scan in global instance order)."""
    import torch.multiprocessing as mp

    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ml = kitten["meshlets"]
    t = H.update_transforms(64)
    g = H.create_globals(meshlet_count=len(ml))
    ref = O.cull_scene(O.View(g), t, ml, tests=3)
    for rank in range(world):
        cmds = np.load(tmp_path / ("cmds_%d.npy" % rank)).astype("<u4")
        assert np.array_equal(cmds[:, 0], ref.commands["taskCount"])
        assert np.array_equal(cmds[:, 1], ref.commands["firstTask"])
        totals = np.load(tmp_path / ("totals_%d.npy" % rank))
        assert int(totals.sum()) == ref.total
    glob = np.concatenate([np.load(tmp_path / ("list_%d.npy" % r)) for r in range(world)])
    assert np.array_equal(glob, ref.list.view("<u4").reshape(-1, 2))
    for rank in range(world):  # the packed list every rank holds after the optional list gather
        assert np.array_equal(np.load(tmp_path / ("packed_%d.npy" % rank)), glob)
    # two-phase frame: the merged depth image of every rank is the unsharded image bit for bit, and the shards'
    # occlusion-tested lists concatenate to the unsharded frame's list
    w, h = 320, 240
    g2 = H.create_globals(width=w, height=h, meshlet_count=len(ml))
    r1 = O.cull_scene(O.View(g2), t, ml, tests=3)
    img = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], ml, t, r1.list, g2, w, h)
    assert (img > 0).any()
    for rank in range(world):
        assert np.load(tmp_path / ("depth_%d.npy" % rank)).view("<u4").tobytes() == img.view("<u4").tobytes()
    r2 = O.cull_scene(O.View(g2, O.hiz_build(img), w, h), t, ml, tests=7)
    glob2 = np.concatenate([np.load(tmp_path / ("list2_%d.npy" % r)) for r in range(world)])
    assert np.array_equal(glob2, r2.list.view("<u4").reshape(-1, 2))


def test_cpp_wrapper_compiles_and_throws_like_vk_check(tmp_path, mc):
    """include/mcull.hpp: compiles as C++17 against the C ABI; without a GPU the constructor throws
    std::runtime_error after printing to stderr (the reference's VK_CHECK behaviour, utils.hpp:68-79)."""
    import subprocess

    src = tmp_path / "t.cpp"
    src.write_text(r"""
#include "mcull.hpp"
#include <cstdio>
int main(int argc, char **) {
    std::printf("version %d\n", mcull_version());
    if (argc > 7) {  // compiled, never run here: the one-process-per-GPU exchange
        mcull::RankExchange x(mcull::RankExchange::createId(), 1, 0, 0);
        x.gatherCommands(nullptr, nullptr, nullptr, 0, nullptr, nullptr);
    }
    try { mcull::Culler c(0); std::printf("created\n"); }
    catch (const std::runtime_error &e) { std::printf("threw %s\n", e.what()); }
    return 0;
}""")
    exe = tmp_path / "t"
    lib_dir = os.path.dirname(mc.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", lib_dir, "-lmcull", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert "version %d" % mc.version() in out.stdout
    if torch.cuda.is_available():
        assert "created" in out.stdout
    else:
        assert "threw mcull error: MCULL_E_HIP" in out.stdout and "mcull error" in out.stderr


def test_header_is_plain_c_and_links_from_c(tmp_path, mc):
    """include/mcull.h from a C99 translation unit (the boundary is a C ABI, not a C++ one): the layout contract
    holds under the C compiler too, the host-only entry points work, and without a GPU mcull_create reports
    MCULL_E_HIP instead of computing anything."""
    import subprocess

    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stddef.h>
#include <stdio.h>
#include "mcull.h"
_Static_assert(sizeof(mcull_meshlet) == 48 && sizeof(mcull_globals) == 212 && sizeof(mcull_transform) == 64, "layout");
_Static_assert(sizeof(mcull_draw_indirect) == 28 && offsetof(mcull_draw_indirect, commandMeshIndirect) == 20, "DrawIndirect");
_Static_assert(sizeof(mcull_visible) == 8 && sizeof(mcull_task_payload) == 136 && sizeof(mcull_vertex) == 24, "layout");
int main(void) {
    mcull_globals g;
    mcull_transform t[3];
    mcull_draw_indirect d[3];
    if (mcull_create_globals(NULL, NULL, NULL, 45.0f, 0.1f, 800, 600, 1, 326, &g) != MCULL_OK) return 1;
    if (mcull_reference_transforms(3, 42, t) != MCULL_OK || mcull_create_indirect_buffer(3, 326, 86832, d) != MCULL_OK) return 2;
    if (g.meshletCount != 326 || d[2].commandMeshIndirect.taskCount != 11 || d[2].commandMeshIndirect.firstTask != 0) return 3;
    mcull_ctx *ctx = NULL;
    int rc = mcull_create(0, NULL, &ctx);
    printf("version %d levels %u create %s\n", mcull_version(), mcull_hiz_level_count(800, 600), mcull_status_string(rc));
    if (ctx) mcull_destroy(ctx);
    return 0;
}
''')
    lib_dir = os.path.dirname(mc.LIB_PATH)
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", lib_dir, "-lmcull", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "levels 10" in out.stdout
    import torch
    if not torch.cuda.is_available():
        assert "create MCULL_E_HIP" in out.stdout
