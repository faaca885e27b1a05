"""The C++ host mirror (include/mcull.hpp) end to end on the GPU: examples/engine_frame.cpp, built with g++,
loads a scene cache file, records one frame with the reference-named methods and prints the visible
count and FNV-1a hashes of the draw list and commands, which must equal the oracle's frame; it also replays
the frame as a HIP graph under a moving camera, splits the scene over three shards and runs the one-process-per-GPU
exchange in a world of one."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = os.path.join(ROOT, "examples", "engine_frame.cpp")


def test_cpp_host_mirror_records_a_frame(tmp_path, mc, O, kitten, config1):
    g, t = config1["globals"], config1["transforms"]
    ml = kitten["meshlets"]
    mc.save_scene(tmp_path / "kitten.mcs", kitten["vertices"], None, kitten["meshlet_data"], ml)
    t.tofile(tmp_path / "transforms.bin")
    np.ascontiguousarray(g).tofile(tmp_path / "globals.bin")
    lib_dir = os.path.dirname(mc.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"), SRC, "-o",
                           str(tmp_path / "frame"), "-L", lib_dir, "-lmcull", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(tmp_path / "frame"), str(tmp_path / "kitten.mcs"), str(tmp_path / "transforms.bin"),
                          str(tmp_path / "globals.bin")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    # the oracle's frame
    w, h = 800, 600
    r1 = O.cull_scene(O.View(g), t, ml, tests=3)
    img = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], ml, t, r1.list, g, w, h)
    r2 = O.cull_scene(O.View(g, O.hiz_build(img), w, h), t, ml, tests=7)
    want = "visible %d list %016x cmds %016x" % (r2.total, O.fnv1a64(r2.list), O.fnv1a64(r2.commands))
    assert want in out.stdout, (want, out.stdout)
    assert "device globals frame equal" in out.stdout
    assert "three shards equal" in out.stdout
    assert "rank exchange equal" in out.stdout  # RankExchange over the system's RCCL, a world of one rank
    assert out.stdout.count("graph frame") == 3
    assert "threw mcull error: MCULL_E_IO" in out.stdout and "mcull error" in out.stderr
