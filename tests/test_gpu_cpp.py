"""The C++ host mirror (include/mcull.hpp) end to end on the GPU: a small program built with g++
loads a scene cache file, records one frame with the reference-named methods and prints the visible
count and FNV-1a hashes of the draw list and commands, which must equal the oracle's frame."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "mcull.hpp"
#include <cstdio>
#include <fstream>
static uint64_t fnv(const void *p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= static_cast<const unsigned char *>(p)[i]; h *= 1099511628211ull; }
    return h;
}
template <typename T> static std::vector<T> slurp(const char *path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<T> v(static_cast<size_t>(f.tellg()) / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char *>(v.data()), v.size() * sizeof(T));
    return v;
}
int main(int argc, char **argv) {
    try {
        mcull::Culler culler(0);
        culler.loadScene(argv[1]);                                  // kitten.mcs
        culler.updateTransforms(slurp<mcull::Transform>(argv[2]));
        const mcull::Globals g = slurp<mcull::Globals>(argv[3]).at(0);
        culler.recordFrame(g, 800, 600);
        auto list = culler.visibleList();
        auto cmds = culler.indirectCommands();
        std::printf("visible %llu list %016llx cmds %016llx\n", (unsigned long long)list.size(),
                    (unsigned long long)fnv(list.data(), list.size() * sizeof(list[0])),
                    (unsigned long long)fnv(cmds.data(), cmds.size() * sizeof(cmds[0])));
        // the error path behaves like VK_CHECK: message on stderr + std::runtime_error
        try { culler.loadScene("/nonexistent.mcs"); } catch (const std::runtime_error &e) { std::printf("threw %s\n", e.what()); }
    } catch (const std::exception &e) {
        std::printf("FAILED %s\n", e.what());
        return 1;
    }
    return 0;
}
"""


def test_cpp_host_mirror_records_a_frame(tmp_path, mc, O, kitten, config1):
    g, t = config1["globals"], config1["transforms"]
    ml = kitten["meshlets"]
    mc.save_scene(tmp_path / "kitten.mcs", kitten["vertices"], None, kitten["meshlet_data"], ml)
    t.tofile(tmp_path / "transforms.bin")
    np.ascontiguousarray(g).tofile(tmp_path / "globals.bin")
    (tmp_path / "frame.cpp").write_text(SRC)
    lib_dir = os.path.dirname(mc.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(tmp_path / "frame.cpp"), "-o",
                           str(tmp_path / "frame"), "-L", lib_dir, "-lmcull", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
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
    assert "threw mcull error: MCULL_E_IO" in out.stdout and "mcull error" in out.stderr
