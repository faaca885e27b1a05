"""Register budgets of the cull kernels, read from the compiler's resource report for gfx950 (no GPU needed).
The streamed mask pass of the headline workload is bound by the bytes it has in flight: at 72 vector registers it
runs seven waves per SIMD, and a source change that looked neutral (its body moved into a lambda) once took it to 79
registers and six waves - 407 instead of 392 microseconds for 100 M pairs.  No kernel may spill."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vulkan-renderer_amd", "csrc", "cull_kernels.hip")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("res") / "cull.o"
    # the flags of the library's build (csrc/Makefile: the cull kernels are built without the SLP vectoriser)
    run = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only",
                          "-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", str(out)],
                         check=True, capture_output=True, text=True, timeout=900)
    kernels, name = {}, None
    for line in run.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and name:
            kernels[name][m.group(1).strip()] = m.group(2)
    assert len(kernels) > 40
    return kernels


def test_no_kernel_spills_vector_registers_or_uses_scratch(usage):
    """(Scalar registers parked in lanes of a vector register are tolerated: the instance pre-pass and the mask-pass
    forms with run-time tests do that with their view constants; it costs no memory traffic.)"""
    for name, u in usage.items():
        assert u["VGPRs Spill"] == "0" and u["ScratchSize"] == "0", (name, u)



def test_streamed_mask_pass_keeps_seven_waves_per_simd(usage):
    # cull_mask_kernel<TESTS = frustum | cone, NG = 4, NT = true, VP = false, FLAGS = false, PK = true | false>:
    # the two-group step (whole scene) and the one-group step (behind a work list)
    name = [k for k in usage if "cull_mask_kernelILj3ELi4ELb1ELb0ELb0E" in k]
    assert len(name) == 2, name
    for n in name:
        u = usage[n]
        assert int(u["VGPRs"]) <= 72 and int(u["Occupancy"]) >= 7 and u["SGPRs Spill"] == "0", (n, u)


def test_multi_view_pass_keeps_eight_waves_per_simd(usage):
    """The multi-view mask pass is bound by instruction issue and by the latency of its two dependent loads: eight waves
    per SIMD (64 vector registers) are what hides both.  (Its by-value view set must never have its address taken -
    a pointer to one of its members handed to a function call once put the whole 3 KB into scratch memory and the pass
    ran thirty times slower; the scratch assertion above covers that.)"""
    name = [k for k in usage if "cull_mask_views_kernel" in k]
    assert len(name) == 4, name
    for n in name:
        u = usage[n]
        assert int(u["VGPRs"]) <= 64 and int(u["Occupancy"]) == 8 and u["SGPRs Spill"] == "0" and u["ScratchSize"] == "0", (n, u)
