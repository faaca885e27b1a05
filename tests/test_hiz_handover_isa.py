"""The Hi-Z kernel that finishes the pyramid inside one launch hands texels from one workgroup to another
through memory: a producer stores its level-6 (and later level-9) texel and takes a ticket with a RELEASE at agent
scope; the workgroup whose ticket is the last of its counter executes an ACQUIRE fence and reads the texels.
That is the memory model's own pattern, so nothing rests on hand-placed instructions any more; this test compiles the
kernels for gfx950 and checks that the pattern is what the compiler emitted - a cache write-back and a drained
store queue in front of every ticket, a cache invalidate behind it on the last arriver's path and in front of its
first hand-over load - and that no inline assembly is left in the file.  No GPU needed (hipcc cross-compiles)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vulkan-renderer_amd", "csrc", "hiz_kernels.hip")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("isa") / "hiz.s"
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "--cuda-device-only", "-S", SRC,
                    "-o", str(out)], check=True, capture_output=True, timeout=600)
    lines = out.read_text().split("\n")
    found = {}
    for i, line in enumerate(lines):
        m = re.match(r"^(_ZN5mcull\d+hiz_\w+?):", line)
        if not m:
            continue
        body = []
        for text in lines[i + 1:]:
            t = text.strip()
            if t.startswith(".Lfunc_end"):
                break
            if t and not t.startswith(";") or "#ASM" in t:
                body.append(t)
        found[m.group(1)] = body
    return found


def single_pass(name):
    m = re.search(r"hiz_tile_kernelILb[01]ELb([01])ELb[01]E", name)
    return m is not None and m.group(1) == "1"


def is_mem(t):
    return t.startswith(("global_", "buffer_", "flat_", "ds_", "scratch_"))


def test_every_instantiation_is_there(kernels):
    tiles = [k for k in kernels if "hiz_tile_kernel" in k]
    assert len(tiles) == 8 and sum(single_pass(k) for k in tiles) == 4
    assert len([k for k in kernels if "hiz_tile_views_kernel" in k]) == 2


def test_handover_is_release_on_the_ticket_and_acquire_by_the_last_arriver(kernels):
    for name, body in kernels.items():
        assert not [t for t in body if "#ASM" in t], (name, "inline assembly")
        atomics = [i for i, t in enumerate(body) if t.startswith("global_atomic")]
        if not single_pass(name):
            assert not atomics, name
            continue
        # two tiers of tickets, each a returning (sc0) add
        assert len(atomics) == 2, (name, [body[i] for i in atomics])
        for n, i in enumerate(atomics):
            assert body[i].startswith("global_atomic_add ") and "sc0" in body[i].split(), (name, body[i])
            # release at agent scope: this XCD's L2 written back and every earlier store of the wave drained
            back = body[max(0, i - 12):i]
            wb = [j for j, t in enumerate(back) if t.startswith("buffer_wbl2") and "sc1" in t.split()]
            assert wb, (name, "no L2 write-back in front of the ticket", back)
            waits = [t for t in back[wb[-1] + 1:] if t.startswith("s_waitcnt") and "vmcnt(0)" in t]
            assert waits, (name, "stores not drained between the write-back and the ticket", back)
            assert not [t for t in back[wb[-1] + 1:] if t.startswith(("global_store", "buffer_store", "flat_store"))], (name, back)
            # acquire by the last arriver: an invalidate between this ticket and the first hand-over load behind it
            end = atomics[n + 1] if n + 1 < len(atomics) else len(body)
            after = body[i + 1:end]
            inv = [j for j, t in enumerate(after) if t.startswith("buffer_inv") and "sc1" in t.split()]
            loads = [j for j, t in enumerate(after) if t.startswith("global_load") and "sc1" in t.split()]
            assert inv, (name, "no invalidate behind ticket %d" % n)
            if loads:
                assert inv[0] < loads[0], (name, "a hand-over load in front of the acquire")
        # the consumers' reads are agent-scope loads: one level-6 texel per lane, then four level-9 texels per lane
        loads = [t for t in body[atomics[0]:] if t.startswith("global_load") and "sc1" in t.split()]
        assert len(loads) == 5 and all(t.startswith("global_load_dword ") for t in loads), (name, loads)
