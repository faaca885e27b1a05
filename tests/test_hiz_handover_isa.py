"""The Hi-Z kernel that finishes the pyramid inside one launch hands texels from one workgroup to another
through memory: a producer stores its level-6 (and later level-9) texel write-through, drains the store
(`s_waitcnt vmcnt(0)`) and only then takes its ticket with a returning atomic; a consumer reads the texels with
agent-scope loads after its own ticket has returned.  That order is what makes the hand-over correct without
any cache-wide flush, and it rests on the instructions the compiler emits.  This test compiles the kernels
for gfx950 and pins them: scope bits on the hand-over stores and loads, the drain directly in front of every
ticket.  No GPU needed (hipcc cross-compiles)."""
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


def test_handover_stores_loads_and_tickets(kernels):
    for name, body in kernels.items():
        stores = [t for t in body if t.startswith("global_store") and "sc1" in t.split()]
        loads = [t for t in body if t.startswith("global_load") and "sc1" in t.split()]
        atomics = [i for i, t in enumerate(body) if t.startswith("global_atomic")]
        if not single_pass(name):
            assert not stores and not loads and not atomics, name
            continue
        # level 6 and level 9: one dword each, written through to memory (agent scope)
        assert len(stores) == 2 and all(t.startswith("global_store_dword ") for t in stores), (name, stores)
        # the consumers' reads: one level-6 texel per lane, then four level-9 texels per lane
        assert len(loads) == 5 and all(t.startswith("global_load_dword ") for t in loads), (name, loads)
        # two tiers of tickets, each a returning (sc0) add ...
        assert len(atomics) == 2, (name, [body[i] for i in atomics])
        for i in atomics:
            assert body[i].startswith("global_atomic_add ") and "sc0" in body[i].split(), (name, body[i])
            # ... with the hand-written drain in front of it and no memory instruction in between
            back = body[max(0, i - 40):i]
            ends = [j for j, t in enumerate(back) if "#ASMEND" in t]
            assert ends, (name, "no inline asm in front of the ticket")
            j = ends[-1]
            assert back[j - 1] == "s_waitcnt vmcnt(0)" and "#ASMSTART" in back[j - 2], (name, back[j - 2:j + 1])
            assert not [t for t in back[j + 1:] if is_mem(t)], (name, back[j + 1:])
        # a consumer's first hand-over load comes after a ticket in program text, never before the first one
        first_load = min(i for i, t in enumerate(body) if t.startswith("global_load") and "sc1" in t.split())
        assert first_load > atomics[0], name
