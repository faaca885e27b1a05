"""Seeded random scenes through the C ABI against the oracle: instance counts, ragged / empty /
overlapping meshlet ranges, cameras, projection flips, pyramid sizes (odd ones included), test
combinations, compaction modes, command layouts, and both kernel families (cache-resident and
streamed bounds).  Everything is compared bit for bit; the atomic compaction modes after sorting."""
import numpy as np
import pytest

from test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def contexts(mc):
    import os

    plain = mc.Context(0)
    os.environ["MCULL_FORCE_STREAMING"] = "1"
    try:
        streamed = mc.Context(0)
    finally:
        del os.environ["MCULL_FORCE_STREAMING"]
    yield plain, streamed
    plain.close()
    streamed.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_scene(mc, O, H, contexts, seed):
    rs = np.random.RandomState(1000 + seed)
    ctx = contexts[seed & 1]
    n_meshlets = int(rs.choice([1, 63, 64, 65, 200, 1000, 5000]))
    n_inst = int(rs.choice([1, 2, 7, 64, 257, 900]))
    cone, sphere = H.synthetic_bounds(n_meshlets, seed=seed)
    sphere = sphere.copy()
    sphere[:, :3] = (sphere[:, :3] - 0.5) * rs.uniform(0.5, 4.0)
    sphere[:, 3] *= rs.uniform(0.2, 5.0)
    t = H.synthetic_transforms(n_inst, extent=float(rs.uniform(2.0, 40.0)), seed=seed + 1)
    if rs.rand() < 0.5:  # non-uniform scale on some instances: the radius follows length(model[0].xyz) as in the reference
        m = t["model"].reshape(n_inst, 4, 4).copy()
        k = rs.randint(0, n_inst, max(1, n_inst // 3))
        m[k, :3, :3] *= rs.uniform(0.1, 3.0, (len(k), 3, 1)).astype("<f4")
        t["model"] = m.reshape(n_inst, 16)
    ranged = rs.rand() < 0.7
    if ranged:
        off = rs.randint(0, n_meshlets, n_inst).astype("<u4")
        cnt = np.minimum(rs.randint(0, min(n_meshlets, 300) + 1, n_inst), n_meshlets - off).astype("<u4")
        cnt[rs.rand(n_inst) < 0.15] = 0
        ranges = (off, cnt)
    else:
        ranges = None
    w, h = int(rs.choice([64, 100, 257, 640, 1023])), int(rs.choice([48, 37, 128, 360, 513]))
    depth = H.synthetic_depth(w, h, n_discs=int(rs.randint(1, 20)), seed=seed + 2)
    eye = tuple(rs.uniform(-30, 30, 3))
    g = H.create_globals(eye=eye, centre=tuple(rs.uniform(-5, 5, 3)), fov_deg=float(rs.uniform(20, 100)),
                         z_near=float(rs.choice([0.01, 0.1, 1.0])), width=w, height=h, flip_y=bool(rs.rand() < 0.5),
                         meshlet_count=n_meshlets)
    view_slot = int(rs.randint(0, mc.MAX_VIEWS))
    ctx.upload_bounds_soa(cone, sphere)
    ctx.upload_transforms(t)
    ctx.set_instance_ranges(*(ranges if ranged else (None, None)))
    ctx.hiz_build(depth, view=view_slot)
    view = O.View(g, O.hiz_build(depth), w, h)
    for _ in range(3):
        tests = int(rs.randint(0, 8))
        compaction = int(rs.choice([0, 0, 1, 2]))
        stride, offset = [(8, 0), (28, 20), (16, 4)][int(rs.randint(0, 3))]
        ref = O.cull_scene(view, t, cone_axis_w=cone, centre_radius=sphere, ranges=ranges, tests=tests)
        ctx.cull(g, tests=tests, hiz_view=view_slot, compaction=compaction, write_mask=True, cmd_stride=stride, cmd_offset=offset)
        assert np.array_equal(ctx.download_mask(), ref.mask), (seed, tests, compaction)
        check_against_oracle(mc, ctx, ref, n_inst, compaction, stride=stride, offset=offset)
    ctx.set_instance_ranges(None, None)
