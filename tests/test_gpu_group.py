"""mcull_group_*: one host thread driving several shards of a scene, against the oracle's cull of the unsharded
scene.  A one-GPU box gives every shard the same ordinal (several shards on one GPU); the exchange then runs
over device-to-device copies on that GPU, the control flow and the packing kernel are the ones several GPUs use."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cuts(n, shards):
    return [(n * r) // shards for r in range(shards + 1)]


def _upload(grp, cuts, meshlets, transforms, depth=None):
    for r, s in enumerate(grp.shards):
        s.upload_meshlets(meshlets)
        s.upload_transforms(transforms[cuts[r]:cuts[r + 1]])
        if depth is not None:
            s.hiz_build(depth, view=0)


@pytest.mark.parametrize("copies", ["0", "1"])
@pytest.mark.parametrize("shards", [1, 3, 8])
def test_group_equals_unsharded_scene(mc, O, H, kitten, config1, shards, copies, monkeypatch):
    """Every device's packed command buffer equals the oracle's commands for the whole scene (DrawIndirect
    layout, 28/20), the concatenated shard lists equal its list, on three frames in a row with different tests;
    with the exchange as one push kernel per shard (direct stores) and as one copy per pair of devices."""
    monkeypatch.setenv("MCULL_GROUP_COPIES", copies)
    g, t = config1["globals"], config1["transforms"]
    depth = H.synthetic_depth(800, 600, n_discs=24, seed=5)
    view = O.View(g, O.hiz_build(depth), 800, 600)
    cuts = _cuts(len(t), shards)
    with mc.Group([0] * shards) as grp:
        _upload(grp, cuts, kitten["meshlets"], t, depth)
        for tests in (7, 3, 1):
            ref = O.cull_scene(view, t, kitten["meshlets"], tests=tests)
            res = grp.cull(g, tests=tests, cmd_stride=28, cmd_offset=20)
            assert [int(r.first_instance) for r in res] == cuts[:-1]
            lists = []
            for r, s in enumerate(grp.shards):
                raw, ft, totals = grp.download(r)
                assert ft is None
                assert int(totals[-1]) == ref.total and int(totals[:-1].sum()) == ref.total
                raw = raw.reshape(len(t), 28)
                assert not raw[:, :20].any()
                cmds = raw[:, 20:28].copy().view(mc.CMD).reshape(-1)
                assert np.array_equal(cmds["taskCount"], ref.commands["taskCount"])
                assert np.array_equal(cmds["firstTask"], ref.commands["firstTask"])
                lst = s.download_list().copy()
                assert len(lst) == int(totals[r])
                lst["drawID"] += cuts[r]
                lists.append(lst)
            assert np.concatenate(lists).tobytes() == ref.list.tobytes()


def test_group_ext_records_and_uneven_shards(mc, O, H, kitten, bunny):
    """EXT records: the scene-wide first_task array is gathered and rebased; shards of different sizes with
    per-instance ranges, one of them without any instance."""
    meshlets = np.concatenate([kitten["meshlets"], bunny["meshlets"]])
    nk, nb = len(kitten["meshlets"]), len(bunny["meshlets"])
    n = 301
    t = H.update_transforms(n)
    off = np.where(np.arange(n) % 2 == 0, 0, nk).astype("<u4")
    cnt = np.where(np.arange(n) % 2 == 0, nk, nb).astype("<u4")
    cnt[::7] = 0
    g = H.create_globals(meshlet_count=0)
    ref = O.cull_scene(O.View(g), t, meshlets, ranges=(off, cnt), tests=3)
    cuts = [0, 100, 100, 250, 301]
    with mc.Group([0, 0, 0, 0]) as grp:
        for r, s in enumerate(grp.shards):
            s.upload_meshlets(meshlets)
            s.upload_transforms(t[cuts[r]:cuts[r + 1]])
            s.set_instance_ranges(off[cuts[r]:cuts[r + 1]], cnt[cuts[r]:cuts[r + 1]])
        grp.cull(g, tests=3, cmd_form=mc.CMD_EXT_GROUPS)
        own = np.where(cnt > 0)[0]
        for r in range(4):
            raw, ft, totals = grp.download(r)
            rec = raw.view(mc.CMD_EXT).reshape(-1)
            assert np.array_equal(rec["groupCountX"], ref.commands["taskCount"])
            assert (rec["groupCountY"] == 1).all() and (rec["groupCountZ"] == 1).all()
            assert np.array_equal(ft[own], ref.commands["firstTask"][own])
            assert int(totals[-1]) == ref.total and int(totals[1]) == 0


def test_group_rejects_what_it_cannot_rebase(mc, kitten, config1):
    g, t = config1["globals"], config1["transforms"][:10]
    with mc.Group([0, 0]) as grp:
        _upload(grp, [0, 5, 10], kitten["meshlets"], t)
        with pytest.raises(mc.McullError) as e:
            grp.cull(g, tests=3, compaction=mc.COMPACT_WAVE_ATOMIC)
        assert e.value.status == mc.E_INVALID
        with pytest.raises(mc.McullError):
            grp.cull(g, tests=3, list_ptr=8)
        with pytest.raises(mc.McullError):
            grp.cull(g, tests=3, cmd_stride=10)
    with pytest.raises(mc.McullError):
        mc.Group([])
    with pytest.raises(mc.McullError):
        mc.Group([99])


@pytest.mark.parametrize("shards", [1, 3])
def test_group_two_phase_frame_equals_the_unsharded_frame(mc, O, kitten, config1, shards):
    """cull -> rasterise per shard -> merged depth -> pyramid -> cull with occlusion: the merged depth image
    (level 0 of every device's pyramid), the scene-wide commands and the concatenated lists equal the oracle's
    two-phase frame of the unsharded scene, on two frames in a row with different cameras."""
    t = config1["transforms"]
    ml, md, vt = kitten["meshlets"], kitten["meshlet_data"], kitten["vertices"]
    w, h = 800, 600
    cuts = _cuts(len(t), shards)
    with mc.Group([0] * shards) as grp:
        for r, s in enumerate(grp.shards):
            s.upload_meshlets(ml)
            s.upload_geometry(vt, md)
            s.upload_transforms(t[cuts[r]:cuts[r + 1]])
        for eye in (None, (3.0, 1.5, 2.5)):
            g = mc.create_globals(w, h, len(ml), eye=eye)
            r1 = O.cull_scene(O.View(g), t, ml, tests=3)
            img = O.rasterize_depth(vt, md, ml, t, r1.list, g, w, h)
            r2 = O.cull_scene(O.View(g, O.hiz_build(img), w, h), t, ml, tests=7)
            assert 0 < r2.total <= r1.total
            grp.frame(g, w, h)
            lists = []
            for r, s in enumerate(grp.shards):
                assert s.hiz_download(view=0)[0].view("<u4").tobytes() == img.view("<u4").tobytes()
                raw, _, totals = grp.download(r)
                cmds = raw.view(mc.CMD).reshape(-1)
                assert int(totals[-1]) == r2.total
                assert np.array_equal(cmds["taskCount"], r2.commands["taskCount"])
                assert np.array_equal(cmds["firstTask"], r2.commands["firstTask"])
                lst = s.download_list().copy()
                lst["drawID"] += cuts[r]
                lists.append(lst)
            assert np.concatenate(lists).tobytes() == r2.list.tobytes()
