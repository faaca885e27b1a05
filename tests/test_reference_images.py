"""The reference's own output as a golden vector: two screenshots of its window that ship with it
(assets/mesh_pipeline_kittens.png, assets/meshlets_mesh_pipeline_kittens.png), reduced to pixel data
by tests/golden/make_reference_image_fixtures.py.

They show DRAW_COUNT = 1000 kittens placed by updateTransforms (Engine.cpp:2353-2377, srand(42)) and
seen through createGlobals' camera (Engine.cpp:2323-2352) at 800x600; the second one paints every
meshlet with the colour of its index (shader.mesh:51-58).  Rendering the same scene here and
comparing pins, against the reference itself and not only against a restatement of it:
  * transforms, camera, projection and viewport conventions: the silhouette;
  * the OBJ loader, the vertex dedup, the triangle reorderer and the meshlet builder: which meshlet
    index covers which pixel (any difference in the partition or in the numbering recolours it).
The captures are 796x599 (the window's client area as grabbed), aligned with the top-left of the
800x600 image; edges are compared with a 2-pixel tolerance.
"""
import os

import numpy as np
import pytest
from scipy import ndimage as ndi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H = 800, 600


@pytest.fixture(scope="module")
def shots():
    z = np.load(os.path.join(GOLDEN, "reference_screenshots.npz"))
    h, w = (int(x) for x in z["size"])
    fg = np.unpackbits(z["foreground_bits"])[: h * w].reshape(h, w).astype(bool)
    return dict(h=h, w=w, foreground=fg, flat_yx=z["flat_yx"].astype(np.int64), flat_rgb=z["flat_rgb"].astype(np.int64))


def all_pairs(O, n_inst, n_meshlets, meshlet=None):
    if meshlet is None:
        lst = np.zeros(n_inst * n_meshlets, dtype=O.VISIBLE)
        lst["drawID"] = np.repeat(np.arange(n_inst), n_meshlets)
        lst["meshletID"] = np.tile(np.arange(n_meshlets), n_inst)
    else:
        lst = np.zeros(n_inst, dtype=O.VISIBLE)
        lst["drawID"] = np.arange(n_inst)
        lst["meshletID"] = meshlet
    return lst


def meshlet_palette(n):
    """shader.mesh:51-58 for indices 0..n-1, written to an sRGB target (Engine.cpp:1960) as 8-bit values."""
    i = np.arange(n, dtype=np.float32)
    hue = i * np.float32(0.618033988749895)
    hue = hue - np.floor(hue)
    k = np.mod((hue * np.float32(6.0))[:, None] + np.array([0.0, 4.0, 2.0], dtype=np.float32), 6.0)
    rgb = 0.9 - 0.7 * np.clip(np.abs(k - 3.0) - 1.0, 0.0, 1.0)
    srgb = np.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * np.power(rgb, 1 / 2.4) - 0.055)
    return np.round(255.0 * srgb).astype(np.int64)


def check_silhouette(depth, shots):
    ours = (depth > 0)[: shots["h"], : shots["w"]]
    ref = shots["foreground"]
    iou = (ours & ref).sum() / (ours | ref).sum()
    # parts of the shaded kittens have nearly the clear colour, so the capture's mask has holes inside
    # the silhouettes; the strict statements are the two one-sided ones
    st = np.ones((5, 5), dtype=bool)
    ref_outside = (ref[2:] & ~ndi.binary_dilation(ours, st)[2:]).sum() / ref[2:].sum()  # rows 0-1 of the capture are window border
    bg_outside = (~ours & ~ndi.binary_dilation(~ref, st)).sum() / (~ours).sum()
    assert iou > 0.94, iou
    assert ref_outside < 1e-3, ref_outside  # the reference draws nothing where this build draws nothing
    assert bg_outside < 1e-3, bg_outside    # and leaves clear what this build leaves clear


def check_meshlet_colours(ids, shots, n_meshlets):
    pal = meshlet_palette(n_meshlets)
    y, x = shots["flat_yx"][:, 0], shots["flat_yx"][:, 1]
    mine = ids[y, x]
    covered = mine >= 0
    assert covered.mean() > 0.999
    diff = np.abs(pal[np.clip(mine, 0, n_meshlets - 1)] - shots["flat_rgb"]).max(-1)
    match = ((diff <= 4) & covered).mean()
    assert match > 0.99, match
    # the statement is not vacuous: a different numbering of the same partition fails it
    perm = np.random.RandomState(0).permutation(n_meshlets)
    wrong = ((np.abs(pal[perm][np.clip(mine, 0, n_meshlets - 1)] - shots["flat_rgb"]).max(-1) <= 4) & covered).mean()
    assert wrong < 0.1, wrong
    assert len(np.unique(mine[covered])) > 150  # and it speaks about most of the mesh's meshlets


def test_oracle_reproduces_the_reference_screenshots(O, kitten, config1, shots):
    g, t = config1["globals"], config1["transforms"]
    ml, md, vt = kitten["meshlets"], kitten["meshlet_data"], kitten["vertices"]
    depth = O.rasterize_depth(vt, md, ml, t, all_pairs(O, len(t), len(ml)), g, W, H)
    check_silhouette(depth, shots)
    ids = np.full((H, W), -1, dtype=np.int64)
    for m in range(len(ml)):
        d = O.rasterize_depth(vt, md, ml, t, all_pairs(O, len(t), len(ml), meshlet=m), g, W, H)
        ids[(d > 0) & (d == depth)] = m
    check_meshlet_colours(ids, shots, len(ml))


def test_oracle_culled_frame_still_shows_the_reference_image(O, kitten, config1, shots):
    """The captures were taken with the task shader's culling active.  Drawing only what survives this
    build's frame (frustum + cone cull, depth pre-pass, pyramid, cull with all three tests) must give
    the same picture: a predicate that removed anything the reference shows would leave holes."""
    g, t = config1["globals"], config1["transforms"]
    ml, md, vt = kitten["meshlets"], kitten["meshlet_data"], kitten["vertices"]
    r1 = O.cull_scene(O.View(g), t, ml, tests=3)
    pre = O.rasterize_depth(vt, md, ml, t, r1.list, g, W, H)
    r2 = O.cull_scene(O.View(g, O.hiz_build(pre), W, H), t, ml, tests=7)
    assert 0 < r2.total <= r1.total < len(t) * len(ml)
    depth = O.rasterize_depth(vt, md, ml, t, r2.list, g, W, H)
    check_silhouette(depth, shots)
    ids = np.full((H, W), -1, dtype=np.int64)
    order = np.argsort(r2.list["meshletID"], kind="stable")
    by_meshlet = r2.list[order]
    cuts = np.searchsorted(by_meshlet["meshletID"], np.arange(len(ml) + 1))
    for m in range(len(ml)):
        part = by_meshlet[cuts[m]:cuts[m + 1]]
        if len(part):
            d = O.rasterize_depth(vt, md, ml, t, part, g, W, H)
            ids[(d > 0) & (d == depth)] = m
    check_meshlet_colours(ids, shots, len(ml))


@pytest.mark.gpu
def test_gpu_path_reproduces_the_reference_screenshots(mc, kitten, shots):
    """The product end to end: the library's own createGlobals / updateTransforms, the GPU cull with
    no test enabled (every pair survives), the GPU rasteriser; one pass per meshlet index (through
    per-instance ranges) gives the index that wins each pixel."""
    ml, md, vt = kitten["meshlets"].view(mc.MESHLET), kitten["meshlet_data"], kitten["vertices"].view(mc.VERTEX)
    n = 1000
    g, t = mc.create_globals(W, H, len(ml)), mc.reference_transforms(n)
    with mc.Context(0) as ctx:
        ctx.upload_meshlets(ml)
        ctx.upload_geometry(vt, md)
        ctx.upload_transforms(t)
        ctx.cull(g, tests=0)
        assert ctx.download_count() == n * len(ml)
        ctx.depth_rasterize(g, W, H)
        depth = ctx.depth_download()
        check_silhouette(depth, shots)
        ids = np.full((H, W), -1, dtype=np.int64)
        ones = np.ones(n, dtype=np.uint32)
        for m in range(len(ml)):
            ctx.set_instance_ranges(np.full(n, m, dtype=np.uint32), ones)
            ctx.cull(g, tests=0)
            ctx.depth_rasterize(g, W, H)
            d = ctx.depth_download()
            ids[(d > 0) & (d == depth)] = m
        check_meshlet_colours(ids, shots, len(ml))
        # the same picture from what survives the culled two-phase frame
        ctx.set_instance_ranges(None, None)
        ctx.cull(g, tests=mc.TEST_FRUSTUM | mc.TEST_CONE)
        n_pre = ctx.download_count()
        ctx.depth_rasterize(g, W, H)
        ptr, _, _ = ctx.depth_buffer()
        ctx.hiz_build(ptr, view=0, width=W, height=H, mem=mc.MEM_DEVICE)
        ctx.cull(g, tests=mc.TEST_ALL)
        assert 0 < ctx.download_count() <= n_pre < n * len(ml)
        ctx.depth_rasterize(g, W, H)
        culled = ctx.depth_download()
        check_silhouette(culled, shots)
        ids[:] = -1
        for m in range(len(ml)):
            ctx.set_instance_ranges(np.full(n, m, dtype=np.uint32), ones)
            ctx.cull(g, tests=mc.TEST_ALL)  # against the pyramid of the whole scene, still in view 0
            ctx.depth_rasterize(g, W, H)
            d = ctx.depth_download()
            ids[(d > 0) & (d == culled)] = m
        check_meshlet_colours(ids, shots, len(ml))
