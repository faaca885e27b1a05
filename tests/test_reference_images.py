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
  * the vertex normals as the loader packs them into bytes (Engine.cpp:265-268: (n/2 + 1/2) * 255, truncated) and the
    vertex each one belongs to: the first capture's colour IS the interpolated normal (shader.mesh:78-79 decodes the
    bytes, shader.frag:12 writes (N + 1) / 2 to an sRGB target).
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
    return dict(h=h, w=w, foreground=fg, flat_yx=z["flat_yx"].astype(np.int64), flat_rgb=z["flat_rgb"].astype(np.int64),
                shade_yx=z["shade_yx"].astype(np.int64), shade_rgb=z["shade_rgb"].astype(np.float64))


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


def winners(O, kitten, config1):
    """Per pixel of the whole scene: depth, and the instance and the meshlet that own the nearest surface."""
    g, t = config1["globals"], config1["transforms"]
    ml, md, vt = kitten["meshlets"], kitten["meshlet_data"], kitten["vertices"]
    depth = O.rasterize_depth(vt, md, ml, t, all_pairs(O, len(t), len(ml)), g, W, H)
    inst = np.full((H, W), -1, dtype=np.int64)
    mid = np.full((H, W), -1, dtype=np.int64)
    one = np.zeros(len(ml), dtype=O.VISIBLE)
    one["meshletID"] = np.arange(len(ml))
    for i in range(len(t)):
        one["drawID"] = i
        d = O.rasterize_depth(vt, md, ml, t, one, g, W, H)
        inst[(d > 0) & (d == depth)] = i
    for m in range(len(ml)):
        d = O.rasterize_depth(vt, md, ml, t, all_pairs(O, len(t), len(ml), meshlet=m), g, W, H)
        mid[(d > 0) & (d == depth)] = m
    return depth, inst, mid


def interpolated_normal_bytes(kitten, config1, inst, mid, ys, xs):
    """The packed normal (three bytes, as real numbers) that a rasteriser hands the fragment shader at the centres of the
    pixels (ys, xs): the front-facing triangle of the pixel's instance and meshlet that is nearest there (reverse Z),
    its three vertex normals weighted perspective-correctly.  Float64 and a plain inside test: the pixels asked about lie
    inside smooth regions, where neither the fill rule nor the last bit of an edge function matters."""
    g, t = config1["globals"].reshape(-1)[0], config1["transforms"]
    ml, md, vt = kitten["meshlets"], kitten["meshlet_data"], kitten["vertices"]
    vp = np.array(g["proj"], dtype=np.float64).reshape(4, 4).T @ np.array(g["view"], dtype=np.float64).reshape(4, 4).T
    pos = vt["pos"][:, :3].astype(np.float64)
    nrm = vt["n"][:, :3].astype(np.float64)
    out = np.full((len(ys), 3), np.nan)
    keys = inst[ys, xs] * len(ml) + mid[ys, xs]
    order = np.argsort(keys, kind="stable")
    uk, starts = np.unique(keys[order], return_index=True)
    ends = list(starts[1:]) + [len(order)]
    for k, b, e in zip(uk, starts, ends):
        if k < 0:
            continue
        sel = order[b:e]
        i, m = divmod(int(k), len(ml))
        mvp = vp @ np.array(t[i]["model"], dtype=np.float64).reshape(4, 4).T
        first, nv, nt = int(ml[m]["dataOffset"]), int(ml[m]["vertexCount"]), int(ml[m]["triangleCount"])
        verts = md[first:first + nv]
        tri = md[first + nv:first + nv + (nt * 3 + 3) // 4].view(np.uint8)[:nt * 3].reshape(-1, 3)
        clip = np.concatenate([pos[verts], np.ones((nv, 1))], axis=1) @ mvp.T
        w = clip[:, 3]
        sx, sy, sz = (clip[:, 0] / w * 0.5 + 0.5) * W, (clip[:, 1] / w * 0.5 + 0.5) * H, clip[:, 2] / w
        a, b_, c = tri[:, 0], tri[:, 1], tri[:, 2]
        area = (sx[b_] - sx[a]) * (sy[c] - sy[a]) - (sx[c] - sx[a]) * (sy[b_] - sy[a])
        px, py = xs[sel] + 0.5, ys[sel] + 0.5

        def edge(p, q):
            return (sx[q] - sx[p])[None, :] * (py[:, None] - sy[p][None, :]) - (sy[q] - sy[p])[None, :] * (px[:, None] - sx[p][None, :])

        w0, w1, w2 = edge(b_, c), edge(c, a), edge(a, b_)
        inside = (w0 <= 0) & (w1 <= 0) & (w2 <= 0) & ((area < 0) & (w[a] > 0) & (w[b_] > 0) & (w[c] > 0))[None, :]
        z = np.where(inside, (w0 * sz[a][None, :] + w1 * sz[b_][None, :] + w2 * sz[c][None, :]) / area[None, :], -1.0)
        best = z.argmax(axis=1)
        rows = np.arange(len(sel))
        hit = inside[rows, best]
        q0, q1, q2 = (w0[rows, best] / w[a[best]], w1[rows, best] / w[b_[best]], w2[rows, best] / w[c[best]])  # (area cancels)
        n = (q0[:, None] * nrm[verts[a[best]]] + q1[:, None] * nrm[verts[b_[best]]] + q2[:, None] * nrm[verts[c[best]]]) / (q0 + q1 + q2)[:, None]
        out[sel[hit]] = n[hit]
    return out


def srgb_bytes(linear):
    x = np.clip(linear, 0.0, 1.0)
    return np.round(255.0 * np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055))


def test_packed_normals_reproduce_the_colours_of_the_shaded_screenshot(O, kitten, config1, shots):
    """shader.mesh:78-79 decodes the vertex's normal bytes to b / 255 * 2 - 1, the rasteriser interpolates, shader.frag:12
    writes (N + 1) / 2 = b / 255 to an sRGB target: a pixel of the first capture shows the interpolated normal BYTES.
    Computed here from this build's vertex buffer (the loader's bytes, its vertex dedup and order, the meshlets' local
    indices) they must give the capture's colours, and give them best as they are: bytes half a unit or a unit higher or
    lower - what rounding instead of Engine.cpp:265-268's truncation, or another scaling, would produce - fit worse."""
    _, inst, mid = winners(O, kitten, config1)
    ys, xs = shots["shade_yx"][:, 0], shots["shade_yx"][:, 1]
    assert ((inst[ys, xs] >= 0) & (mid[ys, xs] >= 0)).mean() > 0.999
    n = interpolated_normal_bytes(kitten, config1, inst, mid, ys, xs)
    got = ~np.isnan(n[:, 0])
    assert got.mean() > 0.999
    captured = shots["shade_rgb"][got]
    error = {}
    for shift in (-1.0, -0.5, 0.0, 0.5, 1.0):
        diff = np.abs(srgb_bytes((n[got] + shift) / 255.0) - captured)
        error[shift] = diff.mean()
        if shift == 0.0:
            assert (diff.max(axis=1) <= 2).mean() > 0.99, (diff.max(axis=1) <= 2).mean()
            assert (diff.max(axis=1) <= 4).mean() > 0.999
    assert error[0.0] < 0.65, error
    assert error[0.0] + 0.01 < error[0.5] < error[1.0] and error[0.0] + 0.05 < error[-0.5] < error[-1.0], error
    # and the statement is about the normals: the same bytes from the wrong vertices (the vertex buffer rotated by one) do not fit
    wrong = dict(kitten)
    wrong["vertices"] = kitten["vertices"].copy()
    wrong["vertices"]["n"] = np.roll(kitten["vertices"]["n"], 1, axis=0)
    m = interpolated_normal_bytes(wrong, config1, inst, mid, ys, xs)
    assert np.abs(srgb_bytes(m[got] / 255.0) - captured).mean() > 3.0 * error[0.0]


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
