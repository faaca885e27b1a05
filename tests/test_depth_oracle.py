"""Known-answer tests of the oracle's depth pre-pass stand-in (software rasteriser): coverage with
the top-left fill rule, exact depth interpolation on dyadic inputs, face culling, eye-plane
rejection.  CPU only."""
import numpy as np


def _scene(O, tris_xyz):
    """Meshlet records for a list of triangles given as (3,3) arrays of clip-space xyz (w = 1)."""
    verts = np.zeros(3 * len(tris_xyz), dtype=O.VERTEX)
    data = []
    meshlets = np.zeros(len(tris_xyz), dtype=O.MESHLET)
    for i, t in enumerate(tris_xyz):
        for k in range(3):
            verts[3 * i + k]["pos"][:3] = t[k]
        meshlets[i]["dataOffset"] = len(data)
        meshlets[i]["vertexCount"] = 3
        meshlets[i]["triangleCount"] = 1
        data += [3 * i, 3 * i + 1, 3 * i + 2, 0 | (1 << 8) | (2 << 16)]
    return verts, np.array(data, dtype="<u4"), meshlets


def _identity_globals(O):
    g = np.zeros(1, dtype=O.GLOBALS)
    eye = np.eye(4, dtype="<f4").reshape(16)
    g["view"][0] = eye
    g["proj"][0] = eye
    return g


def _draw(O, tris, w, h, which=None, model=None):
    verts, data, meshlets = _scene(O, tris)
    t = np.zeros(1, dtype=O.TRANSFORM)
    t["model"][0] = np.eye(4, dtype="<f4").reshape(16) if model is None else model
    ids = range(len(tris)) if which is None else which
    lst = np.array([(0, i) for i in ids], dtype=O.VISIBLE)
    return O.rasterize_depth(verts, data, meshlets, t, lst, _identity_globals(O), w, h)


def _ndc(px, py, w, h):
    return (2.0 * px / w - 1.0, 2.0 * py / h - 1.0)


def _brute(tri_px, z, w, h):
    """Independent evaluation in float64 with the same rules (inputs are dyadic, so exact)."""
    (x0, y0), (x1, y1), (x2, y2) = tri_px
    out = np.zeros((h, w))
    area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
    if not area < 0:
        return out

    def edge(ax, ay, bx, by, px, py):
        return (bx - ax) * (py - ay) - (by - ay) * (px - ax)

    def tl(ax, ay, bx, by):
        return (by - ay) > 0 or ((by - ay) == 0 and (bx - ax) < 0)

    for py in range(h):
        for px in range(w):
            cx, cy = px + 0.5, py + 0.5
            ws = [edge(x1, y1, x2, y2, cx, cy), edge(x2, y2, x0, y0, cx, cy), edge(x0, y0, x1, y1, cx, cy)]
            tls = [tl(x1, y1, x2, y2), tl(x2, y2, x0, y0), tl(x0, y0, x1, y1)]
            if all(wv < 0 or (wv == 0 and t) for wv, t in zip(ws, tls)):
                zz = (ws[0] * z[0] + ws[1] * z[1] + ws[2] * z[2]) / area
                if 0 <= zz <= 1:
                    out[py, px] = max(out[py, px], zz)
    return out


def test_coverage_and_depth_match_brute_force(O):
    w, h = 16, 8
    cases = [
        ([(1.0, 1.0), (3.0, 7.0), (13.0, 2.0)], [0.25, 0.5, 0.75]),     # counter-clockwise on a y-down screen
        ([(0.5, 0.5), (0.5, 6.5), (8.5, 0.5)], [0.125, 0.125, 0.125]),   # edges through pixel centres
        ([(-4.0, -4.0), (-4.0, 20.0), (40.0, -4.0)], [0.5, 0.5, 0.5]),   # larger than the screen
        ([(2.0, 2.0), (2.0, 2.0), (9.0, 5.0)], [0.5, 0.5, 0.5]),         # degenerate
    ]
    for tri_px, z in cases:
        tri = [(*_ndc(x, y, w, h), zz) for (x, y), zz in zip(tri_px, z)]
        got = _draw(O, [np.array(tri, dtype="<f4")], w, h)
        want = _brute(tri_px, z, w, h)
        assert np.array_equal(got > 0, want > 0), (tri_px, got, want)            # coverage: exact
        # depth: numerator and area are exact on these dyadic inputs, only the division rounds
        assert np.all(np.abs(got.astype(np.float64) - want) <= np.spacing(want.astype("<f4")).astype(np.float64))
    assert _draw(O, [np.array([(*_ndc(1, 1, w, h), .5), (*_ndc(3, 7, w, h), .5), (*_ndc(13, 2, w, h), .5)], dtype="<f4")], w, h).any()


def test_back_faces_and_eye_plane_are_dropped(O):
    w, h = 16, 8
    front = [(*_ndc(1, 1, w, h), 0.5), (*_ndc(3, 7, w, h), 0.5), (*_ndc(13, 2, w, h), 0.5)]
    back = [front[0], front[2], front[1]]
    assert _draw(O, [np.array(front, dtype="<f4")], w, h).any()
    assert not _draw(O, [np.array(back, dtype="<f4")], w, h).any()
    # a model matrix that zeroes w (w = 0 is "at the eye plane"): nothing is drawn
    model = np.eye(4, dtype="<f4")
    model[3, 3] = 0.0  # column-major [c][r]: w' = 0 * 1
    assert not _draw(O, [np.array(front, dtype="<f4")], w, h, model=model.reshape(16)).any()
    # depth outside [0, 1] is clipped per fragment
    far = [(x, y, 1.5) for x, y, _ in front]
    assert not _draw(O, [np.array(far, dtype="<f4")], w, h).any()


def test_shared_edge_is_covered_exactly_once_and_nearest_wins(O):
    w, h = 8, 8
    a, b, c, d = _ndc(1, 1, w, h), _ndc(1, 7, w, h), _ndc(7, 7, w, h), _ndc(7, 1, w, h)
    t1 = np.array([(*a, 0.25), (*b, 0.25), (*c, 0.25)], dtype="<f4")   # shares the diagonal a-c
    t2 = np.array([(*a, 0.25), (*c, 0.25), (*d, 0.25)], dtype="<f4")
    c1, c2 = _draw(O, [t1, t2], w, h, which=[0]) > 0, _draw(O, [t1, t2], w, h, which=[1]) > 0
    assert not (c1 & c2).any()                       # the diagonal's pixel centres belong to one triangle only
    both = _draw(O, [t1, t2], w, h) > 0
    assert np.array_equal(both, c1 | c2) and both[1:7, 1:7].all() and both.sum() == 36
    near = np.array([(*a, 0.75), (*b, 0.75), (*c, 0.75)], dtype="<f4")
    img = _draw(O, [t1, near], w, h)
    assert set(np.unique(img)) == {0.0, 0.75}       # reverse-Z: the larger depth is nearer and wins
    img2 = _draw(O, [near, t1], w, h)
    assert np.array_equal(img, img2)                 # order independent


def test_kitten_scene_renders_plausibly(O, H, kitten, config1):
    """Config 1 through the two-phase flow on the CPU: every pixel in [0, 1], a sensible share of the
    screen covered, and the second-phase cull with the resulting pyramid still sees every meshlet
    that contributed the nearest depth of some pixel it covers alone (weak sanity, not parity)."""
    g, t = config1["globals"], config1["transforms"][:60]
    ref = O.cull_scene(O.View(g), t, kitten["meshlets"], tests=3)
    img = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], kitten["meshlets"], t, ref.list, g, 800, 600)
    assert img.min() >= 0.0 and img.max() <= 1.0
    cover = (img > 0).mean()
    assert 0.005 < cover < 0.9
    # drawing the full (unculled) set gives the same image: culled meshlets are back-facing or off screen
    full = np.array([(i, m) for i in range(len(t)) for m in range(len(kitten["meshlets"]))], dtype=O.VISIBLE)
    img_full = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], kitten["meshlets"], t, full, g, 800, 600)
    assert np.array_equal(img, img_full)


def test_near_plane_clipping(O):
    """A triangle with one vertex behind the near plane is clipped in clip space (the visible part is
    a trapezoid), not dropped.  The model matrix makes clip.w = pos.z and clip.z = 0.25, so a vertex is
    inside the near plane iff pos.z >= 0.25; with these dyadic coordinates the intersections fall at
    t = 0.5 exactly and the expected image can be evaluated independently."""
    w = h = 16
    model = np.zeros((4, 4), dtype="<f4")        # [c][r]
    model[0, 0] = 1.0; model[1, 1] = 1.0         # x, y pass through
    model[3, 2] = 0.25                           # clip.z = 0.25 (constant)
    model[2, 3] = 1.0                            # clip.w = pos.z
    A, B, C = (-0.5, -0.5, 1.0), (0.5, -0.5, 1.0), (0.0, 1.0, -0.5)
    front = np.array([A, C, B], dtype="<f4")     # counter-clockwise on the y-down screen after clipping
    back = np.array([A, B, C], dtype="<f4")
    img = _draw(O, [front], w, h, model=model.reshape(16))
    assert not _draw(O, [back], w, h, model=model.reshape(16)).any()
    # clipped polygon in screen space: A, I(A,C), I(B,C), B with NDC depth 0.25 at A and B, 1.0 on the plane
    pa, pac, pbc, pb = (4.0, 4.0), (0.0, 16.0), (16.0, 16.0), (12.0, 4.0)
    want = np.maximum(_brute([pa, pac, pbc], [0.25, 1.0, 1.0], w, h), _brute([pa, pbc, pb], [0.25, 1.0, 0.25], w, h))
    assert np.array_equal(img > 0, want > 0)
    assert (img > 0).sum() > 100 and not (img[:4] > 0).any()
    assert np.all(np.abs(img.astype(np.float64) - want) <= np.spacing(want.astype("<f4")).astype(np.float64))
    # two vertices behind the plane: a single smaller triangle remains
    tri1 = np.array([(-0.5, -0.5, 1.0), (0.0, 1.0, -0.5), (1.0, 1.0, -0.5)], dtype="<f4")
    img1 = _draw(O, [tri1], w, h, model=model.reshape(16))
    img1r = _draw(O, [tri1[[0, 2, 1]]], w, h, model=model.reshape(16))
    assert (img1 > 0).any() != (img1r > 0).any()           # exactly one winding is front-facing
    # everything behind: nothing
    assert not _draw(O, [np.array([(0, 0, -1.0), (1, 0, -1.0), (0, 1, -1.0)], dtype="<f4")], w, h, model=model.reshape(16)).any()


def test_numpy_restatement_agrees_bit_for_bit(O, H, kitten, config1):
    """The separately written numpy rasteriser (vectorised over a triangle's pixel box) and the C
    oracle produce the same image, bit for bit: ordinary instances, and enlarged instances around the
    camera whose triangles are clipped by the near plane and cover hundreds of pixels."""
    from oracle import np_oracle as N

    g = config1["globals"]
    ml = kitten["meshlets"]
    pos = np.ascontiguousarray(kitten["vertices"]["pos"][:, :3])
    w, h = 320, 240
    t = config1["transforms"][:12].copy()
    model = t["model"].reshape(-1, 4, 4).copy()
    model[8:, :3, :3] *= np.float32(6.0)          # four large instances, some around the eye
    model[8:, 3, :3] = np.array([[1.6, 1.7, 1.8], [2.3, 2.2, 2.1], [0.5, 0.4, 0.6], [1.0, 2.5, 1.5]], dtype="<f4")
    t["model"] = model.reshape(-1, 16)
    ref = O.cull_scene(O.View(g), t, ml, tests=1)
    rs = np.random.RandomState(2)
    pick = np.sort(rs.choice(len(ref.list), 260, replace=False))
    lst = ref.list[pick]
    want = O.rasterize_depth(kitten["vertices"], kitten["meshlet_data"], ml, t, lst, g, w, h)
    got = N.rasterize_depth(pos, kitten["meshlet_data"], ml, t["model"], [(int(a), int(b)) for a, b in lst],
                            g["view"][0], g["proj"][0], w, h)
    assert np.array_equal(got.view("<u4"), want.view("<u4"))
    assert (want > 0).mean() > 0.05
