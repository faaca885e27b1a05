"""CPU tests of the oracle itself (no GPU): known-answer cases derived by hand from the reference's
source, the committed golden fixtures, and the independent numpy restatement (oracle/np_oracle.py).

PARITY UNPINNED here: the reference has no tests or golden vectors for this path; these pin the
oracle against its own fixtures, a second restatement and hand-derived answers.  (What the
reference's screenshots can pin is in tests/test_reference_images.py.)
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ASSETS = "/root/reference/assets"
f32 = np.float32


@pytest.fixture(scope="module")
def NP():
    import oracle.np_oracle as m
    return m


# ------------------------------------------------------------------ K1: layout contract
def test_struct_layout_table(tmp_path):
    """sizeof / offsetof of every struct in include/mcull.h against the reference's GPU layouts
    (shaders/common.h:1-34, Engine.hpp:102-105)."""
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "mcull.h"
#define S(t) printf(#t " %zu\n", sizeof(t))
#define O(t, f) printf(#t "." #f " %zu\n", offsetof(t, f))
int main(void) {
  S(mcull_vertex); O(mcull_vertex, vx); O(mcull_vertex, nx); O(mcull_vertex, tu); O(mcull_vertex, tv);
  S(mcull_meshlet); O(mcull_meshlet, cone); O(mcull_meshlet, coneApex); O(mcull_meshlet, radius);
  O(mcull_meshlet, dataOffset); O(mcull_meshlet, triangleCount); O(mcull_meshlet, vertexCount); O(mcull_meshlet, padding);
  S(mcull_globals); O(mcull_globals, view); O(mcull_globals, proj); O(mcull_globals, frustum); O(mcull_globals, meshletCount);
  S(mcull_transform); S(mcull_mesh_tasks_cmd); O(mcull_mesh_tasks_cmd, taskCount); O(mcull_mesh_tasks_cmd, firstTask);
  S(mcull_draw_indirect); O(mcull_draw_indirect, commandMeshIndirect); S(mcull_visible); S(mcull_task_payload);
  O(mcull_task_payload, drawID); O(mcull_task_payload, taskCount);
  return 0; }''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.rsplit(" ", 1) for line in subprocess.check_output([str(exe)], text=True).strip().splitlines())
    want = {
        "mcull_vertex": 24, "mcull_vertex.vx": 0, "mcull_vertex.nx": 16, "mcull_vertex.tu": 20, "mcull_vertex.tv": 22,
        "mcull_meshlet": 48, "mcull_meshlet.cone": 0, "mcull_meshlet.coneApex": 16, "mcull_meshlet.radius": 32,
        "mcull_meshlet.dataOffset": 36, "mcull_meshlet.triangleCount": 40, "mcull_meshlet.vertexCount": 41,
        "mcull_meshlet.padding": 42,
        "mcull_globals": 212, "mcull_globals.view": 0, "mcull_globals.proj": 64, "mcull_globals.frustum": 128,
        "mcull_globals.meshletCount": 208,
        "mcull_transform": 64, "mcull_mesh_tasks_cmd": 8, "mcull_mesh_tasks_cmd.taskCount": 0,
        "mcull_mesh_tasks_cmd.firstTask": 4, "mcull_draw_indirect": 28, "mcull_draw_indirect.commandMeshIndirect": 20,
        "mcull_visible": 8, "mcull_task_payload": 136, "mcull_task_payload.drawID": 128, "mcull_task_payload.taskCount": 132,
    }
    assert {k: int(v) for k, v in got.items()} == want


def test_numpy_dtypes_match_layout(O, mc):
    for mod in (O, mc):
        assert mod.VERTEX.itemsize == 24 and mod.MESHLET.itemsize == 48 and mod.GLOBALS.itemsize == 212
        assert mod.TRANSFORM.itemsize == 64 and mod.CMD.itemsize == 8 and mod.DRAW_INDIRECT.itemsize == 28
        assert mod.MESHLET.fields["radius"][1] == 32 and mod.MESHLET.fields["dataOffset"][1] == 36
        assert mod.GLOBALS.fields["frustum"][1] == 128 and mod.GLOBALS.fields["meshletCount"][1] == 208
        assert mod.DRAW_INDIRECT.fields["commandMeshIndirect"][1] == 20


# ------------------------------------------------------------------ helpers
IDENT = np.eye(4, dtype=f32).reshape(16)


def simple_view(O, planes, cam=(0.0, 0.0, 0.0)):
    """A view whose frustum planes are given directly; view = translation by -cam so that
    inverse(view)[3] = cam."""
    g = np.zeros(1, dtype=O.GLOBALS)
    v = np.eye(4, dtype=f32)
    v[3, :3] = [-cam[0], -cam[1], -cam[2]]
    g["view"][0] = v.reshape(16)
    g["proj"][0] = IDENT
    fr = np.zeros((5, 4), dtype=f32)
    fr[:, 3] = 1e6  # far away: always inside
    for i, p in enumerate(planes):
        fr[i] = p
    g["frustum"][0] = fr
    return O.View(g)


# ------------------------------------------------------------------ K2: frustum planes
@pytest.mark.parametrize("plane_index", range(5))
def test_frustum_tangent_inside_outside(O, plane_index):
    """shader.task:51: culled iff dot(n, c) + d + r < 0.  Plane x >= 0, sphere radius 1:
    centre x = -1 is exactly tangent (sum == 0 -> kept), -1 + 1e-4 kept, -1 - 1e-4 culled."""
    planes = [[0, 0, 0, 1e6]] * 5
    planes[plane_index] = [1, 0, 0, 0]
    v = simple_view(O, planes)
    cone = [0, 0, 1, 1]
    for x, culled in ((-1.0, 0), (-1.0 + 1e-4, 0), (-1.0 - 1e-4, 1), (5.0, 0), (-50.0, 1)):
        r = O.cull_pair(v, IDENT, cone, [x, 0, 0], 1.0, tests=O.TEST_FRUSTUM)
        assert r.frustum_culled == culled and r.visible == (not culled)
        assert r.frustum_margin == f32(f32(f32(x) + f32(0.0)) + f32(1.0))


def test_frustum_uses_world_space_and_scaled_radius(O):
    """shader.task:91,99: centre goes through the model matrix, radius is scaled by |model[0].xyz|."""
    v = simple_view(O, [[1, 0, 0, 0]])
    m = np.eye(4, dtype=f32) * f32(2.0)
    m[3] = [-10, 0, 0, 1]  # translate x by -10, uniform scale 2
    r = O.cull_pair(v, m.reshape(16), [0, 0, 1, 1], [3, 0, 0], 1.5, tests=O.TEST_FRUSTUM)
    # world centre x = 2*3 - 10 = -4, world radius = 1.5 * 2 = 3 -> -4 + 3 = -1 < 0
    assert r.frustum_culled == 1 and r.frustum_margin == f32(-1.0)
    r = O.cull_pair(v, m.reshape(16), [0, 0, 1, 1], [3, 0, 0], 2.0, tests=O.TEST_FRUSTUM)
    assert r.frustum_culled == 0 and r.frustum_margin == f32(0.0)


# ------------------------------------------------------------------ K3: backface cone
def test_cone_cull_cases(O):
    """shader.task:37-41: culled iff dot(axis, normalize(cam - centre)) < -coneW."""
    cam = (0.0, 0.0, 10.0)
    v = simple_view(O, [], cam=cam)
    centre = [0, 0, 0]
    # view direction = +z.  axis = -z: dot = -1.
    for cone_w, culled in ((0.0, 1), (0.5, 1), (1.0, 0)):      # -1 < -w ?
        r = O.cull_pair(v, IDENT, [0, 0, -1, cone_w], centre, 0.1, tests=O.TEST_CONE)
        assert r.cone_culled == culled, cone_w
        assert r.cone_margin == f32(-1.0 + cone_w)
    # axis = +z (facing the camera): dot = +1, never culled
    for cone_w in (0.0, 0.5, 1.0):
        assert O.cull_pair(v, IDENT, [0, 0, 1, cone_w], centre, 0.1, tests=O.TEST_CONE).cone_culled == 0
    # boundary angle: axis at 120 degrees from the view direction, dot = -0.5
    ax = [np.sin(np.radians(120.0)), 0, np.cos(np.radians(120.0))]
    assert O.cull_pair(v, IDENT, ax + [0.5 - 1e-4], centre, 0.1, tests=O.TEST_CONE).cone_culled == 1
    assert O.cull_pair(v, IDENT, ax + [0.5 + 1e-4], centre, 0.1, tests=O.TEST_CONE).cone_culled == 0
    # NaN axis (degenerate meshlet, Engine.cpp:493): comparison with NaN is false -> never culled
    assert O.cull_pair(v, IDENT, [np.nan, np.nan, np.nan, 0.0], centre, 0.1, tests=O.TEST_CONE).cone_culled == 0


def test_camera_position_is_inverse_view_column(O, NP):
    """shader.task:88-89 for the reference camera: eye (2,2,2)."""
    g = O.create_globals()
    v = O.View(g)
    assert np.allclose(v.cam_pos, [2, 2, 2], atol=1e-6)
    assert np.array_equal(v.cam_pos, NP.camera_position(g["view"][0]))
    assert np.array_equal(v.vp, NP.mat_mul(g["proj"][0], g["view"][0]))


# ------------------------------------------------------------------ K4: Hi-Z pyramid
def test_hiz_levels_and_layout(O, mc):
    for w, h, levels in ((800, 600, 10), (3840, 2160, 12), (4096, 4096, 13), (7680, 4320, 13), (8192, 8192, 14), (1, 1, 1), (5, 3, 3)):
        assert O.hiz_layout(w, h)[0] == levels == mc.hiz_layout(w, h)[0]
        assert O.hiz_layout(w, h) == mc.hiz_layout(w, h)
    n, lw, lh, off, total = O.hiz_layout(800, 600)
    assert lw == [800, 400, 200, 100, 50, 25, 12, 6, 3, 1] and lh == [600, 300, 150, 75, 37, 18, 9, 4, 2, 1]
    assert total == sum(a * b for a, b in zip(lw, lh))


def test_hiz_counter_pattern_closed_form(O):
    """Level k texel = min over its 2^k x 2^k block of level 0; odd remainders are ignored."""
    for h, w in ((4, 4), (3, 5), (600, 800), (37, 129)):
        d = (np.arange(h * w, dtype=f32).reshape(h, w) * f32(0.001) + f32(0.25))[::-1, ::-1].copy()
        lv = O.hiz_levels(O.hiz_build(d), w, h)
        assert np.array_equal(lv[0], d)
        for k in range(1, len(lv)):
            lh, lw = max(1, h >> k), max(1, w >> k)
            assert lv[k].shape == (lh, lw)
            if (h >> k) and (w >> k):  # no collapsed dimension: pure block minimum
                blk = d[: lh << k, : lw << k].reshape(lh, 1 << k, lw, 1 << k)
                assert np.array_equal(lv[k], blk.min(axis=(1, 3)))
    # 5x3 by hand: level 1 is 2x1 (column 4 and row 2 ignored), level 2 is 1x1
    d = np.array([[9, 8, 7, 6, 0], [5, 4, 3, 2, 0], [0, 0, 0, 0, 0]], dtype=f32)
    lv = O.hiz_levels(O.hiz_build(d), 5, 3)
    assert lv[1].tolist() == [[4.0, 2.0]] and lv[2].tolist() == [[2.0]]


def test_hiz_matches_numpy_restatement(O, NP):
    rs = np.random.RandomState(3)
    for h, w in ((600, 800), (65, 127), (768, 1024), (1, 300), (2, 2)):
        d = rs.uniform(0, 1, (h, w)).astype(f32)
        d.flat[rs.randint(0, h * w, 3)] = [np.nan, -0.0, np.inf][: min(3, h * w)]
        a = O.hiz_levels(O.hiz_build(d), w, h)
        b = NP.hiz_levels(d)
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert x.view("<u4").tobytes() == y.view("<u4").tobytes()
    d = rs.uniform(0, 1, (601, 803)).astype(f32)
    assert O.hiz_build(d, threads=4).tobytes() == O.hiz_build(d, threads=1).tobytes()


# ------------------------------------------------------------------ K5: occlusion
def test_mip_level_exact(O, NP):
    """shader.task:71-72 evaluated exactly: powers of two, their float neighbours, NaN, inf."""
    xs = [0.0, 0.5, 1.0, np.nextafter(f32(1), f32(2)), 2.0, np.nextafter(f32(2), f32(3)), 3.0, 4.0,
          np.nextafter(f32(4), f32(0)), 1023.9, 1024.0, 1024.1, 1e30, np.inf, -5.0, np.nan]
    want = [0, 0, 0, 1, 1, 2, 2, 2, 2, 10, 10, 11, 11, 11, 0, 0]
    assert [O.mip_level(x, 12) for x in xs] == want
    assert NP.mip_level(np.array(xs, dtype=f32), 12).tolist() == want
    assert O.mip_level(1e30, 3) == 2


def test_occlusion_constant_depth_and_bias(O):
    """shader.task:74-76: culled iff ndc.z < depth - 0.05 (reverse-Z: larger = nearer)."""
    w, h = 64, 32
    g = O.create_globals(eye=(0, -10, 0), centre=(0, 0, 0), up=(0, 0, 1), width=w, height=h)
    for depth_value in (0.0, 0.02, 0.3):
        pyr = O.hiz_build(np.full((h, w), depth_value, dtype=f32))
        v = O.View(g, pyr, w, h)
        # point at distance t in front of the eye: clip.w = t, clip.z = near = 0.1 -> ndc.z = 0.1 / t
        for t in (0.5, 2.0, 5.0, 20.0):
            r = O.cull_pair(v, IDENT, [0, 0, 1, 1], [0, -10 + t, 0], 0.01, tests=O.TEST_HIZ)
            ndc_z = f32(0.1) / f32(t)
            assert r.occlusion_culled == int(ndc_z < f32(depth_value) - f32(0.05))
            assert abs(r.occlusion_margin - (ndc_z - (f32(depth_value) - f32(0.05)))) < 1e-6


def test_occlusion_repeat_wrap_and_flip(O):
    """Sampler state Engine.cpp:1623-1627 (nearest, REPEAT) and uv.y = 1 - uv.y (shader.task:66)."""
    w, h = 8, 8
    d = np.arange(64, dtype=f32).reshape(8, 8) / f32(64)
    g = O.create_globals(eye=(0, -10, 0), centre=(0, 0, 0), up=(0, 0, 1), fov_deg=90.0, width=w, height=h, flip_y=False)
    v = O.View(g, O.hiz_build(d), w, h)
    # on the optical axis: ndc = (0,0) -> uv = (0.5, 0.5) -> texel (4, 4) of level 0 (tiny radius)
    r = O.cull_pair(v, IDENT, [0, 0, 1, 1], [0, 0, 0], 1e-4, tests=O.TEST_HIZ)
    assert (r.mip, r.tx, r.ty) == (0, 4, 4)
    # ndc.x = 1.5 (outside the viewport) wraps: uv.x = 1.25 -> floor(10) mod 8 = 2
    r = O.cull_pair(v, IDENT, [0, 0, 1, 1], [15.0, 0, 0], 1e-4, tests=O.TEST_HIZ)
    assert (r.mip, r.tx) == (0, 2)
    # negative wrap: ndc.x = -1.5 -> uv.x = -0.25 -> floor(-2) mod 8 = 6
    r = O.cull_pair(v, IDENT, [0, 0, 1, 1], [-15.0, 0, 0], 1e-4, tests=O.TEST_HIZ)
    assert r.tx == 6
    # world +z is up; unflipped projection maps it to ndc.y > 0 -> uv.y = 1 - (0.5*ndc.y + 0.5) < 0.5
    r = O.cull_pair(v, IDENT, [0, 0, 1, 1], [0, 0, 5.0], 1e-4, tests=O.TEST_HIZ)
    assert r.ty == 2  # ndc.y = 0.5 -> uv.y = 0.25 -> texel row 2


def test_mip_selection_flipped_vs_unflipped(O):
    """With the reference's proj[1][1] < 0 (Engine.cpp:2335) ndcRadius is negative in front of the
    camera, so the level is always 0; with an unflipped projection the level grows with the radius."""
    w, h = 1024, 512
    pyr = O.hiz_build(np.zeros((h, w), dtype=f32))
    for flip, expect in ((True, [0, 0, 0]), (False, [0, 6, 10])):
        g = O.create_globals(eye=(0, -10, 0), centre=(0, 0, 0), up=(0, 0, 1), fov_deg=90.0, width=w, height=h, flip_y=flip)
        v = O.View(g, pyr, w, h)
        got = [O.cull_pair(v, IDENT, [0, 0, 1, 1], [0, 0, 0], rad, tests=O.TEST_HIZ).mip for rad in (1e-4, 0.25, 50.0)]
        # unflipped: px = (1 * r / 10) * 2 * 1024: r=0.25 -> 51.2 -> ceil(log2) = 6; r=50 -> clamps to level 10
        assert got == expect


# ------------------------------------------------------------------ K6: bounds + cone by hand
def test_flat_quad_meshlet_by_hand(O, NP):
    """Two coplanar triangles in z = 0: axis (0,0,1), min dot 1 -> coneW = sqrt(1-1) = 0,
    apex = centroid of the 4 vertices, radius = half diagonal."""
    pos = np.array([[0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0]], dtype=f32)
    corners = np.array([[0, -1, -1], [1, -1, -1], [2, -1, -1], [0, -1, -1], [2, -1, -1], [3, -1, -1]], dtype="<i4")
    m = O.build_mesh(pos, corners, optimize=False)
    assert len(m.meshlets) == 1
    ml = m.meshlets[0]
    assert ml["vertexCount"] == 4 and ml["triangleCount"] == 2 and ml["dataOffset"] == 0
    assert ml["cone"].tolist() == [0.0, 0.0, 1.0, 0.0]
    assert ml["coneApex"].tolist() == [1.0, 1.0, 0.0, 0.0]
    assert ml["radius"] == np.sqrt(f32(2.0))
    # meshletData = [4 global ids][6 local ids packed 4 per word, zero padded] (Engine.cpp:577-584)
    assert m.meshlet_data.tolist() == [0, 1, 2, 3, 0x00020100, 0x00000302]
    cone, apex, radius = NP.meshlet_cone(pos, m.meshlet_data, 0, 4, 2)
    assert cone.tolist() == ml["cone"].tolist() and apex.tolist() == ml["coneApex"][:3].tolist() and radius == ml["radius"]


def test_tetrahedron_cones(O):
    """Face normals of a regular tetrahedron have pairwise dot -1/3.
    Three faces: axis = minus the missing face's normal, every dot = 1/3 -> coneW = sqrt(1 - 1/9).
    All four faces: the normals cancel, normalize(0) = NaN axis, NaN dots are ignored -> coneW = 0."""
    pos = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], dtype=f32)
    tris = [(0, 1, 2), (0, 3, 1), (0, 2, 3), (1, 3, 2)]
    corners = np.array([[v, -1, -1] for t in tris[:3] for v in t], dtype="<i4")
    ml = O.build_mesh(pos, corners, optimize=False).meshlets[0]
    assert ml["triangleCount"] == 3 and ml["vertexCount"] == 4
    assert abs(ml["cone"][3] - np.sqrt(8.0) / 3.0) < 1e-6
    n4 = np.cross(pos[3] - pos[1], pos[2] - pos[1]); n4 /= np.linalg.norm(n4)
    assert np.allclose(ml["cone"][:3], -n4, atol=1e-6)
    assert np.allclose(ml["coneApex"][:3], 0, atol=1e-7) and np.isclose(ml["radius"], np.sqrt(3.0))
    corners = np.array([[v, -1, -1] for t in tris for v in t], dtype="<i4")
    ml = O.build_mesh(pos, corners, optimize=False).meshlets[0]
    assert np.isnan(ml["cone"][:3]).all() and ml["cone"][3] == 0.0


def test_cone_wider_than_a_hemisphere_is_disabled(O):
    """min dot < 0 -> coneW = 1.0, which the cone test can never beat (Engine.cpp:526)."""
    a = np.radians(170.0)
    # two triangles with normal +z, one folded back by 170 degrees about the x axis
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, np.cos(a), np.sin(a)]], dtype=f32)
    tris = [(0, 1, 2), (1, 3, 2), (0, 1, 4)]
    corners = np.array([[v, -1, -1] for t in tris for v in t], dtype="<i4")
    ml = O.build_mesh(pos, corners, optimize=False).meshlets[0]
    assert ml["cone"][3] == 1.0 and ml["cone"][2] > 0.9


def test_degenerate_triangle_gives_nan_axis(O):
    """Zero-area triangle -> normalize(0) = NaN normal -> NaN axis; std::min ignores the NaN dot so
    halfAngle stays 1 and coneW = 0 (Engine.cpp:493, :509-513, :526)."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0]], dtype=f32)
    ml = O.build_mesh(pos, np.array([[0, -1, -1], [1, -1, -1], [2, -1, -1]], dtype="<i4"), optimize=False).meshlets[0]
    assert np.isnan(ml["cone"][:3]).all() and ml["cone"][3] == 0.0


def test_meshlet_limits_and_flush(O):
    """Engine.cpp:572: flush when vertices would exceed 64 or triangles reach 124."""
    # a fan around vertex 0: triangle k = (0, k+1, k+2): 2 new vertices then 1 per triangle
    n = 300
    ang = np.linspace(0, 6, n + 2)
    pos = np.concatenate([[[0, 0, 0]], np.stack([np.cos(ang), np.sin(ang), 0 * ang], 1)]).astype(f32)
    corners = np.array([[v, -1, -1] for k in range(n) for v in (0, k + 1, k + 2)], dtype="<i4")
    m = O.build_mesh(pos, corners, optimize=False)
    assert m.meshlets["vertexCount"].max() == 64 and m.meshlets["triangleCount"].max() == 62
    assert m.meshlets["triangleCount"].astype(int).sum() == n
    # 124-triangle limit: the same 3 vertices repeated
    corners = np.array([[v, -1, -1] for k in range(300) for v in (0, 1, 2)], dtype="<i4")
    m = O.build_mesh(pos, corners, optimize=False)
    assert m.meshlets["triangleCount"].tolist() == [124, 124, 52] and m.meshlets["vertexCount"].tolist() == [3, 3, 3]
    offs = m.meshlets["dataOffset"].tolist()
    assert offs == [0, 3 + 93, 2 * (3 + 93)]


# ------------------------------------------------------------------ loader pieces
def test_float_to_half_matches_ieee(O):
    rs = np.random.RandomState(0)
    xs = np.concatenate([rs.uniform(-2, 2, 2000), rs.uniform(-70000, 70000, 500), [0.0, -0.0, 1.0, 65504.0, 65519.9, 65520.0,
                         6e-8, 5.96e-8, 2.98e-8, 2.99e-8, 6.1e-5, np.inf, -np.inf]]).astype(f32)
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([O.lib().orc_float_to_half(float(x)) for x in xs], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_vertex_score(O):
    """computeVertexScore (Engine.cpp:308-335)."""
    s = O.lib().orc_vertex_score
    assert s(5, 0) == -1.0
    assert s(-1, 4) == 1.0                       # 0 + 2 * 4^-0.5
    assert s(0, 1) == s(2, 1) == 2.75            # 0.75 + 2
    assert abs(s(31, 1) - (2.0 + (1 / 29) ** 1.5)) < 1e-6   # oldest cache slot: (1 - 28/29)^1.5
    assert abs(s(3, 1) - 3.0) < 1e-6
    assert abs(s(17, 16) - (np.power(1 - 14 / 29, 1.5) + 0.5)) < 1e-6


def test_obj_loader_dedup_and_quantisation(O, tmp_path):
    """loadModel (Engine.cpp:252-281): dedup key (pos, u8 normal, half uv), first-seen order,
    normal (n*0.5+0.5)*255 truncated, v flipped."""
    p = tmp_path / "t.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nvn 0 0 2\nvt 0.25 0.75\nvt 1 0\n"
                 "f 1/1/1 2/1/1 3/1/1\nf 2/1/2 4/2/1 3/1/1\nf -4/1/1 -3/1/1 -2/1/1\n")
    m = O.load_obj(str(p), optimize=False, meshletize=False)
    # vn 0 0 2 normalises to the same u8 normal -> same vertex; -4.. are the same vertices again
    assert len(m.vertices) == 4 and m.indices.tolist() == [0, 1, 2, 1, 3, 2, 0, 1, 2]
    v = m.vertices[0]
    assert v["n"].tolist() == [127, 127, 255, 0]                  # (0*.5+.5)*255 = 127.5 -> 127
    assert v["uv"].tolist() == [int(np.float16(0.25).view(np.uint16)), int(np.float16(0.25).view(np.uint16))]
    assert m.vertices[3]["uv"].tolist() == [int(np.float16(1.0).view(np.uint16)), int(np.float16(1.0).view(np.uint16))]
    # no normals at all: normalize(0) = NaN -> 0 (bunny.obj has no vn records)
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    assert O.load_obj(str(p), optimize=False, meshletize=False).vertices["n"].tolist() == [[0, 0, 0, 0]] * 3


def test_forsyth_is_a_permutation_and_deterministic(O):
    rs = np.random.RandomState(5)
    n = 40
    pos = rs.uniform(0, 1, (n * n, 3)).astype(f32)
    tris = []
    for y in range(n - 1):
        for x in range(n - 1):
            a = y * n + x
            tris += [(a, a + 1, a + n), (a + 1, a + n + 1, a + n)]
    corners = np.array([[v, -1, -1] for t in tris for v in t], dtype="<i4")
    raw = O.build_mesh(pos, corners, optimize=False, meshletize=False)
    opt = O.build_mesh(pos, corners, optimize=True, meshletize=False)
    key = lambda ind: sorted(map(tuple, ind.reshape(-1, 3).tolist()))
    assert key(raw.indices) == key(opt.indices) and not np.array_equal(raw.indices, opt.indices)
    assert np.array_equal(opt.indices, O.build_mesh(pos, corners, optimize=True, meshletize=False).indices)
    # the first triangle is the FIRST maximum of the initial scores: the corner triangle 0, whose
    # vertex 0 has valence 1 (score 2 + 2/sqrt(3) + 2/sqrt(3)); later ties do not replace it (Engine.cpp:404)
    assert opt.indices[:3].tolist() == raw.indices[:3].tolist()


# ------------------------------------------------------------------ K7: assets end to end
@pytest.mark.parametrize("name", ["kitten", "bunny", "viking_room"])
def test_asset_pipeline_matches_manifest(O, manifest, name):
    path = os.path.join(REF_ASSETS, name + ".obj")
    if not os.path.exists(path):
        pytest.skip("reference assets are not available on this machine")
    want = manifest[name]
    m = O.load_obj(path)
    assert len(m.vertices) == want["vertices"] and len(m.indices) // 3 == want["triangles"]
    assert len(m.meshlets) == want["meshlets"] and len(m.meshlet_data) == want["meshlet_data_words"]
    assert "%016x" % O.fnv1a64(m.vertices) == want["fnv_vertices"]
    assert "%016x" % O.fnv1a64(m.indices) == want["fnv_indices_forsyth"]
    assert "%016x" % O.fnv1a64(m.meshlet_data) == want["fnv_meshlet_data"]
    assert "%016x" % O.fnv1a64(m.meshlets) == want["fnv_meshlets"]
    assert int(m.meshlets["triangleCount"].astype(np.int64).sum()) == want["triangles"]
    assert m.meshlets["vertexCount"].max() <= 64 and m.meshlets["triangleCount"].max() <= 124


@pytest.mark.parametrize("name", ["kitten", "bunny", "viking_room"])
def test_scene_fixture_consistency(O, NP, manifest, name):
    """The committed scene fixtures: hashes, invariants, and bounds recomputed two ways."""
    from conftest import load_scene
    sc = load_scene(name, O)
    want = manifest[name]
    assert "%016x" % O.fnv1a64(sc["meshlets"]) == want["fnv_meshlets"]
    assert "%016x" % O.fnv1a64(sc["meshlet_data"]) == want["fnv_meshlet_data"]
    assert "%016x" % O.fnv1a64(sc["vertices"]) == want["fnv_vertices"]
    ml = sc["meshlets"]
    words = ml["vertexCount"].astype(np.int64) + (ml["triangleCount"].astype(np.int64) * 3 + 3) // 4
    assert np.array_equal(ml["dataOffset"], np.concatenate([[0], np.cumsum(words)[:-1]]))
    assert int(words.sum()) == len(sc["meshlet_data"])
    pos = sc["vertices"]["pos"][:, :3]
    for i in list(range(0, len(ml), max(1, len(ml) // 40))) + [len(ml) - 1]:
        again = O.build_meshlet_cone(sc["vertices"], sc["meshlet_data"], ml[i])
        assert again.tobytes() == ml[i].tobytes()
        cone, apex, radius = NP.meshlet_cone(pos, sc["meshlet_data"], int(ml[i]["dataOffset"]), int(ml[i]["vertexCount"]),
                                             int(ml[i]["triangleCount"]))
        assert cone.tobytes() == ml[i]["cone"].tobytes() and apex.tobytes() == ml[i]["coneApex"][:3].tobytes()
        assert radius == ml[i]["radius"]


# ------------------------------------------------------------------ K8: config 1
def test_config1_inputs_reproduced(O, H, config1, manifest):
    """createGlobals / updateTransforms restated twice (C with glibc rand, numpy with its own
    TYPE_3 generator) against the committed inputs."""
    g, t = config1["globals"], config1["transforms"]
    assert "%016x" % O.fnv1a64(g) == manifest["config1"]["fnv_globals"]
    assert O.create_globals(meshlet_count=326).tobytes() == g.tobytes()
    assert H.create_globals(meshlet_count=326).tobytes() == g.tobytes()
    assert O.update_transforms(1000).tobytes() == t.tobytes()
    th = H.update_transforms(1000)
    # numpy's float32 sin/cos may differ from glibc's in the last place; the rand() stream must not
    assert np.abs(th["model"] - t["model"]).max() <= 2.4e-7
    assert np.array_equal(th["model"][:, 12:15], t["model"][:, 12:15])  # translations come straight from rand()
    d = O.create_indirect_buffer(1000, 326, 28944 * 3)
    assert d.tobytes() == H.create_indirect_buffer(1000, 326, 28944 * 3).tobytes()
    assert d["commandMeshIndirect"]["taskCount"][0] == 11 and d["commandMeshIndirect"]["firstTask"].max() == 0


def test_library_host_inputs_equal_golden(O, mc, config1):
    """The library's own C++ createGlobals / updateTransforms / createIndirectBuffer (host code, no
    GPU) against the committed config-1 inputs and, for other cameras and seeds, against the oracle."""
    g, t = config1["globals"], config1["transforms"]
    assert mc.create_globals(800, 600, 326).tobytes() == g.tobytes()
    assert mc.reference_transforms(1000, 42).tobytes() == t.tobytes()
    assert mc.create_indirect_buffer(1000, 326, 28944 * 3).tobytes() == O.create_indirect_buffer(1000, 326, 28944 * 3).tobytes()
    for eye, centre, up, fov, zn, w, h, flip in [((0, -10, 0), (0, 0, 0), (0, 0, 1), 90.0, 0.1, 64, 48, False),
                                                 ((51.0, 54.4, 40.8), (1, 2, 3), (0, 1, 0), 33.0, 0.5, 3840, 2160, True),
                                                 ((-3, 4, -5), (0.5, 0.25, 8), (1, 1, 0), 120.0, 0.01, 8192, 8192, True)]:
        a = mc.create_globals(w, h, 7, eye=eye, centre=centre, up=up, fov_deg=fov, z_near=zn, flip_y=flip)
        b = O.create_globals(eye=eye, centre=centre, up=up, fov_deg=fov, z_near=zn, width=w, height=h, flip_y=flip, meshlet_count=7)
        assert a.tobytes() == b.tobytes()
        a["frustum"] = 0
        assert mc.extract_frustum(a).tobytes() == b.tobytes()
    # the private generator leaves the process-wide rand() alone and restarts from the seed each call
    assert mc.reference_transforms(10, 42).tobytes() == t[:10].tobytes()
    assert mc.reference_transforms(10, 43).tobytes() != t[:10].tobytes()
    libc = C.CDLL(None)
    libc.srand(5)
    first = libc.rand()
    libc.srand(5)
    mc.reference_transforms(3, 42)
    assert libc.rand() == first
    assert mc.reference_transforms(0, 42).shape == (0,)
    # a degenerate camera (eye == centre) gives NaNs, as the reference would, not an error
    assert np.isnan(mc.create_globals(8, 8, 0, eye=(1, 1, 1), centre=(1, 1, 1))["view"]).any()
    with pytest.raises(mc.McullError):
        mc.create_globals(0, 600, 1)


def test_config1_results_match_manifest_and_numpy(O, NP, kitten, config1, manifest):
    g, t = config1["globals"], config1["transforms"]
    ml = kitten["meshlets"]
    pyr = O.hiz_build(np.zeros((600, 800), dtype=f32))
    view = O.View(g, pyr, 800, 600)
    res = {}
    for label, tests in (("all", 7), ("frustum_cone", 3), ("frustum", 1), ("cone", 2)):
        r = O.cull_scene(view, t, ml, tests=tests, want_margins=True)
        want = manifest["config1"]["results"][label]
        assert r.total == want["total"] and r.culled == want["culled"]
        assert "%016x" % O.fnv1a64(r.list) == want["fnv_list"]
        assert "%016x" % O.fnv1a64(r.mask) == want["fnv_mask"]
        assert "%016x" % O.fnv1a64(r.commands) == want["fnv_commands"]
        assert int(r.commands["taskCount"].astype(np.int64).sum()) == r.total
        assert np.array_equal(r.commands["firstTask"], np.concatenate([[0], np.cumsum(r.commands["taskCount"])[:-1]]))
        res[label] = r
        # multi-threaded run is identical
        r8 = O.cull_scene(view, t, ml, tests=tests, threads=8)
        assert r8.list.tobytes() == r.list.tobytes() and np.array_equal(r8.mask, r.mask)
    # sampled rows with margins
    z = np.load(os.path.join(ROOT, "tests", "golden", "config1_rows.npz"))
    assert np.array_equal(res["all"].margins[z["rows"]].view("<u4"), z["margins"].view("<u4"))
    # independent numpy restatement: every pair of every instance, with the Hi-Z test on
    cone = ml["cone"]
    sphere = np.concatenate([ml["coneApex"][:, :3], ml["radius"][:, None]], axis=1)
    lv = NP.hiz_levels(np.zeros((600, 800), dtype=f32))
    vis_bits = np.unpackbits(res["all"].mask.view("u1"), bitorder="little").reshape(1000, -1)[:, : len(ml)]
    for i in range(0, 1000, 7):
        vis, mg = NP.cull_instance(g, t["model"][i], cone, sphere, tests=7, pyramid=lv)
        assert np.array_equal(vis, vis_bits[i].astype(bool)), i
        m = res["all"].margins.reshape(1000, len(ml), 3)[i]
        assert np.array_equal(mg["frustum"].view("<u4"), m[:, 0].view("<u4"))
        assert np.array_equal(mg["cone"].view("<u4"), m[:, 2].view("<u4"))
        assert np.array_equal(mg["occlusion"].view("<u4"), m[:, 1].view("<u4"))


def test_hiz_cull_matches_numpy_with_real_occluders(O, NP, H, kitten, config1):
    g, t = config1["globals"], config1["transforms"][:200]
    ml = kitten["meshlets"]
    depth = H.synthetic_depth(800, 600, n_discs=24, seed=5)
    for flip in (True, False):
        gg = O.create_globals(meshlet_count=len(ml), flip_y=flip)
        view = O.View(gg, O.hiz_build(depth), 800, 600)
        r = O.cull_scene(view, t, ml, tests=7)
        lv = NP.hiz_levels(depth)
        cone = ml["cone"]
        sphere = np.concatenate([ml["coneApex"][:, :3], ml["radius"][:, None]], axis=1)
        bits = np.unpackbits(r.mask.view("u1"), bitorder="little").reshape(len(t), -1)[:, : len(ml)]
        occluded_somewhere = False
        for i in range(0, len(t), 3):
            vis, mg = NP.cull_instance(gg, t["model"][i], cone, sphere, tests=7, pyramid=lv)
            assert np.array_equal(vis, bits[i].astype(bool))
            occluded_somewhere |= bool((mg["culled"][1] & ~mg["culled"][0]).any())
        assert occluded_somewhere


def test_task_payloads_are_the_list(O, kitten, config1):
    """The union of the reference's per-32 payloads (shader.task:103-118) is the global list."""
    g, t = config1["globals"], config1["transforms"][:50]
    view = O.View(g)
    p = O.task_payloads(view, t, kitten["meshlets"], tests=3)
    r = O.cull_scene(view, t, kitten["meshlets"], tests=3)
    assert len(p) == 50 * 11
    flat = [(int(q["drawID"]), int(m)) for q in p for m in q["meshletIndices"][: q["taskCount"]]]
    assert flat == [(int(a), int(b)) for a, b in r.list.tolist()]
    assert all((q["meshletIndices"][q["taskCount"]:] == 0).all() for q in p)


def test_ranges_and_capacity(O, kitten, H):
    ml = kitten["meshlets"]
    t = H.update_transforms(20)
    g = H.create_globals(meshlet_count=len(ml))
    view = O.View(g)
    off = np.array([0, 10, 0, 326 - 65] * 5, dtype="<u4")
    cnt = np.array([326, 64, 0, 65] * 5, dtype="<u4")
    r = O.cull_scene(view, t, ml, ranges=(off, cnt), tests=2)
    assert r.n_pairs == int(cnt.sum()) and r.n_groups == int(((cnt + 63) // 64).sum())
    assert set(r.list["drawID"].tolist()) <= {i for i in range(20) if cnt[i]}
    for i in range(20):
        ids = r.list["meshletID"][r.list["drawID"] == i]
        assert ((ids >= off[i]) & (ids < off[i] + cnt[i])).all()
        assert len(ids) == r.commands["taskCount"][i]
    small = O.cull_scene(view, t, ml, ranges=(off, cnt), tests=2, list_capacity=10)
    assert small.overflow and small.total == r.total
