#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the reference's OBJ assets.

Run in the build container only (needs /root/reference/assets): the GPU box has no reference
checkout, so the meshletised scenes travel as data.  Everything here is produced by the CPU oracle
(oracle/mcull_oracle.c); PARITY UNPINNED — the reference holds no golden vectors of its own, so
these pin the oracle against regressions and against the independent numpy restatement
(oracle/np_oracle.py), not against a run of the reference.

    python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle as O  # noqa: E402

ASSETS = "/root/reference/assets"


def scene(name):
    m = O.load_obj(os.path.join(ASSETS, name + ".obj"))
    np.savez_compressed(
        os.path.join(HERE, name + "_scene.npz"),
        vertices=m.vertices.view("u1"), meshlet_data=m.meshlet_data, meshlets=m.meshlets.view("u1"),
    )
    return m


def main():
    manifest = {}
    for name in ("kitten", "bunny", "viking_room"):
        m = scene(name)
        raw = O.load_obj(os.path.join(ASSETS, name + ".obj"), optimize=False, meshletize=False)
        manifest[name] = dict(
            vertices=int(len(m.vertices)), triangles=int(len(m.indices) // 3), meshlets=int(len(m.meshlets)),
            meshlet_data_words=int(len(m.meshlet_data)),
            fnv_vertices="%016x" % O.fnv1a64(m.vertices), fnv_indices_loaded="%016x" % O.fnv1a64(raw.indices),
            fnv_indices_forsyth="%016x" % O.fnv1a64(m.indices), fnv_meshlet_data="%016x" % O.fnv1a64(m.meshlet_data),
            fnv_meshlets="%016x" % O.fnv1a64(m.meshlets),
            sum_triangle_count=int(m.meshlets["triangleCount"].astype(np.int64).sum()),
            max_vertex_count=int(m.meshlets["vertexCount"].max()), max_triangle_count=int(m.meshlets["triangleCount"].max()),
        )
        if name == "kitten":
            kitten = m
    # config 1: kitten, 1000 instances (srand(42) recipe), 800x600 camera of the reference
    g = O.create_globals(meshlet_count=len(kitten.meshlets))
    t = O.update_transforms(1000)
    depth0 = np.zeros((600, 800), dtype="<f4")  # cleared reverse-Z depth (Engine.cpp:2404)
    pyr0 = O.hiz_build(depth0)
    view = O.View(g, pyr0, 800, 600)
    out = {}
    for label, tests in (("all", 7), ("frustum_cone", 3), ("frustum", 1), ("cone", 2)):
        r = O.cull_scene(view, t, kitten.meshlets, tests=tests, want_margins=(label == "all"))
        out[label] = dict(total=r.total, culled=r.culled, fnv_list="%016x" % O.fnv1a64(r.list),
                          fnv_mask="%016x" % O.fnv1a64(r.mask), fnv_commands="%016x" % O.fnv1a64(r.commands))
        if label == "all":
            rs = np.random.RandomState(7)
            rows = np.sort(rs.choice(r.n_pairs, 256, replace=False))
            mc, gpi = len(kitten.meshlets), (len(kitten.meshlets) + 63) // 64
            bits = np.array([(int(r.mask[int(p) // mc * gpi + int(p) % mc // 64]) >> (int(p) % mc % 64)) & 1 for p in rows],
                            dtype="u1")
            np.savez_compressed(os.path.join(HERE, "config1_rows.npz"), rows=rows, margins=r.margins[rows], visible=bits)
    manifest["config1"] = dict(instances=1000, width=800, height=600, results=out,
                               fnv_globals="%016x" % O.fnv1a64(g), fnv_transforms="%016x" % O.fnv1a64(t))
    np.savez_compressed(os.path.join(HERE, "config1_inputs.npz"), globals=g.view("u1"), transforms=t.view("u1"))
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
