"""One context per thread (the library's threading rule): several host threads, each with its own context and
stream, run whole frames at the same time on one GPU.  ctypes releases the interpreter lock during the
calls, so the library's host code and kernels really overlap; every thread's results must still equal the
oracle's.  Guards against state shared between contexts (static buffers, counters, scratch)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_contexts_on_concurrent_threads(mc, O, H, kitten, bunny):
    import torch

    scenes = []
    for k, mesh in enumerate((kitten, bunny, kitten, bunny)):
        n = 60 + 25 * k
        t = H.synthetic_transforms(n, extent=6.0 + k, seed=90 + k)
        w, h = (320, 240) if k % 2 == 0 else (513, 257)
        g = H.create_globals(eye=(9.0 + k, 8.0, 7.0 - k), width=w, height=h, meshlet_count=len(mesh["meshlets"]))
        ml, md, vt = mesh["meshlets"], mesh["meshlet_data"], mesh["vertices"]
        r1 = O.cull_scene(O.View(g), t, ml, tests=3)
        img = O.rasterize_depth(vt, md, ml, t, r1.list, g, w, h)
        r2 = O.cull_scene(O.View(g, O.hiz_build(img), w, h), t, ml, tests=7)
        scenes.append(dict(t=t, g=g, w=w, h=h, ml=ml, md=md, vt=vt, img=img, ref=r2))
    errors = []
    start = threading.Barrier(len(scenes))

    def worker(sc):
        try:
            stream = torch.cuda.Stream(device=0)
            with mc.Context(0, stream=stream.cuda_stream) as ctx:
                ctx.upload_meshlets(sc["ml"])
                ctx.upload_geometry(sc["vt"], sc["md"])
                ctx.upload_transforms(sc["t"])
                start.wait(timeout=60)
                for _ in range(25):
                    ctx.cull(sc["g"], tests=3)
                    ctx.depth_rasterize(sc["g"], sc["w"], sc["h"])
                    ptr, _, _ = ctx.depth_buffer()
                    ctx.hiz_build(ptr, view=1, width=sc["w"], height=sc["h"], mem=mc.MEM_DEVICE)
                    ctx.cull(sc["g"], tests=7, hiz_view=1)
                    assert ctx.download_count() == sc["ref"].total
                    assert ctx.download_list().tobytes() == sc["ref"].list.tobytes()
                assert ctx.depth_download().view("<u4").tobytes() == sc["img"].view("<u4").tobytes()
                cmds = ctx.download_commands(len(sc["t"])).view(mc.CMD).reshape(-1)
                assert np.array_equal(cmds["firstTask"], sc["ref"].commands["firstTask"])
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(sc,)) for sc in scenes]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=100)
    assert not any(th.is_alive() for th in threads), "a worker did not finish"
    assert not errors, errors
