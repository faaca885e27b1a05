#!/usr/bin/env python3
"""Turns the two kitten screenshots the reference ships (assets/mesh_pipeline_kittens.png and
assets/meshlets_mesh_pipeline_kittens.png, 796x599 captures of its 800x600 window) into small data
fixtures.  They are outputs of the reference itself - DRAW_COUNT = 1000 kittens placed by
updateTransforms, seen through createGlobals' camera, rasterised by the hardware - and pin, through
the image, what no test of the reference pins: the transforms, the camera and projection
conventions, the OBJ loader and, in the second capture, the meshlet partition and numbering
(every meshlet is painted with shader.mesh:51-58's colour of its index).

    python tests/golden/make_reference_image_fixtures.py [/root/reference]

Writes tests/golden/reference_screenshots.npz:
  size               (height, width) of the captures
  foreground_bits    np.packbits of "pixel is not the clear colour" in the shaded capture
  flat_yx, flat_rgb  position and colour of every pixel of the meshlet capture whose 5x5
                     neighbourhood has one single colour other than the clear colour (the interior
                     of a meshlet's footprint), i.e. pixels whose colour names one meshlet index
  shade_yx, shade_rgb  position and colour of every second pixel of the shaded capture that lies well inside the
                     kittens (5x5 erosion of the foreground) and whose 3x3 neighbourhood varies by at most 4 levels per
                     channel: places where the interpolated normal, which is what the fragment shader writes
                     (shader.frag:12), changes slowly, so that the colour does not depend on how a rasteriser snaps
                     sub-pixel positions
Only pixel data derived from the images is stored; nothing of the reference's source.
"""
import os
import sys

import numpy as np
from PIL import Image
from scipy import ndimage as ndi

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    shaded = np.asarray(Image.open(os.path.join(ref, "assets", "mesh_pipeline_kittens.png")).convert("RGB")).astype(np.int32)
    painted = np.asarray(Image.open(os.path.join(ref, "assets", "meshlets_mesh_pipeline_kittens.png")).convert("RGB")).astype(np.int32)
    assert shaded.shape == painted.shape == (599, 796, 3)
    clear = np.array([255, 188, 188])  # sRGB of the clear colour, the most frequent pixel value of both captures
    vals, counts = np.unique(shaded.reshape(-1, 3), axis=0, return_counts=True)
    assert tuple(vals[counts.argmax()]) == tuple(clear)
    foreground = np.abs(shaded - clear).sum(-1) > 24
    flat = np.ones(painted.shape[:2], dtype=bool)
    for c in range(3):
        flat &= ndi.maximum_filter(painted[..., c], size=5) == ndi.minimum_filter(painted[..., c], size=5)
    flat &= np.abs(painted - clear).sum(-1) > 24
    flat[:2, :] = False  # the capture's top rows are window border
    yx = np.argwhere(flat).astype(np.uint16)
    rgb = painted[flat].astype(np.uint8)
    smooth = ndi.binary_erosion(foreground, np.ones((5, 5), dtype=bool))
    for c in range(3):
        smooth &= (ndi.maximum_filter(shaded[..., c], size=3) - ndi.minimum_filter(shaded[..., c], size=3)) <= 4
    smooth[:2, :] = False
    shade_yx = np.argwhere(smooth)[::2].astype(np.uint16)
    shade_rgb = shaded[shade_yx[:, 0].astype(np.int64), shade_yx[:, 1].astype(np.int64)].astype(np.uint8)
    out = os.path.join(HERE, "reference_screenshots.npz")
    np.savez_compressed(out, size=np.array(shaded.shape[:2], dtype=np.uint16), foreground_bits=np.packbits(foreground),
                        flat_yx=yx, flat_rgb=rgb, shade_yx=shade_yx, shade_rgb=shade_rgb)
    print("foreground %.4f, flat pixels %d, distinct colours %d, shaded pixels %d -> %s (%d bytes)"
          % (foreground.mean(), len(yx), len(np.unique(rgb, axis=0)), len(shade_yx), out, os.path.getsize(out)))


if __name__ == "__main__":
    main()
