#!/usr/bin/env python3
"""Full-size spot-check vectors from the REAL reference (oracle/_ref/refprobe).
Build container only.

For each of the five BASELINE configs: the 20000 seeded pixels of
tests/golden/sparse_<config>.npz, their camera rays at FULL resolution (pixel
centre plus the MSAA offsets of main.cpp:452-458 where the config uses MSAA 4)
traced by the reference's own performRayTracing / checkHit, and the pixel the
frame driver makes of them (main.cpp:435-511: centre-gated samples, ordered
float sum / 4, depth = mean minT of the samples that hit; RGBValue clamp of
imageWriter.h:23-44).  The reference build here has no image library, so the
scenes are untextured (Material::hasTexture() false): the GPU test renders the
same untextured scene at full size and compares these pixels.

  python tools/make_sparse_ref.py   -> tests/golden/sparse_ref_<config>.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import oracle as O  # noqa: E402
from make_counters import CONFIGS, pixel_rays  # noqa: E402
from make_golden import run_ref, tri_ids  # noqa: E402

FLT_MAX = np.float32(3.4028234663852886e38)
OFFS4 = [(-1 / 8, -3 / 8), (3 / 8, -1 / 8), (-3 / 8, 1 / 8), (1 / 8, 3 / 8)]  # main.cpp:452-458
BG = np.float32([0.2, 0.5, 1.0])


def compose_pixels(s, sp, W, H, msaa, textured=False):
    """The sparse pixels of fixture `sp` from the reference's per-ray answers: main.cpp:435-511."""
    x, y, corners, lights = sp["x"], sp["y"], sp["corners"], sp["lights"]
    s.set_lights(lights)
    n = len(x)
    offs = [(0.0, 0.0)] + (OFFS4 if msaa == 4 else [])
    o = np.concatenate([pixel_rays(corners, W, H, x + ox, y + oy)[0] for ox, oy in offs])
    d = np.concatenate([pixel_rays(corners, W, H, x + ox, y + oy)[1] for ox, oy in offs])
    _, _, rays = run_ref(s, lights, o, d, textured=textured)
    tid = tri_ids(s, rays).reshape(len(offs), n)
    rgb = rays["rgb"].reshape(len(offs), n, 3)
    minT = rays["minT"].reshape(len(offs), n)
    hit = rays["bHit"].reshape(len(offs), n) != 0
    assert ((minT == FLT_MAX) == ~hit).all()
    if msaa == 1:
        px, depth = rgb[0].copy(), minT[0].copy()
    else:
        acc = rgb[1].copy()
        for k in range(2, 5):
            acc = (acc + rgb[k]).astype(np.float32)
        px = (acc / np.float32(4)).astype(np.float32)
        dsum, cnt = np.zeros(n, np.float32), np.zeros(n, np.int32)
        for k in range(1, 5):
            dsum = np.where(hit[k], (dsum + minT[k]).astype(np.float32), dsum)
            cnt += hit[k]
        depth = np.where(cnt > 0, (dsum / np.maximum(cnt, 1).astype(np.float32)).astype(np.float32), FLT_MAX)
        miss = ~hit[0]                       # centre-gated: a missed centre is the centre's own colour
        px[miss], depth[miss] = rgb[0][miss], FLT_MAX
    px = np.clip(px, 0, 1).astype(np.float32)
    compose_pixels.last = (o, d, rgb, tid, hit)
    return px, depth, tid[0], len(o)


def main():
    for name, W, H, msaa in CONFIGS:
        key = f"{name}_{W}x{H}_msaa{msaa}"
        sp = np.load(os.path.join(ROOT, "tests", "golden", f"sparse_{key}.npz"))
        x, y, corners, lights = sp["x"], sp["y"], sp["corners"], sp["lights"]
        s = O.Scene.load(os.path.join(ROOT, "scenes", name + ".obj"), textures=False)
        n = len(x)
        px, depth, tid0, n_rays = compose_pixels(s, sp, W, H, msaa)
        o, d, rgb, tid, hit = compose_pixels.last
        # what the oracle says for the same untextured pixels (informational)
        s.settings.MSAA = 1
        orgb, omt, otri = s.trace_rays(o[:n], d[:n], threads=8)
        print(f"{key}: {len(o)} rays through the reference; oracle centre tri equal "
              f"{np.mean(otri == tid[0]):.6f}, centre rgb bit-equal {np.mean((orgb == rgb[0]).all(1)):.6f}, "
              f"hit fraction {hit[0].mean():.4f}")
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"sparse_ref_{key}.npz"),
                            x=x, y=y, corners=corners, lights=lights, tri=tid[0], rgb=px, depth=depth,
                            rgb8=O.to_bytes(px))


if __name__ == "__main__":
    main()
