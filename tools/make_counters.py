#!/usr/bin/env python3
"""Full-size frames of the five BASELINE configs on the CPU oracle.

Writes
  tests/golden/frame_counters.json  - per config the work counters of the
      reference's algorithm (ray counts, T = ray/triangle tests, S = leaf steps,
      H, Hl, C), for the whole frame and for the pixel-centre rays alone, and
      the algorithmic flops F = 45 T + 40 S + 30 H + 60 Hl + 25 C
      (flops_frame).  The reference spends part of F on results it never uses:
      with MSAA it shades the pixel-centre ray and throws the colour away when
      the centre hits (main.cpp:439-445; only that ray's closest-hit query,
      counted in centre_primary, decides anything), and it traces shadow
      queries whose answer cannot change shadowMax (counted in moot_*).
      flops_needed = F less both, the figure bench.py's roofline uses;
  tests/golden/sparse_<config>.npz  - 20000 seeded pixels of each full-size
      frame (x, y, primary triangle, 8-bit RGB, depth) for spot checks of
      full-resolution GPU frames.
Takes ~15 minutes on 8 cores; run once, results are committed.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CONFIGS = [
    ("ball", 800, 800, 1),
    ("redballs", 1920, 1080, 1),
    ("refractionballs", 1920, 1080, 1),
    ("lego", 3840, 2160, 4),
    ("final", 3840, 2160, 4),
    # not a BASELINE config: the reference's largest scene (13444 triangles) and, besides refractionballs, the
    # only one with a refractive material (fin.mtl:122-129); camera and lights of final.sav
    ("fin", 1920, 1080, 1),
]


def pixel_rays(corners, W, H, x, y):
    """or_pixel_ray for arrays of pixel coordinates (float32 arithmetic, same order)."""
    f = np.float32
    xsc = 1.0 - x.astype(np.float64) / (W - 1)
    ysc = 1.0 - y.astype(np.float64) / (H - 1)
    xs, xs1, ys, ys1 = (v.astype(f)[:, None] for v in (xsc, 1 - xsc, ysc, 1 - ysc))
    o00, d00, o01, d01, o10, d10, o11, d11 = np.asarray(corners, f).reshape(8, 3)

    def mix(a00, a10, a01, a11):
        return (a00 * xs + a10 * xs1) * ys + (a01 * xs + a11 * xs1) * ys1
    return mix(o00, o10, o01, o11).astype(f), mix(d00, d10, d01, d11).astype(f)


def centre_primary(s, corners, W, H, threads):
    """Counters of the closest-hit queries of the pixel-centre rays alone."""
    o1, d1 = pixel_rays(corners, W, H, np.array([17, W - 3]), np.array([H - 5, 11]))
    for i, (x, y) in enumerate(((17, H - 5), (W - 3, 11))):  # the vector form equals the oracle's
        o0, d0 = O.pixel_ray(corners, W, H, x, y)
        assert (o0 == o1[i]).all() and (d0 == d1[i]).all()
    c = O.Counters()
    for y0 in range(0, H, 135):
        ys, xs = np.mgrid[y0:min(y0 + 135, H), 0:W]
        o, d = pixel_rays(corners, W, H, xs.ravel(), ys.ravel())
        s.check_hits(o, d, threads=threads, counters=c)
    return c


def main():
    threads = int(os.environ.get("THREADS", "8"))
    only = sys.argv[1:]
    path = os.path.join(ROOT, "tests", "golden", "frame_counters.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name, W, H, msaa in CONFIGS:
        if only and name not in only:
            continue
        s = O.Scene.load(os.path.join(ROOT, "scenes", name + ".obj"))
        m, lights = O.read_camera_sav(os.path.join(ROOT, "scenes", "cameras", name + ".sav"))
        s.set_lights(lights)
        s.settings.MSAA = msaa
        corners = O.camera_corners(m, W, H)
        c, cen = O.Counters(), O.Counters()
        t = time.time()
        rgb, depth, tri = s.render(corners, W, H, threads=threads, counters=c, centre=cen)
        dt = time.time() - t
        key = f"{name}_{W}x{H}_msaa{msaa}"
        out[key] = dict(scene=name, width=W, height=H, msaa=msaa, lights=len(lights),
                        frame=c.as_dict(), centre=cen.as_dict(), flops_frame=c.flops(),
                        flops_centre=cen.flops(), oracle_seconds=round(dt, 2), oracle_threads=threads,
                        hit_fraction=float((tri >= 0).mean()))
        needed = c.flops() - c.flops_moot()
        if msaa > 1:
            cp = centre_primary(s, corners, W, H, threads)
            out[key].update(centre_primary=cp.as_dict(), flops_centre_primary=cp.flops())
            needed += cp.flops() - (cen.flops() - cen.flops_moot())
        out[key].update(flops_moot=c.flops_moot(), flops_needed=needed)
        rng = np.random.Generator(np.random.PCG64(1806))
        idx = rng.choice(W * H, 20000, replace=False)
        ys, xs = idx // W, idx % W
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"sparse_{key}.npz"),
                            x=xs.astype(np.int32), y=ys.astype(np.int32), tri=tri[ys, xs],
                            rgb8=O.to_bytes(rgb[ys, xs]), depth=depth[ys, xs], corners=corners,
                            lights=lights)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)
        print(key, f"{dt:.1f}s", out[key]["frame"], flush=True)


if __name__ == "__main__":
    main()
