#!/usr/bin/env python3
"""Adds `centre_primary` to tests/golden/frame_counters.json for the MSAA configs.

The reference shades the pixel-centre ray completely and then throws its colour
away when the centre hits (main.cpp:439-445); only the centre ray's closest-hit
query decides anything.  `centre_primary` counts that query alone (T and S of
the camera rays through the pixel centres), so that
    flops_needed = flops_frame - flops_centre + flops_centre_primary
is the work of the reference's algorithm for the rays whose results reach the
image.  bench.py reports the roofline figure for both.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def pixel_rays(corners, W, H, x, y):
    """or_pixel_ray for arrays of pixel coordinates (float32 arithmetic, same order)."""
    f = np.float32
    xsc = 1.0 - x.astype(np.float64) / (W - 1)
    ysc = 1.0 - y.astype(np.float64) / (H - 1)
    xs, xs1, ys, ys1 = (v.astype(f)[:, None] for v in (xsc, 1 - xsc, ysc, 1 - ysc))
    c = np.asarray(corners, f).reshape(8, 3)
    o00, d00, o01, d01, o10, d10, o11, d11 = c

    def mix(a00, a10, a01, a11):
        return (a00 * xs + a10 * xs1) * ys + (a01 * xs + a11 * xs1) * ys1
    return mix(o00, o10, o01, o11).astype(f), mix(d00, d10, d01, d11).astype(f)


def main():
    threads = int(os.environ.get("THREADS", "8"))
    path = os.path.join(ROOT, "tests", "golden", "frame_counters.json")
    out = json.load(open(path))
    for key, e in out.items():
        if e["msaa"] == 1:
            continue
        W, H = e["width"], e["height"]
        s = O.Scene.load(os.path.join(ROOT, "scenes", e["scene"] + ".obj"))
        m, lights = O.read_camera_sav(os.path.join(ROOT, "scenes", "cameras", e["scene"] + ".sav"))
        corners = O.camera_corners(m, W, H)
        o1, d1 = pixel_rays(corners, W, H, np.array([17, W - 3]), np.array([H - 5, 11]))
        for i, (x, y) in enumerate(((17, H - 5), (W - 3, 11))):  # the vector form equals the oracle's
            o0, d0 = O.pixel_ray(corners, W, H, x, y)
            assert (o0 == o1[i]).all() and (d0 == d1[i]).all()
        c = O.Counters()
        for y0 in range(0, H, 135):
            ys, xs = np.mgrid[y0:min(y0 + 135, H), 0:W]
            o, d = pixel_rays(corners, W, H, xs.ravel(), ys.ravel())
            s.check_hits(o, d, threads=threads, counters=c)
        e["centre_primary"] = c.as_dict()
        e["flops_centre_primary"] = c.flops()
        e["flops_needed"] = e["flops_frame"] - e["flops_centre"] + c.flops()
        print(key, c.as_dict(), e["flops_needed"] / e["flops_frame"], flush=True)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
