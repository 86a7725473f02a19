#!/usr/bin/env python3
"""Full-size frames of the five BASELINE configs on the CPU oracle.

Writes
  tests/golden/frame_counters.json  - per config the work counters of the
      reference's algorithm (ray counts, T = ray/triangle tests, S = leaf steps,
      H, Hl, C), for the whole frame and for the pixel-centre rays alone, and
      the algorithmic flops F = 45 T + 40 S + 30 H + 60 Hl + 25 C that
      bench.py's roofline figure is computed from;
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
]


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
