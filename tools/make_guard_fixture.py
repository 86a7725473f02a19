#!/usr/bin/env python3
"""The rays on which the reference's leaf walk never ends (decision D7 of oracle/rt_oracle.c).
Build container only.

The reference's checkHit loop (raytracing.cpp:303-349) has no bound on its steps.  The oracle restates
it with a guard of 100000 steps and counts the walks the guard stops; frame_counters.json records 2 for
lego.obj at 3840x2160 MSAA 4.  This tool renders that frame on the oracle, collects the guarded rays
(origin, destination - they may be camera, shadow or bounce rays), and then hands each of them to the
REAL reference (oracle/_ref/refprobe) under a time limit to record what it does: it does not return.

  python tools/make_guard_fixture.py  -> tests/golden/walk_guard_lego.npz
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import oracle as O  # noqa: E402
from make_golden import scene_blob  # noqa: E402

NAME, W, H, MSAA = "lego", 3840, 2160, 4
LIMIT_S = 20


def reference_returns(s, lights, o, d):
    probe = os.path.join(ROOT, "oracle", "_ref", "refprobe")
    with tempfile.TemporaryDirectory() as td:
        fs, fr, fo = (os.path.join(td, n) for n in ("scene.bin", "rays.bin", "out.bin"))
        open(fs, "wb").write(scene_blob(s, lights))
        open(fr, "wb").write(struct.pack("<i", 1) + o.astype("<f4").tobytes() + d.astype("<f4").tobytes())
        try:
            subprocess.run([probe, fs, fr, fo], check=True, stdout=subprocess.DEVNULL, timeout=LIMIT_S)
            return True
        except subprocess.TimeoutExpired:
            return False


def main():
    threads = int(os.environ.get("THREADS", "8"))
    s = O.Scene.load(os.path.join(ROOT, "scenes", NAME + ".obj"))
    m, lights = O.read_camera_sav(os.path.join(ROOT, "scenes", "cameras", NAME + ".sav"))
    s.set_lights(lights)
    s.settings.MSAA = MSAA
    corners = O.camera_corners(m, W, H)
    O.guard_rays()
    c = O.Counters()
    for y0 in range(0, H, 120):
        s.render(corners, W, H, y0=y0, y1=min(y0 + 120, H), threads=threads, counters=c)
        print(f"rows {y0}..: guarded walks so far {c.walk_guard}", flush=True)
    rays = O.guard_rays()
    assert len(rays) == c.walk_guard
    rays = np.unique(rays, axis=0)
    o, d = rays[:, :3].copy(), rays[:, 3:].copy()
    # the oracle's answers (stopped walk) and the closest hit over all triangles with the same test
    tri, a, b, t, hp = s.check_hits(o, d)
    btri, bt = s.brute_hits(o, d)
    O.guard_rays()
    su = O.Scene.load(os.path.join(ROOT, "scenes", NAME + ".obj"), textures=False)
    returns = np.array([reference_returns(su, lights, o[i:i + 1], d[i:i + 1]) for i in range(len(o))])
    for i in range(len(o)):
        print(f"ray {i}: o {o[i]} d {d[i]} oracle tri {tri[i]} t {t[i]!r}; brute force tri {btri[i]} t {bt[i]!r}; "
              f"the reference returns within {LIMIT_S} s: {bool(returns[i])}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "walk_guard_lego.npz"), origins=o, dests=d,
                        oracle_tri=tri, oracle_t=t, brute_tri=btri, brute_t=bt, reference_returns=returns,
                        frame_walks=np.int64(c.walk_guard))


if __name__ == "__main__":
    main()
