"""Scratch GPU check: renders the BASELINE configs once and prints timing + stats."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import gpu_scene_from_oracle, pkg
from oracle import oracle as O
capi = pkg().capi
cfgs = [("ball", 800, 800, 1), ("redballs", 1920, 1080, 1), ("refractionballs", 1920, 1080, 1),
        ("lego", 3840, 2160, 4), ("final", 3840, 2160, 4)]
for name, W, H, msaa in cfgs:
    s = O.Scene.load(os.path.join(ROOT, "scenes", name + ".obj"))
    m, lights = O.read_camera_sav(os.path.join(ROOT, "scenes", "cameras", name + ".sav"))
    gs = gpu_scene_from_oracle(s)
    corners = O.camera_corners(m, W, H)
    p = capi.default_params(W, H, msaa)
    for it in range(2):
        t = time.time()
        rgb, depth, tri, st = gs.render(corners, lights, p)
        dt = time.time() - t
    rays = st.rays_closest + st.rays_shadow
    print(f"{name} {W}x{H} msaa{msaa}: kernel {st.kernel_ms:.2f} ms wall {dt*1e3:.1f} ms rays {rays/1e6:.1f}M "
          f"-> {rays/st.kernel_ms/1e3:.1f} Mrays/s tests/ray {st.tri_tests/rays:.1f} nodes/ray {st.node_visits/rays:.1f} "
          f"hitfrac {(tri>=0).mean():.3f}", flush=True)
