"""GPU check: renders the five BASELINE configs once and prints timing + stats."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("raytracing-project_amd")
capi, host = pkg.capi, pkg.host
cfgs = [("ball", 800, 800, 1), ("redballs", 1920, 1080, 1), ("refractionballs", 1920, 1080, 1),
        ("lego", 3840, 2160, 4), ("final", 3840, 2160, 4)]
for name, W, H, msaa in cfgs:
    hs = host.HostScene(os.path.join(ROOT, "scenes", name + ".obj"))
    m, lights = host.read_camera_sav(os.path.join(ROOT, "scenes", "cameras", name + ".sav"))
    gs = hs.to_gpu(0)
    corners = host.camera_corners(m, W, H)
    p = capi.default_params(W, H, msaa)
    for it in range(2):
        t = time.time()
        rgb, depth, tri, st = gs.render(corners, lights, p)
        dt = time.time() - t
    rays = st.rays_closest + st.rays_shadow
    print(f"{name} {W}x{H} msaa{msaa}: kernel {st.kernel_ms:.2f} ms wall {dt*1e3:.1f} ms rays {rays/1e6:.1f}M "
          f"-> {rays/st.kernel_ms/1e3:.1f} Mrays/s tests/ray {st.tri_tests/rays:.1f} nodes/ray {st.node_visits/rays:.1f} "
          f"hitfrac {(tri>=0).mean():.3f}", flush=True)
