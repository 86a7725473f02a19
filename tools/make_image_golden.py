#!/usr/bin/env python3
"""Golden vectors for the image half of the hot path (depth normalisation, the
13x13 depth-of-field kernel, the blur, the BMP/PPM writers incl. the per-row
update), made by the reference's own imageWriter.h compiled here as
oracle/_ref/refimage (make -C oracle ref).  Build container only.

  python tools/make_image_golden.py        -> tests/golden/image_<case>.npz

Every file holds the inputs (rgb before the RGBValue clamp, depth with FLT_MAX
for misses, focusDepth) and what the reference made of them: the clamped image,
the depth plane after writeDepthBMP (normalised once) and after writeBlurredBMP
(normalised twice), the blurred float image, the gauseanMap of the reference's
own 't'-key list, and the bytes of progress.bmp (blank file + per-row updates),
result.ppm, result.bmp, depth.bmp and blurred.bmp.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REFIMAGE = os.path.join(ROOT, "oracle", "_ref", "refimage")
FLT_MAX = np.float32(3.4028234663852886e38)
T_KEY_LIST = np.array([0.001, 0.01, 0.1, 0.3, 0.7, 0.9, 1.0], np.float32)  # main.cpp:239-245


def synthetic(W, H, focus, seed):
    """Colours on both sides of the [0,1] clamp, depths in 0.2..0.8 with a tenth
    of the pixels missed, one pixel exactly in focus."""
    rng = np.random.default_rng(seed)
    rgb = rng.uniform(-0.25, 1.25, (H, W, 3)).astype(np.float32)
    depth = rng.uniform(0.2, 0.8, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.1] = FLT_MAX
    depth[H // 2, W // 2] = np.float32(focus) if focus > 0 else depth[H // 2, W // 2]
    depth[H // 3, W // 3] = np.float32(0.93)   # the farthest pixel, inside the blurred area
    return rgb, depth


def rendered(W, H):
    """A small frame of final.obj from the CPU oracle: realistic colour and depth planes."""
    from oracle import oracle as O
    s = O.Scene.load(os.path.join(ROOT, "scenes", "final.obj"))
    m, lights = O.read_camera_sav(os.path.join(ROOT, "scenes", "cameras", "final.sav"))
    s.set_lights(lights)
    rgb, depth, _ = s.render(O.camera_corners(m, W, H), W, H, threads=8)
    return rgb, depth


def run(rgb, depth, focus):
    H, W = depth.shape
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "in.bin"), "wb") as f:
            f.write(np.array([W, H], np.int32).tobytes())
            f.write(np.array([focus], np.float32).tobytes())
            f.write(np.ascontiguousarray(rgb, np.float32).tobytes())
            f.write(np.ascontiguousarray(depth, np.float32).tobytes())
        subprocess.check_call([REFIMAGE, os.path.join(tmp, "in.bin"), tmp], stdout=subprocess.DEVNULL)
        dump = np.fromfile(os.path.join(tmp, "dump.bin"), np.float32)
        n, nb = W * H, (W - 12) * (H - 12)
        parts = np.split(dump, np.cumsum([3 * n, n, n, 3 * nb]))
        assert parts[4].size == 7 * 169
        out = dict(rgb_in=rgb, depth_in=depth, focus=np.float32(focus),
                   image=parts[0].reshape(H, W, 3), depth1=parts[1].reshape(H, W),
                   depth2=parts[2].reshape(H, W), blurr=parts[3].reshape(H - 12, W - 12, 3),
                   t_key_list=T_KEY_LIST, maps=parts[4].reshape(7, 13, 13))
        for name in ("progress.bmp", "result.ppm", "result.bmp", "depth.bmp", "blurred.bmp"):
            out[name.replace(".", "_")] = np.fromfile(os.path.join(tmp, name), np.uint8)
    return out


def main():
    if not os.path.exists(REFIMAGE):
        raise SystemExit("build oracle/_ref first: make -C oracle ref")
    cases = {
        "aligned_64x48_focus0": (*synthetic(64, 48, 0.0, 1806), 0.0),
        "ragged_63x47_focus037": (*synthetic(63, 47, 0.37, 1807), 0.37),
        "final_96x54_focus0": (*rendered(96, 54), 0.0),
        "final_90x50_focus05": (*rendered(90, 50), 0.5),
    }
    for name, (rgb, depth, focus) in cases.items():
        out = run(rgb, depth, focus)
        path = os.path.join(ROOT, "tests", "golden", f"image_{name}.npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
