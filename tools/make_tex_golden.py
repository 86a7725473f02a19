#!/usr/bin/env python3
"""Material tables and TEXTURED rays from the REAL reference (oracle/_ref/refprobe).
Build container only.

  1. Mesh::loadMtl (mesh.cpp:354-480) run by the reference itself on each of the nine .mtl files in
     scenes/ (byte-identical to the reference's): per material, in file order, what the loader left
     in it, including the fields that leak from one `newmtl` block into the next because
     Material::cleanup() (mesh.h:62-72) resets only the flags  -> tests/golden/mtl_tables.npz
  2. Material::getTexture / getTexture2 (mesh.cpp:485-553) run by the reference itself inside its
     performRayTracing, on the pixels of scenes/textures/*.rtex (what a lossless PNG decoder hands
     over) followed by two zero rows (mesh.cpp:499 reads up to one row past the image):
       * the golden rays of tests/golden/{lego,final}.npz, textured  -> tests/golden/reftex_<scene>.npz
       * the 20000 full-size pixels of sparse_<config>.npz of the two textured configs, composed as
         main.cpp:435-511 composes them  -> tests/golden/sparse_reftex_<config>.npz

  python tools/make_tex_golden.py
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
from make_counters import CONFIGS, pixel_rays  # noqa: E402
from make_golden import run_ref, tri_ids  # noqa: E402
from make_sparse_ref import compose_pixels  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
PROBE = os.path.join(ROOT, "oracle", "_ref", "refprobe")
MTL = ["ball", "checkercube", "depth", "fin", "final", "lego", "lego_transparant", "redballs", "refractionballs"]
TEXTURED = ["lego", "final", "fin"]


def read_mtl_dump(path):
    b = open(path, "rb").read()
    pos = 0
    (n,) = struct.unpack_from("<i", b, pos)
    pos += 4
    names, vals, illum, flags, tex = [], [], [], [], []
    for _ in range(n):
        (ln,) = struct.unpack_from("<i", b, pos)
        pos += 4
        names.append(b[pos:pos + ln].decode("latin-1"))
        pos += ln
        vals.append(struct.unpack_from("<9f", b, pos))
        pos += 36
        il, fl = struct.unpack_from("<2i", b, pos)
        pos += 8
        illum.append(il), flags.append(fl)
        (ln,) = struct.unpack_from("<i", b, pos)
        pos += 4
        tex.append(b[pos:pos + ln].decode("latin-1"))
        pos += ln + 4
    return names, np.array(vals, np.float32).reshape(n, 9), np.array(illum, np.int32), np.array(flags, np.int32), tex


def mtl_tables():
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name in MTL:
            dump = os.path.join(td, name + ".bin")
            subprocess.run([PROBE, "--mtl", os.path.join(ROOT, "scenes", name + ".mtl"), dump], check=True,
                           stdout=subprocess.DEVNULL)
            names, vals, illum, flags, tex = read_mtl_dump(dump)
            out[name + "_names"] = np.array(names)
            out[name + "_values"] = vals      # Kd[3] Ks[3] Ns Ni Tr
            out[name + "_illum"] = illum
            out[name + "_flags"] = flags      # has_Kd 1, Ka 2, Ks 4, Ns 8, Ni 16, illum 32, Tr 64 (own block only)
            out[name + "_texture"] = np.array(tex)
            print(f"{name}.mtl: {len(names)} materials, textured {[n for n, t in zip(names, tex) if t]}")
    np.savez_compressed(os.path.join(GOLDEN, "mtl_tables.npz"), **out)


def textured_rays():
    for name in TEXTURED[:2]:   # (fin.obj has no reduced-camera ray fixture; its full-size pixels follow)
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        s = O.Scene.load(os.path.join(ROOT, "scenes", name + ".obj"), textures=True)
        s.set_lights(g["lights"])
        _, _, rays = run_ref(s, g["lights"], g["origins"], g["dests"], textured=True)
        tid = tri_ids(s, rays)
        assert np.array_equal(tid, g["ref_tri"]) and np.array_equal(rays["minT"], g["ref_minT"])
        rgb, _, _ = s.trace_rays(g["origins"], g["dests"], threads=8)
        print(f"{name}: {len(rays)} textured rays; differ from the untextured answers on "
              f"{np.mean((rays['rgb'] != g['ref_rgb']).any(1)):.4f}; oracle bit-equal "
              f"{np.mean((rgb == rays['rgb']).all(1)):.6f}")
        np.savez_compressed(os.path.join(GOLDEN, f"reftex_{name}.npz"), ref_rgb_tex=rays["rgb"])


def textured_pixels():
    for name, W, H, msaa in CONFIGS:
        if name not in TEXTURED:
            continue
        key = f"{name}_{W}x{H}_msaa{msaa}"
        sp = np.load(os.path.join(GOLDEN, f"sparse_{key}.npz"))
        s = O.Scene.load(os.path.join(ROOT, "scenes", name + ".obj"), textures=True)
        px, depth, tid, n_rays = compose_pixels(s, sp, W, H, msaa, textured=True)
        same = np.mean((O.to_bytes(px) == sp["rgb8"]).all(1))
        print(f"{key}: {n_rays} textured rays through the reference; bytes equal to the oracle's frame on {same:.6f}")
        np.savez_compressed(os.path.join(GOLDEN, f"sparse_reftex_{key}.npz"), x=sp["x"], y=sp["y"],
                            corners=sp["corners"], lights=sp["lights"], tri=tid, rgb=px, depth=depth,
                            rgb8=O.to_bytes(px))


if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "all", "ref"])
    mtl_tables()
    textured_rays()
    textured_pixels()
