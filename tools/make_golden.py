#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/refprobe).

Build-container only: needs /root/reference (through `make -C oracle ref`).
For every scene it hands the mesh (as parsed by the oracle's loader, textures
off - the reference build here has no image library) to the reference code,
lets the reference compute vertex normals, build its KD tree and trace a fixed
set of rays, and stores inputs + the reference's answers.  The committed .npz
files are what tests compare the oracle (and, through it, the GPU path) with
where the reference itself cannot travel.

Ray sets per scene (seeded, deterministic):
  * camera rays of the scene's camera.sav at a reduced resolution, pixel
    centres plus the four MSAA-4 offsets of main.cpp:452-458 on a subset;
  * random interior rays: origins uniform in the mesh AABB, destinations on a
    sphere of radius 10 around the origin (numpy PCG64, seed 1806).
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SCENES = {
    # name: (camera W, H for the reduced camera set)
    "ball": (64, 64),
    "redballs": (96, 54),
    "refractionballs": (96, 54),
    "lego": (96, 54),
    "final": (96, 54),
}
N_RANDOM = 3000


def scene_blob(s, lights, textured=False):
    vpos, _ = s.vertices()
    tv, tt, tm = s.triangles()
    tc = s.texcoords()
    mats = s.materials()
    b = struct.pack("<i", len(vpos)) + vpos.astype("<f4").tobytes()
    b += struct.pack("<i", len(tv)) + tv.astype("<i4").tobytes() + tt.astype("<i4").tobytes()
    b += tm.astype("<i4").tobytes()
    b += struct.pack("<i", len(tc)) + tc.astype("<f4").tobytes()
    b += struct.pack("<i", len(mats)) + mats.astype("<f4").tobytes()
    b += struct.pack("<i", len(lights)) + lights.astype("<f4").tobytes()
    if textured:  # pixels as the image library would deliver them + the two zero rows; material rows name them
        tex = s.textures()
        b += struct.pack("<i", len(tex))
        for px, w, h in tex:
            assert px.size == (h + 2) * w * 4 and not px[h * w * 4:].any()
            b += struct.pack("<2i", w, h) + px.tobytes()
    return b


def make_rays(name, s, W, H):
    m, lights = O.read_camera_sav(os.path.join(ROOT, "scenes", "cameras", name + ".sav"))
    corners = O.camera_corners(m, W, H)
    o, d = [], []
    offs = [(0, 0), (-1 / 8, -3 / 8), (3 / 8, -1 / 8), (-3 / 8, 1 / 8), (1 / 8, 3 / 8)]
    for y in range(H):
        for x in range(W):
            use = offs if (x + y) % 7 == 0 else offs[:1]
            for ox, oy in use:
                a, b = O.pixel_ray(corners, W, H, x + ox, y + oy)
                o.append(a), d.append(b)
    rng = np.random.Generator(np.random.PCG64(1806))
    vpos, _ = s.vertices()
    lo, hi = vpos.min(0), vpos.max(0)
    ro = (lo + (hi - lo) * rng.random((N_RANDOM, 3))).astype(np.float32)
    v = rng.normal(size=(N_RANDOM, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    rd = (ro + 10.0 * v).astype(np.float32)
    o = np.concatenate([np.array(o, np.float32), ro])
    d = np.concatenate([np.array(d, np.float32), rd])
    return m, lights, corners, o, d


def parse_tree(buf, pos):
    """preorder dump -> arrays in preorder"""
    bounds, leaf, off, cnt, tris = [], [], [], [], []

    def rec(pos):
        b = struct.unpack_from("<6f", buf, pos)
        pos += 24
        lf, c = struct.unpack_from("<2i", buf, pos)
        pos += 8
        t = np.frombuffer(buf, "<i4", c, pos)
        pos += 4 * c
        bounds.append(b), leaf.append(lf), off.append(len(tris)), cnt.append(c)
        tris.extend(t.tolist())
        if not lf:
            pos = rec(pos)
            pos = rec(pos)
        return pos

    import sys as _s
    _s.setrecursionlimit(100000)
    pos = rec(pos)
    return pos, dict(bounds=np.array(bounds, np.float32), leaf=np.array(leaf, np.int32),
                     tri_off=np.array(off, np.int32), tri_cnt=np.array(cnt, np.int32),
                     tris=np.array(tris, np.int32))


def run_ref(s, lights, o, d, textured=False):
    probe = os.path.join(ROOT, "oracle", "_ref", "refprobe")
    with tempfile.TemporaryDirectory() as td:
        fs, fr, fo = (os.path.join(td, n) for n in ("scene.bin", "rays.bin", "out.bin"))
        open(fs, "wb").write(scene_blob(s, lights, textured))
        open(fr, "wb").write(struct.pack("<i", len(o)) + o.astype("<f4").tobytes()
                             + d.astype("<f4").tobytes())
        subprocess.run([probe, fs, fr, fo], check=True, stdout=subprocess.DEVNULL)
        buf = open(fo, "rb").read()
    pos = 0
    (nv,) = struct.unpack_from("<i", buf, pos)
    pos += 4
    normals = np.frombuffer(buf, "<f4", 3 * nv, pos).reshape(nv, 3).copy()
    pos += 12 * nv
    pos, tree = parse_tree(buf, pos)
    (nr,) = struct.unpack_from("<i", buf, pos)
    pos += 4
    rec = np.dtype([("bHit", "<i4"), ("v", "<i4", 3), ("ab", "<f4", 2), ("t", "<f4"),
                    ("hp", "<f4", 3), ("rgb", "<f4", 3), ("minT", "<f4")])
    rays = np.frombuffer(buf, rec, nr, pos).copy()
    return normals, tree, rays


def tri_ids(s, rays):
    """The reference returns a Triangle copy, not an index: recover it by its
    vertex triple (first match; duplicates share all three vertices)."""
    tv, _, _ = s.triangles()
    key = {}
    for i, v in enumerate(tv):
        key.setdefault(tuple(int(x) for x in v), i)
    out = np.full(len(rays), -1, np.int32)
    for i, r in enumerate(rays):
        if r["bHit"]:
            out[i] = key[tuple(int(x) for x in r["v"])]
    return out


def oracle_preorder(s):
    nd = s.nodes()
    order = []
    stack = [0]
    while stack:
        n = stack.pop()
        order.append(n)
        if nd["left"][n] >= 0:
            stack.append(nd["right"][n])
            stack.append(nd["left"][n])
    return nd, np.array(order)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "all", "ref"])
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    for name, (W, H) in SCENES.items():
        s = O.Scene.load(os.path.join(ROOT, "scenes", name + ".obj"), textures=False)
        m, lights, corners, o, d = make_rays(name, s, W, H)
        s.set_lights(lights)
        normals, tree, rays = run_ref(s, lights, o, d)
        tid = tri_ids(s, rays)
        # ---- report how the restatement compares (informational; tests assert)
        _, onrm = s.vertices()
        nd, order = oracle_preorder(s)
        same_tree = (len(order) == len(tree["leaf"])
                     and np.array_equal(nd["bounds"][order], tree["bounds"])
                     and np.array_equal(nd["tri_cnt"][order], tree["tri_cnt"]))
        rgb, minT, ftri = s.trace_rays(o, d, threads=8)
        htri, a, b, t, hp = s.check_hits(o, d, threads=8)
        print(f"{name}: rays {len(o)} normals_equal {np.array_equal(onrm, normals)} "
              f"tree_equal {same_tree} tri_equal {np.mean(htri == tid):.6f} "
              f"t_equal {np.mean(t == rays['t']):.6f} ab_equal "
              f"{np.mean((a == rays['ab'][:, 0]) & (b == rays['ab'][:, 1])):.6f} "
              f"rgb_bitequal {np.mean(np.all(rgb == rays['rgb'], axis=1)):.6f} "
              f"rgb_maxdiff {np.abs(rgb - rays['rgb']).max():.3g}")
        np.savez_compressed(
            os.path.join(ROOT, "tests", "golden", name + ".npz"),
            modelview=m, lights=lights, cam_wh=np.array([W, H]), corners=corners,
            origins=o, dests=d,
            ref_normals=normals,
            ref_tree_bounds=tree["bounds"], ref_tree_leaf=tree["leaf"],
            ref_tree_tri_off=tree["tri_off"], ref_tree_tri_cnt=tree["tri_cnt"],
            ref_tree_tris=tree["tris"],
            ref_hit=rays["bHit"].astype(np.int8), ref_tri=tid, ref_a=rays["ab"][:, 0],
            ref_b=rays["ab"][:, 1], ref_t=rays["t"], ref_hitpoint=rays["hp"],
            ref_rgb=rays["rgb"], ref_minT=rays["minT"])


if __name__ == "__main__":
    main()
