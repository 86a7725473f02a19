"""The CPU oracle against the golden vectors produced by the REAL reference
(tools/make_golden.py -> oracle/_ref/refprobe).  Everything here is bit-exact:
the oracle restates the reference's arithmetic in the reference's order."""
import os

import numpy as np
import pytest

from conftest import SCENES, load_golden

ALL = ["ball", "redballs", "refractionballs", "lego", "final"]


@pytest.fixture(scope="module", params=ALL)
def case(request, oracle):
    name = request.param
    # the reference build has no image library: no material is textured there
    s = oracle.Scene.load(os.path.join(SCENES, name + ".obj"), textures=False)
    g = load_golden(name)
    s.set_lights(g["lights"])
    return name, s, g


def test_vertex_normals_equal_reference(case):
    _, s, g = case
    assert np.array_equal(s.vertices()[1], g["ref_normals"])


def test_kd_tree_equals_reference(case):
    _, s, g = case
    nd = s.nodes()
    order, stack = [], [0]
    while stack:  # preorder, left first: the order the reference dump uses
        n = stack.pop()
        order.append(n)
        if nd["left"][n] >= 0:
            stack += [nd["right"][n], nd["left"][n]]
    order = np.array(order)
    assert len(order) == len(g["ref_tree_leaf"])
    assert np.array_equal(nd["bounds"][order], g["ref_tree_bounds"])
    assert np.array_equal((nd["left"][order] < 0).astype(np.int32), g["ref_tree_leaf"])
    assert np.array_equal(nd["tri_cnt"][order], g["ref_tree_tri_cnt"])
    mine = np.concatenate([nd["tris"][nd["tri_off"][n]:nd["tri_off"][n] + nd["tri_cnt"][n]]
                           for n in order])
    assert np.array_equal(mine, g["ref_tree_tris"])


def test_closest_hits_equal_reference(case):
    _, s, g = case
    tri, a, b, t, hp = s.check_hits(g["origins"], g["dests"], threads=4)
    assert np.array_equal(tri >= 0, g["ref_hit"].astype(bool))
    assert np.array_equal(tri, g["ref_tri"])
    assert np.array_equal(t, g["ref_t"])
    hit = tri >= 0
    assert np.array_equal(a[hit], g["ref_a"][hit]) and np.array_equal(b[hit], g["ref_b"][hit])
    assert np.array_equal(hp[hit], g["ref_hitpoint"][hit])


def test_shading_recursion_equals_reference(case):
    _, s, g = case
    rgb, minT, tri = s.trace_rays(g["origins"], g["dests"], threads=4)
    assert np.array_equal(minT, g["ref_minT"])
    assert np.array_equal(rgb.view(np.uint32), g["ref_rgb"].view(np.uint32))


def test_golden_rays_come_from_the_committed_camera(case, oracle):
    name, _, g = case
    m, lights = oracle.read_camera_sav(os.path.join(SCENES, "cameras", name + ".sav"))
    assert np.array_equal(m, g["modelview"]) and np.array_equal(lights, g["lights"])
    W, H = (int(v) for v in g["cam_wh"])
    assert np.array_equal(oracle.camera_corners(m, W, H), g["corners"])
    o, d = oracle.pixel_ray(g["corners"], W, H, 0, 0)
    assert np.array_equal(o, g["origins"][0]) and np.array_equal(d, g["dests"][0])


@pytest.mark.parametrize("name,W,H", [("ball", 800, 800), ("redballs", 1920, 1080), ("refractionballs", 1920, 1080)])
def test_full_size_pixels_equal_reference(oracle, name, W, H):
    """Full-resolution camera rays of the MSAA-1 configs: the oracle's pixel ray + shading
    recursion against the reference's answers for the same 20000 pixels (untextured build)."""
    g = load_golden(f"sparse_ref_{name}_{W}x{H}_msaa1")
    s = oracle.Scene.load(os.path.join(SCENES, name + ".obj"), textures=False)
    s.set_lights(g["lights"])
    o = np.zeros((len(g["x"]), 3), np.float32)
    d = np.zeros_like(o)
    for i, (x, y) in enumerate(zip(g["x"], g["y"])):
        o[i], d[i] = oracle.pixel_ray(g["corners"], W, H, float(x), float(y))
    rgb, min_t, tri = s.trace_rays(o, d, threads=8)
    assert np.array_equal(tri, g["tri"]) and np.array_equal(min_t, g["depth"])
    assert np.array_equal(np.clip(rgb, 0, 1), g["rgb"])


# ---------------------------------------------------------------------------
# the material loader and the texture lookup, pinned by the reference itself
# (tools/make_tex_golden.py: Mesh::loadMtl and Material::getTexture run unmodified)
# ---------------------------------------------------------------------------
MTL_KEYS = {"Kd": "Kd", "Ka": "Ka", "Ks": "Ks", "Ns": "Ns", "Ni": "Ni", "illum": "illum", "Tr": "Tr", "d": "Tr",
            "map_Kd": "map_Kd"}


def _fields_set_so_far(path):
    """Per material block of a .mtl, in file order: the keys some line at or above the block has set.
    (Not a loader: it only says which fields of the reference's dump are defined - the reference's
    `Material mat` starts uninitialised, mesh.cpp:369.)"""
    seen, out, name = set(), [], None
    for line in open(path, "r", errors="replace"):
        if line.startswith("newmtl "):
            name = line.split()[1]
            out.append([name, None])
        else:
            for k, field in MTL_KEYS.items():
                if line.startswith(k + " "):
                    seen.add(field)
        if out:
            out[-1][1] = set(seen)
    return out


def _rtex_pixels(path):
    import struct
    import zlib
    b = open(path, "rb").read()
    assert b[:4] == b"RTEX"
    w, h, n = struct.unpack_from("<III", b, 4)
    return np.frombuffer(zlib.decompress(b[16:16 + n]), np.uint8), w, h


@pytest.mark.parametrize("name", ["ball", "checkercube", "depth", "fin", "final", "lego", "redballs",
                                  "refractionballs"])
def test_material_table_equals_reference(oracle, name):
    """What Mesh::loadMtl leaves in every material - including what leaks from one newmtl block into
    the next (mesh.h:62-72 resets flags only): Kd, Ks, Ns, Ni and whether its own block set it, illum,
    and the texture name - against the oracle's loader.  Material 0 of the oracle is the built-in
    default of mesh.cpp:110-119; the file's materials follow."""
    g = load_golden("mtl_tables")
    names, vals = list(g[name + "_names"]), g[name + "_values"]
    illum, flags, tex = g[name + "_illum"], g[name + "_flags"], list(g[name + "_texture"])
    s = oracle.Scene.load(os.path.join(SCENES, name + ".obj"), textures=True, build=False)
    m = s.materials()
    assert s.material_names()[1:] == names
    blocks = [b for b in _fields_set_so_far(os.path.join(SCENES, name + ".mtl"))]
    by_name = {}
    for nm, seen in blocks:            # a name defined twice keeps its first definition (mesh.cpp:385)
        by_name.setdefault(nm, seen)
    textures = s.textures()
    for i, nm in enumerate(names):
        row, seen = m[i + 1], by_name[nm]
        if "Kd" in seen:
            assert np.array_equal(row[0:3], vals[i, 0:3]), nm
        if "Ks" in seen:
            assert np.array_equal(row[3:6], vals[i, 3:6]), nm
        if "Ns" in seen:
            assert row[6] == vals[i, 6], nm
        if "Ni" in seen:
            assert row[7] == vals[i, 7], nm
        if "illum" in seen:
            assert int(row[8]) == illum[i], nm
        assert bool(row[9]) == bool(flags[i] & 16), nm          # has_Ni: the block's own flag
        # the texture: the name the reference holds, resolved the way the loaders resolve it
        # (backslashes, relative to the scene; absolute Windows paths do not exist anywhere)
        rel = tex[i].replace("\\", "/").strip()
        rtex = os.path.join(SCENES, rel + ".rtex")
        want = bool(rel) and not (len(rel) > 1 and rel[1] == ":") and os.path.exists(rtex)
        assert (int(row[10]) >= 0) == want, (nm, rel)
        if want:
            px, w, h = _rtex_pixels(rtex)
            opx, ow, oh = textures[int(row[10])]
            assert (w, h) == (ow, oh) and np.array_equal(px, opx[: w * h * 4]) and not opx[w * h * 4:].any()


def test_material_table_fixture_holds_the_leaks():
    """The fixture itself shows the quirk the headline scene's look hangs on: in final.mtl only
    Checkerboard has a map_Kd line, and the five materials after it carry its texture name."""
    g = load_golden("mtl_tables")
    names, tex = list(g["final_names"]), list(g["final_texture"])
    at = names.index("Checkerboard")
    assert tex[at].endswith("checkerboard.png") and all(t == "" for t in tex[:at])
    assert all(t == tex[at] for t in tex[at:]) and len(tex) - at == 6
    lego = dict(zip(g["lego_names"], g["lego_texture"]))
    assert lego["Light"].endswith("Space Marine Legs.png")


@pytest.mark.parametrize("name", ["lego", "final"])
def test_textured_shading_equals_reference(oracle, name):
    """Material::getTexture / getTexture2 (mesh.cpp:485-553) inside the reference's own
    performRayTracing, on the .rtex pixels + two zero rows: colours of the golden rays, bit for bit."""
    g, t = load_golden(name), load_golden("reftex_" + name)
    s = oracle.Scene.load(os.path.join(SCENES, name + ".obj"), textures=True)
    s.set_lights(g["lights"])
    rgb, min_t, tri = s.trace_rays(g["origins"], g["dests"], threads=4)
    assert np.array_equal(tri, g["ref_tri"]) and np.array_equal(min_t, g["ref_minT"])
    assert np.array_equal(rgb.view(np.uint32), t["ref_rgb_tex"].view(np.uint32))
    assert (t["ref_rgb_tex"] != g["ref_rgb"]).any(axis=1).mean() > 0.02     # the texture does show


@pytest.mark.parametrize("name,W,H", [("lego", 3840, 2160), ("final", 3840, 2160)])
def test_full_size_textured_pixels_equal_reference(oracle, name, W, H):
    """The two textured configs at full size: the oracle's committed frame pixels (sparse_*) carry the
    bytes the reference's textured rays compose to (sparse_reftex_*), and a fresh oracle run of the
    first 2500 pixels' rays (centre + four samples, main.cpp:435-458) gives the same floats."""
    key = f"{name}_{W}x{H}_msaa4"
    g, r = load_golden("sparse_" + key), load_golden("sparse_reftex_" + key)
    assert np.array_equal(g["x"], r["x"]) and np.array_equal(g["y"], r["y"])
    assert np.array_equal(g["rgb8"], r["rgb8"]) and np.array_equal(g["tri"], r["tri"])
    n = 2500
    s = oracle.Scene.load(os.path.join(SCENES, name + ".obj"), textures=True)
    s.set_lights(r["lights"])
    offs = [(0.0, 0.0), (-1 / 8, -3 / 8), (3 / 8, -1 / 8), (-3 / 8, 1 / 8), (1 / 8, 3 / 8)]
    o = np.zeros((5, n, 3), np.float32)
    d = np.zeros_like(o)
    for k, (ox, oy) in enumerate(offs):
        for i in range(n):
            o[k, i], d[k, i] = oracle.pixel_ray(r["corners"], W, H, float(r["x"][i]) + ox, float(r["y"][i]) + oy)
    rgb, min_t, tri = s.trace_rays(o.reshape(-1, 3), d.reshape(-1, 3), threads=8)
    rgb, tri = rgb.reshape(5, n, 3), tri.reshape(5, n)
    acc = rgb[1].copy()
    for k in range(2, 5):
        acc = (acc + rgb[k]).astype(np.float32)
    px = (acc / np.float32(4)).astype(np.float32)
    px[tri[0] < 0] = rgb[0][tri[0] < 0]
    assert np.array_equal(np.clip(px, 0, 1), r["rgb"][:n])


def test_rays_on_which_the_reference_never_returns(oracle):
    """Decision D7: the two rays of lego.obj 3840x2160 MSAA 4 on which the reference's leaf walk
    (raytracing.cpp:303-349 has no bound) steps between neighbouring leaves for ever - the fixture
    records that the real reference did not return within 20 s on either (tools/make_guard_fixture.py).
    The oracle stops the walk after 100000 steps, counts it, and keeps the closest hit found until
    then, which is the closest hit over all triangles."""
    g = load_golden("walk_guard_lego")
    assert len(g["origins"]) == 2 and not g["reference_returns"].any() and int(g["frame_walks"]) == 2
    s = oracle.Scene.load(os.path.join(SCENES, "lego.obj"))
    oracle.guard_rays()
    c = oracle.Counters()
    tri, a, b, t, hp = s.check_hits(g["origins"], g["dests"], counters=c)
    assert c.walk_guard == 2 and c.leaf_steps > 200000
    assert np.array_equal(oracle.guard_rays()[:, :3], g["origins"])
    assert np.array_equal(tri, g["oracle_tri"]) and np.array_equal(t, g["oracle_t"])
    btri, bt = s.brute_hits(g["origins"], g["dests"])
    assert np.array_equal(tri, btri) and np.array_equal(t, bt)
    assert np.array_equal(btri, g["brute_tri"]) and np.array_equal(bt, g["brute_t"])
