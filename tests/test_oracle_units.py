"""Properties and known answers of the CPU oracle that do not need the
reference build: loader quirks, walk vs brute force, MSAA gating, image code."""
import os

import numpy as np
import pytest

from conftest import SCENES

FLT_MAX = np.float32(3.402823466e+38)


def _load(oracle, name, **kw):
    return oracle.Scene.load(os.path.join(SCENES, name + ".obj"), **kw)


# ---- scene sizes recorded from a run of the reference itself ----------------
@pytest.mark.parametrize("name,tris,verts,nodes,depth,root_tris", [
    ("ball", 962, 486, 239, 17, 178), ("redballs", 3846, 1940, 867, 18, 4),
    ("lego", 4292, 4079, 735, 23, 18), ("final", 3846, 1940, 891, 20, 50),
    ("fin", 13444, 6756, 2675, 22, 168)])
def test_tree_shape_matches_reference_run(oracle, name, tris, verts, nodes, depth, root_tris):
    s = _load(oracle, name)
    c = s.counts()
    assert (c["triangles"], c["vertices"], c["nodes"], c["depth"]) == (tris, verts, nodes, depth)
    nd = s.nodes()
    assert nd["tri_cnt"][0] == root_tris
    assert nd["tri_cnt"].sum() == tris                      # every triangle in exactly one node
    assert sorted(nd["tris"].tolist()) == list(range(tris))
    # the root split sits at x = 0 (the depth-0 cost is a division by zero)
    assert nd["bounds"][nd["left"][0]][1] == 0.0


def test_material_carry_over_quirk(oracle):
    """mesh.cpp:368-393: fields a newmtl block omits keep the previous block's value."""
    s = _load(oracle, "final")
    names, m = s.material_names(), s.materials()
    assert names[0] == "StandardMaterialInitFromTriMesh" and m[0, 6] == np.float32(96.7)
    tex = dict(zip(names, m[:, 10]))
    assert tex["Blue_Mirror_Ball"] == -1 and tex["Checkerboard"] == 0
    for later in ("Lego_Transparant", "Red_diffuse_Ball", "White_Mirror_Wall",
                  "White_Transparant_ball", "Yellow_Mirror_Ball"):
        assert tex[later] == 0, later
    lego = _load(oracle, "lego")
    t = dict(zip(lego.material_names(), lego.materials()[:, 10]))
    assert t["Light"] == t["Legs-_Red"] >= 0       # inherited texture name
    assert t["Black"] == -1 and t["Material.001"] == -1  # absolute Windows paths fail
    assert lego.counts()["textures"] == 5


def test_illum3_wins_over_refraction(oracle):
    s = _load(oracle, "final")
    m = dict(zip(s.material_names(), s.materials()))
    assert m["White_Transparant_ball"][8] == 3 and m["White_Transparant_ball"][7] > 1.0
    r = _load(oracle, "refractionballs")
    mr = dict(zip(r.material_names(), r.materials()))
    assert mr["Refraction_Ball"][8] == 2 and mr["Refraction_Ball"][7] == np.float32(1.075)


def test_ngon_fan_triangulation(oracle, tmp_path):
    obj = tmp_path / "quad.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0\nf 1 2 3 4 5\nf 1//1 2//1 3//1\n")
    s = oracle.Scene.load(str(obj), build=False)
    v, t, m = s.triangles()
    assert v.tolist() == [[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 1, 2]]
    assert (t == 0).all() and (m == 0).all()


def test_empty_and_tiny_scenes(oracle, tmp_path):
    obj = tmp_path / "one.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    s = oracle.Scene.load(str(obj))
    assert s.counts()["nodes"] == 1
    tri, a, b, t, _ = s.check_hits([[0.2, 0.2, 1.0], [5, 5, 1]], [[0.2, 0.2, 0.0], [5, 5, 0]])
    assert tri.tolist() == [0, -1] and t[0] == 1.0 and t[1] == FLT_MAX


@pytest.mark.parametrize("name", ["ball", "final", "lego"])
def test_leaf_walk_is_closest_hit(oracle, name):
    """The leaf walk is an exact closest-hit search up to the neighbour-step quirk
    (a handful of rays per 100k)."""
    s = _load(oracle, name)
    rng = np.random.default_rng(7)
    vp, _ = s.vertices()
    lo, hi = vp.min(0), vp.max(0)
    o = (lo + (hi - lo) * rng.random((20000, 3))).astype(np.float32)
    v = rng.normal(size=(20000, 3))
    d = (o + 10 * v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    tri, _, _, t, _ = s.check_hits(o, d, threads=4)
    btri, bt = s.brute_hits(o, d)
    assert (tri == btri).mean() >= 0.9999
    same = tri == btri
    assert np.array_equal(t[same], bt[same])


def test_degenerate_determinant_counts_as_hit_at_origin(oracle, tmp_path):
    """Matrix33.h:132-134: det == 0 gives (0,0,0) -> a = b = 0, t = -0.0 -> accepted."""
    obj = tmp_path / "tri.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    s = oracle.Scene.load(str(obj))
    tri, a, b, t, hp = s.check_hits([[5, 5, 1.0]], [[6, 5, 1.0]])  # parallel to the triangle
    assert tri[0] == 0 and a[0] == 0 and b[0] == 0 and t[0] == 0 and np.signbit(t[0])
    assert hp[0].tolist() == [5, 5, 1]


def test_msaa_is_centre_gated(oracle):
    s = _load(oracle, "ball")
    m, lights = oracle.read_camera_sav(os.path.join(SCENES, "cameras", "ball.sav"))
    s.set_lights(lights)
    W = H = 48
    corners = oracle.camera_corners(m, W, H)
    s.settings.MSAA = 1
    c1 = oracle.Counters()
    rgb1, d1, tri1 = s.render(corners, W, H, counters=c1)
    s.settings.MSAA = 4
    c4 = oracle.Counters()
    rgb4, d4, tri4 = s.render(corners, W, H, counters=c4)
    miss = tri1 < 0
    assert 0 < miss.sum() < miss.size
    assert np.array_equal(tri1, tri4)
    assert np.array_equal(rgb4[miss], rgb1[miss])          # centre ray only
    assert (rgb4[miss] == np.float32([0.2, 0.5, 1.0])).all()
    assert (d4[miss] == FLT_MAX).all() and (d1[miss] == FLT_MAX).all()
    assert c1.camera_rays == W * H
    assert c4.camera_rays == W * H + 4 * int((~miss).sum())


def test_depth_normalisation(oracle):
    d = np.float32([0.5, 0.25, FLT_MAX, 1.0, 0.75])
    n = oracle.normalize_depth(d, 0.0)
    assert n.tolist() == [0.5, 0.25, 0.0, 1.0, 0.75]
    assert np.array_equal(oracle.normalize_depth(n, 0.0), n)    # idempotent only at focus 0
    f = oracle.normalize_depth(d, 0.5)
    assert f.tolist() == [0.0, 0.5, 0.0, 1.0, 0.5]
    assert not np.array_equal(oracle.normalize_depth(f, 0.5), f)


@pytest.mark.parametrize("intensity", [0.001, 0.01, 0.1, 0.3, 0.7, 0.9, 1.0])
def test_gaussian_map_known_shape(oracle, intensity):
    """The reference prints these seven maps on key 't' (main.cpp:238-245)."""
    m = oracle.gaussian_map(intensity)
    I = (1 - np.float32(intensity)) ** 2
    if intensity == 1.0:
        assert np.isnan(m).all()            # 1/(0*..) * e^0 = inf; inf/inf
        return
    assert abs(float(m.sum()) - 1.0) < 1e-5
    assert np.array_equal(m, m.T) and np.array_equal(m, m[::-1, ::-1])
    assert m[6, 6] == m.max()
    # the exponent uses integer division: (-(1) - 0)/2 == 0, so the 4-neighbours of the
    # centre weigh exactly as much as the centre
    assert m[6, 5] == m[6, 6] == m[5, 6]
    expect = np.exp(np.float32(-1.0) * (I * 2.5) ** 2)       # taps at squared distance 2
    assert abs(m[5, 5] / m[6, 6] - expect) < 1e-5


def test_gaussian_map_identity_when_in_focus(oracle):
    m = oracle.gaussian_map(0.0)
    assert m[6, 6] == 1 and m.sum() == 1


def test_blur_identity_and_nan_pixel(oracle):
    rng = np.random.default_rng(3)
    rgb = rng.random((20, 24, 3), dtype=np.float32)
    dn = np.zeros((20, 24), np.float32)
    out = oracle.blur(rgb, dn)
    assert out.shape == (8, 12, 3) and np.array_equal(out, rgb[6:-6, 6:-6])
    dn[10, 12] = 1.0                      # farthest pixel: inf/inf kernel
    out = oracle.blur(rgb, dn)
    assert np.isnan(out[4, 6]).all()
    assert oracle.to_bytes(out[4, 6]).tolist() == [0, 0, 0]


def test_quantisation_truncates(oracle):
    v = np.float32([0.0, 0.999, 1.0, 0.5, 1 / 255 - 1e-6, np.nan])
    assert oracle.to_bytes(v).tolist() == [0, 254, 255, 127, 0, 0]


def test_bmp_layout(oracle, tmp_path):
    W, H = 8, 4
    rgb = np.zeros((H, W, 3), np.float32)
    rgb[0, 0] = [1.0, 0.5, 0.25]
    p = tmp_path / "r.bmp"
    assert oracle.lib().or_write_bmp(str(p).encode(), rgb.reshape(-1), W, H) == 1
    b = p.read_bytes()
    assert len(b) == 54 + 3 * W * H and b[:2] == b"BM"
    assert int.from_bytes(b[2:6], "little") == 3 * W * H           # size field, no + 54
    assert int.from_bytes(b[10:14], "little") == 54
    assert int.from_bytes(b[18:22], "little") == W
    assert int.from_bytes(b[22:26], "little", signed=True) == -H    # top-down
    assert b[28] == 24 and list(b[54:57]) == [63, 127, 255]         # BGR, truncated
    p2 = tmp_path / "r.ppm"
    assert oracle.lib().or_write_ppm(str(p2).encode(), rgb.reshape(-1), W, H) == 1
    ppm = p2.read_bytes()
    assert ppm.startswith(b"P6\n8 4\n255\n") and list(ppm[11:14]) == [255, 127, 63]


def test_texture_wrap_and_padding(oracle):
    """final.obj: the checkerboard floor is textured; the five later materials inherit the
    texture but their faces carry no texcoords, so u = v = 0 wraps to 1.0 and the lookup
    lands in the zero padding rows -> ambient term 0."""
    s = _load(oracle, "final")
    t = s.textures()
    assert len(t) == 1 and t[0][1:] == (100, 100)
    px = t[0][0].reshape(102, 100, 4)
    assert (px[100:] == 0).all() and px[:100].any()


def test_soft_shadow_generator_and_loop(oracle):
    """The counter-based stand-in for rand() is a pure function of its key, and soft shadows
    (settings.h:13 shadowSamples > 1) shade penumbrae between lit and fully shadowed."""
    L = oracle.lib()
    import ctypes as C
    L.or_soft_rand.restype = C.c_uint32
    L.or_soft_rand.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    a = [L.or_soft_rand(12345, 6, 1, k, 2) for k in range(64)]
    assert a == [L.or_soft_rand(12345, 6, 1, k, 2) for k in range(64)] and len(set(a)) == 64
    assert max(a) < 2 ** 31 and L.or_soft_rand(12345, 6, 1, 0, 2) != L.or_soft_rand(12346, 6, 1, 0, 2)
    s = oracle.Scene.load(os.path.join(SCENES, "redballs.obj"))
    m, lights = oracle.read_camera_sav(os.path.join(SCENES, "cameras", "redballs.sav"))
    s.set_lights(lights)
    W, H = 64, 36
    corners = oracle.camera_corners(m, W, H)
    c1, c2 = oracle.Counters(), oracle.Counters()
    hard, _, _ = s.render(corners, W, H, threads=4, counters=c1)
    s.settings.shadowSamples = 2
    soft, _, _ = s.render(corners, W, H, threads=4, counters=c2)
    again, _, _ = s.render(corners, W, H, threads=2)
    assert np.array_equal(soft, again)                       # repeatable whatever the thread count
    assert c2.rays_shadow == 8 * c1.rays_shadow and not np.array_equal(hard, soft)
