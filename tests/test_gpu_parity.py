"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
identical inputs, and against the golden vectors produced by the real
reference.  Integer results (triangle ids) are compared exactly; a, b, t and
the hit point are bit-exact because the GPU evaluates the reference's
arithmetic in the reference's order."""
import os

import numpy as np
import pytest

from conftest import SCENES, gpu_scene_from_oracle, load_golden, pkg

pytestmark = pytest.mark.gpu

ALL = ["ball", "redballs", "refractionballs", "lego", "final"]


def _scene(oracle, name, textures=True):
    s = oracle.Scene.load(os.path.join(SCENES, name + ".obj"), textures=textures)
    g = load_golden(name)
    s.set_lights(g["lights"])
    return s, g


@pytest.mark.parametrize("name", ALL)
def test_check_hits_match_oracle_and_reference(oracle, name):
    s, g = _scene(oracle, name)
    gs = gpu_scene_from_oracle(s)
    o, d = g["origins"], g["dests"]
    tri, a, b, t, hp = gs.check_hits(o, d)
    otri, oa, ob, ot, ohp = s.check_hits(o, d, threads=8)
    # the GPU walks the tree front to back with a stack, the reference steps
    # from leaf to leaf; both give the closest hit.  The reference's neighbour
    # search mis-steps on a handful of rays (KDtree.cpp:288-587), so allow 1e-4.
    same = tri == otri
    assert same.mean() >= 1 - 1e-4, f"tri mismatch {np.flatnonzero(~same)[:10]}"
    ok = same & (tri >= 0)
    assert np.array_equal(a[ok], oa[ok]) and np.array_equal(b[ok], ob[ok])
    assert np.array_equal(t[same], ot[same])
    assert np.array_equal(hp[ok], ohp[ok])
    # and against the real reference's answers (golden vectors)
    assert (tri == g["ref_tri"]).mean() >= 1 - 1e-4
    ok = (tri == g["ref_tri"]) & (tri >= 0)
    assert np.array_equal(t[ok], g["ref_t"][ok])
    assert np.array_equal(a[ok], g["ref_a"][ok])


@pytest.mark.parametrize("name", ALL)
def test_trace_rays_match_oracle(oracle, name):
    s, g = _scene(oracle, name)
    gs = gpu_scene_from_oracle(s)
    o, d = g["origins"], g["dests"]
    rgb, mt, tri, st = gs.trace_rays(o, d, g["lights"])
    orgb, omt, otri = s.trace_rays(o, d, threads=8)
    assert (tri == otri).mean() >= 1 - 1e-4
    assert np.array_equal(mt[tri == otri], omt[tri == otri])
    # colour: path weights instead of recursion, device powf -> a few ulp
    q, oq = oracle.to_bytes(np.clip(rgb, 0, 1)), oracle.to_bytes(np.clip(orgb, 0, 1))
    diff = np.abs(q.astype(int) - oq.astype(int)).max(axis=1)
    assert (diff <= 1).mean() >= 0.999, f"{(diff > 1).sum()} rays differ by more than 1 LSB"
    assert np.abs(rgb - orgb).max() < 2.0
    close = np.abs(rgb - orgb).max(axis=1) < 1e-4
    assert close.mean() >= 0.999
    assert st.rays_closest > 0 and st.camera_rays == len(o)


@pytest.mark.parametrize("name", ["ball", "refractionballs", "final"])
def test_trace_rays_untextured_match_reference(oracle, name):
    """The golden colours come from the reference built without an image
    library, i.e. with no material textured."""
    s, g = _scene(oracle, name, textures=False)
    gs = gpu_scene_from_oracle(s)
    rgb, mt, tri, _ = gs.trace_rays(g["origins"], g["dests"], g["lights"])
    same = tri == g["ref_tri"]
    assert same.mean() >= 1 - 1e-4
    q, rq = oracle.to_bytes(np.clip(rgb, 0, 1)), oracle.to_bytes(np.clip(g["ref_rgb"], 0, 1))
    diff = np.abs(q.astype(int) - rq.astype(int)).max(axis=1)
    assert (diff <= 1).mean() >= 0.999


def test_rays_on_which_the_reference_never_returns(oracle):
    """The two rays of lego.obj at 4K MSAA 4 on which the reference's unbounded leaf walk does not end
    (fixture walk_guard_lego.npz records that the real reference did not return): the stack walk has no
    such state and returns the closest hit over all triangles, bit for bit."""
    g = load_golden("walk_guard_lego")
    gs = pkg().host.HostScene(os.path.join(SCENES, "lego.obj")).to_gpu(0)
    tri, a, b, t, hp = gs.check_hits(g["origins"], g["dests"])
    assert np.array_equal(tri, g["brute_tri"]) and np.array_equal(t, g["brute_t"])
    rgb, mt, tri2, _ = gs.trace_rays(g["origins"], g["dests"], np.float32([[0, 10, 0], [4, 6, 5]]))
    assert np.array_equal(tri2, g["brute_tri"]) and np.array_equal(mt, g["brute_t"]) and np.isfinite(rgb).all()


@pytest.mark.parametrize("name", ["lego", "final"])
def test_trace_rays_textured_match_reference(oracle, name):
    """The same rays with the scenes' textures: colours the reference's own Material::getTexture
    (mesh.cpp:485-553) produced on the .rtex pixels + two zero rows (tools/make_tex_golden.py) - through
    the product's loader, so the material table with its inherited texture names is part of what is compared."""
    P = pkg()
    g, t = load_golden(name), load_golden("reftex_" + name)
    gs = P.host.HostScene(os.path.join(SCENES, name + ".obj")).to_gpu(0)
    rgb, mt, tri, _ = gs.trace_rays(g["origins"], g["dests"], g["lights"])
    same = tri == g["ref_tri"]
    assert same.mean() >= 1 - 1e-4 and np.array_equal(mt[same], g["ref_minT"][same])
    q, rq = oracle.to_bytes(np.clip(rgb, 0, 1)), oracle.to_bytes(np.clip(t["ref_rgb_tex"], 0, 1))
    diff = np.abs(q.astype(int) - rq.astype(int)).max(axis=1)
    assert (diff <= 1).mean() >= 0.999 and (diff == 0).mean() >= 0.98
    assert (np.abs(rgb - t["ref_rgb_tex"]).max(axis=1) < 1e-4).mean() >= 0.999


@pytest.mark.parametrize("name,msaa,W,H", [("ball", 1, 200, 200), ("redballs", 1, 240, 135),
                                          ("refractionballs", 1, 240, 135),
                                          ("lego", 4, 240, 135), ("final", 4, 240, 135),
                                          ("final", 16, 96, 54), ("lego", 1, 101, 67),
                                          # ragged and minimal frames: partial tiles, partial 8x8 blocks, 2x2
                                          ("ball", 4, 2, 2), ("final", 4, 33, 17), ("refractionballs", 1, 65, 3),
                                          ("lego", 16, 9, 31)])
def test_render_frame_matches_oracle(oracle, name, msaa, W, H):
    s, g = _scene(oracle, name)
    gs = gpu_scene_from_oracle(s)
    corners = oracle.camera_corners(g["modelview"], W, H)
    s.settings.MSAA = msaa
    orgb, odepth, otri = s.render(corners, W, H, threads=8)
    p = pkg().capi.default_params(W, H, msaa)
    rgb, depth, tri, st = gs.render(corners, g["lights"], p)
    assert (tri == otri).mean() >= 0.9999
    q, oq = oracle.to_bytes(rgb), oracle.to_bytes(orgb)
    diff = np.abs(q.astype(int) - oq.astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999, f"{(diff > 1).sum()} of {diff.size} pixels off by > 1 LSB"
    miss = odepth == np.float32(3.402823466e+38)
    assert np.array_equal(depth == np.float32(3.402823466e+38), miss) or \
        ((depth == np.float32(3.402823466e+38)) == miss).mean() >= 0.9999
    both = ~miss & (depth != np.float32(3.402823466e+38))
    assert np.allclose(depth[both], odepth[both], rtol=1e-5, atol=0) or \
        (np.abs(depth[both] - odepth[both]) <= 1e-5 * np.abs(odepth[both])).mean() >= 0.9995


def test_degenerate_and_extreme_triangles(oracle, tmp_path):
    """A scene made to leave the common path of the triangle test: a ray parallel to the
    triangles (determinant exactly 0: a "hit" at the ray origin, Matrix33.h:132-134),
    triangles 1e-5 and 1e4 units large (determinants far outside the range where the
    certain-rejection shortcuts apply), rays grazing edges and vertices.  GPU against the
    oracle, bit for bit.  Which triangle a determinant-0 pseudo-hit is credited to depends
    on which triangles a walk happens to test (the reference tests every triangle on the root
    paths of the leaves it visits, the GPU those of the cells the ray crosses), so for such
    rays only the hit itself is compared."""
    rng = np.random.default_rng(77)
    v = [(0, 0, 0), (1, 0, 0), (0, 1, 0),                      # unit triangle in z = 0
         (3, 0, 0), (3.00001, 0, 0), (3, 0.00001, 0),          # tiny
         (-5000, -5000, -3), (5000, -5000, -3), (0, 9000, -3),  # huge
         (0, 0, 1), (1, 0, 1), (0, 1, 1)]                      # a second unit triangle above the first
    obj = tmp_path / "odd.obj"
    obj.write_text("".join(f"v {x} {y} {z}\n" for x, y, z in v) +
                   "".join(f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}\n" for i in range(4)))
    s = oracle.Scene.load(str(obj))
    gs = gpu_scene_from_oracle(s)
    o = [[5, 5, 1.0], [0.25, 0.25, 5], [0, 0, 5], [1, 0, 5], [0.5, 0.5, 5], [3.000003, 0.000003, 4],
         [0.3, 0.3, 0.5], [0.3, 0.3, -1], [2, 2, 5]]
    d = [[6, 5, 1.0], [0.25, 0.25, -5], [0, 0, -5], [1, 0, -5], [0.5, 0.5, -5], [3.000003, 0.000003, -4],
         [0.3, 0.3, 3], [0.3, 0.3, -2], [2, 2, -5]]
    lo, hi = np.float32([-2, -2, -4]), np.float32([4, 4, 3])
    ro = (lo + (hi - lo) * rng.random((20000, 3))).astype(np.float32)
    rv = rng.normal(size=(20000, 3))
    rd = (ro + 10 * rv / np.linalg.norm(rv, axis=1, keepdims=True)).astype(np.float32)
    o = np.concatenate([np.float32(o), ro])
    d = np.concatenate([np.float32(d), rd])
    tri, a, b, t, hp = gs.check_hits(o, d)
    rtri, ra, rb, rt_, rhp = s.check_hits(o, d, threads=4)
    pseudo = (rt_ == 0) & np.signbit(rt_) & (rtri >= 0)
    assert pseudo[0] and pseudo.sum() < 10
    assert (tri[pseudo] >= 0).all() and (t[pseudo] == 0).all() and np.signbit(t[pseudo]).all()
    assert np.array_equal(hp[pseudo], rhp[pseudo])              # the ray origin
    real = ~pseudo
    assert np.array_equal(tri[real], rtri[real])
    hit = real & (rtri >= 0)
    assert hit.sum() > 1000 and len(set(rtri[hit].tolist())) == 4
    for g, r in ((a, ra), (b, rb), (t, rt_), (hp, rhp)):
        assert np.array_equal(g[hit].view(np.uint32), r[hit].view(np.uint32))


@pytest.mark.parametrize("scale", [1e-3, 1.0, 1e3])
def test_filters_of_the_frame_walk_on_fine_and_coarse_meshes(oracle, scale):
    """The frame walks put two filters before the exact ray/triangle test: the plane heads (where does the ray
    cross the triangle's plane) and the edge records (on which side of its edges), both in fused arithmetic with
    error bounds scaled by the ray and the scene.  A filter that drops a test the reference passes loses a hit.
    Here nearly every ray runs along edges: a wavy sheet of 2 x 80 x 80 small triangles over two huge ones,
    seen at a grazing angle and from above, the whole scene scaled by 1e-3, 1 and 1e3.  Pixel-centre triangle
    and depth of the frame against (1) the library's per-ray entry point on the same rays - an exact closest-hit
    search with neither filter, every pixel, bit for bit - and (2) the oracle's frame where the scene has the
    size the reference was written for: its walk steps from leaf to leaf with absolute epsilons
    (KDtree.cpp:288-587), mis-steps on a few rays per 100 000 at scale 1 and on one in twenty at scale 1000,
    where a brute-force search over all triangles in double agrees with the GPU on every pixel the two differ on."""
    n = 80
    u, v = np.meshgrid(np.linspace(-2, 2, n + 1), np.linspace(-2, 2, n + 1), indexing="ij")
    z = 0.15 * np.sin(3 * u) * np.cos(2 * v)
    sheet = np.stack([u, v, z], axis=-1).reshape(-1, 3)
    idx = lambda i, j: i * (n + 1) + j
    tris = []
    for i in range(n):
        for j in range(n):
            tris.append((idx(i, j), idx(i + 1, j), idx(i, j + 1)))
            tris.append((idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)))
    base = len(sheet)
    floor = np.float64([[-400, -400, -1], [400, -400, -1], [400, 400, -1], [-400, 400, -1]])
    vpos = (np.concatenate([sheet, floor]) * scale).astype(np.float32)
    tri_v = np.int32(tris + [(base, base + 1, base + 2), (base, base + 2, base + 3)])
    mats = np.zeros((1, 16), np.float32)
    mats[0, :10] = [0.7, 0.7, 0.7, 0.2, 0.2, 0.2, 10, 1.0, 2, 1]
    s = oracle.Scene.from_arrays(vpos, tri_v, np.zeros(len(tri_v), np.int32), mats)
    lights = np.float32([[1, 2, 5]]) * np.float32(scale)
    s.set_lights(lights)
    gs = gpu_scene_from_oracle(s)
    W, H = 300, 200
    for eye, up in (((0.0, -6.0, 0.4), (0, 0, 1)), ((0.3, 0.2, 7.0), (0, 1, 0))):     # grazing, from above
        e = np.float64(eye) * scale
        f = -e / np.linalg.norm(e)
        r = np.cross(f, np.float64(up)); r /= np.linalg.norm(r)
        w = np.cross(r, f)
        # corner rays of a pinhole camera (RayTracer::raytrace blends them, main.cpp:514-526): origins at the
        # eye, destinations on a plane 10 units (scaled) ahead; order (x, y) = (0, 0), (0, H - 1), (W - 1, 0), (W - 1, H - 1)
        L, tx, ty = 10.0 * scale, 0.45, 0.30
        corners = np.zeros(24, np.float32)
        for k, (sx, sy) in enumerate(((-1, 1), (-1, -1), (1, 1), (1, -1))):
            corners[6 * k:6 * k + 3] = e
            corners[6 * k + 3:6 * k + 6] = e + L * (f + sx * tx * r + sy * ty * w)
        grgb, gdepth, gtri, st = gs.render(corners, lights, pkg().capi.default_params(W, H, 1))
        assert (gtri >= 0).mean() > 0.5
        O, D = np.zeros((H, W, 3), np.float32), np.zeros((H, W, 3), np.float32)
        for y in range(H):
            for x in range(W):
                O[y, x], D[y, x] = oracle.pixel_ray(corners, W, H, x, y)
        ctri, _, _, ct, _ = gs.check_hits(O.reshape(-1, 3), D.reshape(-1, 3))
        ctri, ct = ctri.reshape(H, W), ct.reshape(H, W)
        assert np.array_equal(gtri, ctri), (scale, eye, (gtri != ctri).sum())
        hit = ctri >= 0
        assert np.array_equal(gdepth[hit].view(np.uint32), ct[hit].view(np.uint32))
        assert (gdepth[~hit] == np.float32(3.402823466e+38)).all()
        if scale <= 1.0:
            frgb, fdepth, ftri = s.render(corners, W, H, threads=8)
            same = gtri == ftri
            assert same.mean() >= 0.9998, (scale, eye, (~same).sum())
            assert np.array_equal(gdepth[same].view(np.uint32), fdepth[same].view(np.uint32))


def test_hard_bright_highlights_through_the_frame_pipeline(oracle):
    """The frame pipeline forms the specular power as 2^(n log2 x) with the hardware's log2 and exp2 and
    normalises the view and halfway vectors with its reciprocal square root (rt_wavefront.h: wf_pow, wf_unit);
    the reference calls std::pow on exactly normalised vectors (raytracing.cpp:131-134).  The relative error of
    the power grows with the hardness n.  A wavy sheet with n = 1000 and a specular colour of 3 (brighter than
    any of the reference's materials: highlights clip) under three lights, against the oracle: float colours
    within 5e-4 of a colour unit (an eighth of a byte step), bytes within 1 LSB everywhere."""
    n = 60
    u, v = np.meshgrid(np.linspace(-2, 2, n + 1), np.linspace(-2, 2, n + 1), indexing="ij")
    z = 0.25 * np.sin(2.5 * u) * np.cos(2 * v)
    vpos = np.stack([u, v, z], axis=-1).reshape(-1, 3).astype(np.float32)
    idx = lambda i, j: i * (n + 1) + j
    tris = []
    for i in range(n):
        for j in range(n):
            tris.append((idx(i, j), idx(i + 1, j), idx(i, j + 1)))
            tris.append((idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)))
    mats = np.zeros((1, 16), np.float32)
    mats[0, :10] = [0.05, 0.05, 0.08, 3.0, 3.0, 3.0, 1000, 1.0, 2, 1]
    s = oracle.Scene.from_arrays(vpos, np.int32(tris), np.zeros(len(tris), np.int32), mats)
    lights = np.float32([[0.5, 0.5, 4], [-3, 1, 2], [2, -3, 3]])
    s.set_lights(lights)
    gs = gpu_scene_from_oracle(s)
    W, H = 320, 240
    e = np.float64([0.2, -2.5, 3.5])
    f = -e / np.linalg.norm(e)
    r = np.cross(f, np.float64([0, 0, 1])); r /= np.linalg.norm(r)
    w = np.cross(r, f)
    corners = np.zeros(24, np.float32)
    for k, (sx, sy) in enumerate(((-1, 1), (-1, -1), (1, 1), (1, -1))):
        corners[6 * k:6 * k + 3] = e
        corners[6 * k + 3:6 * k + 6] = e + 10.0 * (f + sx * 0.5 * r + sy * 0.375 * w)
    frgb, fdepth, ftri = s.render(corners, W, H, threads=8)
    grgb, gdepth, gtri, st = gs.render(corners, lights, pkg().capi.default_params(W, H, 1))
    same = (gtri == ftri) & (ftri >= 0)
    assert same.mean() > 0.5
    lit = same & (frgb.max(axis=2) > 0.2)            # pixels inside a highlight
    assert lit.sum() > 500
    assert np.abs(grgb[same] - frgb[same]).max() < 5e-4
    d = np.abs(oracle.to_bytes(grgb[same]).astype(int) - oracle.to_bytes(frgb[same]).astype(int))
    assert d.max() <= 1


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_scenes_match_oracle(oracle, seed):
    """Scenes no fixture knows: a few hundred random triangles (a soup: they intersect each
    other, many share no vertex) with diffuse, mirror and refractive materials, four lights.
    Closest hits bit for bit, ray colours and a small MSAA frame within the 8-bit bar."""
    rng = np.random.default_rng(seed)
    n_tri = 300 + 100 * (seed % 3)
    centre = rng.uniform(-2, 2, (n_tri, 1, 3))
    size = rng.choice([0.05, 0.3, 1.5], (n_tri, 1, 1))
    vpos = (centre + size * rng.normal(size=(n_tri, 3, 3))).reshape(-1, 3).astype(np.float32)
    floor = np.float32([[-6, -6, -2.5], [6, -6, -2.5], [6, 6, -2.5], [-6, 6, -2.5]])
    vpos = np.concatenate([vpos, floor])
    tri_v = np.concatenate([np.arange(3 * n_tri).reshape(-1, 3), 3 * n_tri + np.int32([[0, 1, 2], [0, 2, 3]])])
    mats = np.zeros((4, 16), np.float32)
    #          Kd               Ks               Ns  Ni   illum has_Ni
    mats[0, :10] = [0.8, 0.3, 0.2, 0.5, 0.5, 0.5, 40, 1.0, 2, 1]      # diffuse
    mats[1, :10] = [0.2, 0.2, 0.9, 1.0, 1.0, 1.0, 96, 1.0, 3, 1]      # mirror
    mats[2, :10] = [0.9, 0.9, 0.9, 0.3, 0.3, 0.3, 20, 1.3, 2, 1]      # refractive
    mats[3, :10] = [0.3, 0.8, 0.3, 0.2, 0.2, 0.2, 0, 1.0, 2, 0]       # no Ni line, hardness 0: the specular power is 1
    tri_mat = np.concatenate([rng.integers(0, 4, n_tri), [0, 3]]).astype(np.int32)
    s = oracle.Scene.from_arrays(vpos, tri_v, tri_mat, mats)
    lights = np.float32([[0, 0, 6], [4, -3, 5], [-5, 2, 1], [1, 5, -1]])
    s.set_lights(lights)
    gs = gpu_scene_from_oracle(s)
    o = rng.uniform(-5, 5, (30000, 3)).astype(np.float32)
    v = rng.normal(size=(30000, 3))
    d = (o + 10 * v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    tri, a, b, t, hp = gs.check_hits(o, d)
    rtri, ra, rb, rt_, rhp = s.check_hits(o, d, threads=4)
    same = tri == rtri
    assert same.mean() >= 0.9999
    hit = same & (rtri >= 0)
    assert hit.mean() > 0.3
    for g, r in ((a, ra), (b, rb), (t, rt_), (hp, rhp)):
        assert np.array_equal(g[hit].view(np.uint32), r[hit].view(np.uint32))
    rgb, mt, tri2, _ = gs.trace_rays(o[:8000], d[:8000], lights)
    orgb, omt, otri = s.trace_rays(o[:8000], d[:8000], threads=4)
    ok = tri2 == otri
    assert ok.mean() >= 0.9999
    q, oq = oracle.to_bytes(np.clip(rgb[ok], 0, 1)), oracle.to_bytes(np.clip(orgb[ok], 0, 1))
    assert (np.abs(q.astype(int) - oq.astype(int)) <= 1).mean() >= 0.999
    # a frame through the wavefront pipeline (refraction: two children per hit)
    W, H = 120, 80
    m = np.float64([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, -7, 1])
    corners = oracle.camera_corners(m, W, H)
    s.settings.MSAA = 4
    frgb, fdepth, ftri = s.render(corners, W, H, threads=8)
    grgb, gdepth, gtri, st = gs.render(corners, lights, pkg().capi.default_params(W, H, 4))
    assert (gtri == ftri).mean() >= 0.999
    diff = np.abs(oracle.to_bytes(grgb).astype(int) - oracle.to_bytes(frgb).astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999, f"{(diff > 1).sum()} of {diff.size} pixels off by > 1 LSB"
    assert st.rays_closest > W * H


@pytest.mark.parametrize("name,n,msaa,W,H", [("redballs", 2, 1, 120, 68), ("refractionballs", 2, 4, 96, 54),
                                             ("final", 3, 4, 64, 36), ("ball", 4, 1, 64, 64)])
def test_soft_shadows_match_oracle(oracle, name, n, msaa, W, H):
    """shadowSamples = n > 1 (raytracing.cpp:145-160): up to n^3 jittered shadow rays per light,
    the jitter from the counter-based generator both sides share (the reference's rand() stream
    is not reproduced).  Frames through the wavefront pipeline and batched rays through
    rt_trace_rays."""
    capi = pkg().capi
    s = oracle.Scene.load(os.path.join(SCENES, name + ".obj"))
    m, lights = oracle.read_camera_sav(os.path.join(SCENES, "cameras", name + ".sav"))
    s.set_lights(lights)
    s.settings.MSAA = msaa
    s.settings.shadowSamples = n
    gs = gpu_scene_from_oracle(s)
    corners = oracle.camera_corners(m, W, H)
    p = capi.default_params(W, H, msaa)
    p.shadowSamples = n
    rgb, depth, tri, st = gs.render(corners, lights, p)
    c = oracle.Counters()
    orgb, odepth, otri = s.render(corners, W, H, threads=8, counters=c)
    FLT_MAX = np.float32(3.402823466e+38)
    same_tri = tri == otri
    assert same_tri.mean() >= 0.9999
    assert ((depth == FLT_MAX) == (odepth == FLT_MAX))[same_tri].all()   # the same pixels miss
    if msaa == 1:                                            # depth of the camera ray, bit for bit
        assert np.array_equal(depth[same_tri].view(np.uint32), odepth[same_tri].view(np.uint32))
    else:                                                    # mean over the samples that hit
        hit = same_tri & (odepth != FLT_MAX)
        assert np.isclose(depth[hit], odepth[hit], rtol=1e-5, atol=0).mean() >= 0.999
    d = np.abs(oracle.to_bytes(rgb).astype(int) - oracle.to_bytes(orgb).astype(int)).max(axis=2)
    assert (d <= 1).mean() >= 0.999, (d > 1).sum()
    x, delta, loops = np.float32(-1), np.float32(2) / np.float32(n - 1), 0
    while x <= np.float32(1):                                # the reference's loop, in float
        x, loops = np.float32(x + delta), loops + 1
    assert st.rays_shadow % loops ** 3 == 0 and st.rays_shadow > 0
    if msaa == 1:                                            # no query is elided with soft shadows
        assert abs(st.rays_shadow - c.rays_shadow) <= 1e-3 * c.rays_shadow
    else:                                                    # (the discarded pixel-centre subtrees are not traced)
        assert st.rays_shadow <= c.rays_shadow
    hard = gs.render(corners, lights, capi.default_params(W, H, msaa))
    assert not np.array_equal(hard[0], rgb)                    # penumbrae differ from hard shadows
    # the same through the batched entry point: ray i is keyed by its index
    g = load_golden(name)
    o, dd = g["origins"][:2000], g["dests"][:2000]
    p1 = capi.default_params(64, 64)
    p1.shadowSamples = n
    grgb, gmt, gtri, _ = gs.trace_rays(o, dd, lights, params=p1)
    s.settings.MSAA = 1
    rrgb, rmt, rtri = s.trace_rays(o, dd, threads=8)
    assert (gtri == rtri).mean() >= 0.9999
    same = gtri == rtri
    assert np.abs(grgb[same] - rrgb[same]).max() < 2e-5
    p1.shadowSamples = 7
    with pytest.raises(capi.RtError, match="shadowSamples"):
        gs.trace_rays(o[:1], dd[:1], lights, params=p1)


def _zoom_corners(eye, half, z_plane=0.0):
    """The four corner rays of a camera at `eye` looking at the square [-half, half]^2 of the plane
    z = z_plane, in the layout of RayTracer::raytrace (main.cpp:514-526): origin and destination of the
    corners (0,0), (0,H-1), (W-1,0), (W-1,H-1)."""
    c = []
    for x, y in ((-half, half), (-half, -half), (half, half), (half, -half)):
        c += list(eye) + [x, y, z_plane]
    return np.float32(c)


def test_distant_camera_matches_oracle(oracle):
    """A camera a thousand mesh sizes away: the cell margins of the stack traversal are relative to
    the ray parameter, whose rounding error grows with the distance of the origin.  The frame through
    the wavefront pipeline still equals the oracle's: same triangles, colours within the 8-bit bar."""
    s, g = _scene(oracle, "ball")
    gs = gpu_scene_from_oracle(s)
    W, H = 160, 160
    for eye in ([0.3, 0.2, 1000.0], [900.0, -400.0, 300.0]):
        corners = _zoom_corners(eye, 0.8)
        s.settings.MSAA = 1
        orgb, odepth, otri = s.render(corners, W, H, threads=8)
        rgb, depth, tri, _ = gs.render(corners, g["lights"], pkg().capi.default_params(W, H, 1))
        assert (otri >= 0).mean() > 0.05
        assert (tri == otri).mean() >= 0.9999
        same = (tri == otri) & (otri >= 0)
        assert np.array_equal(depth[same], odepth[same])
        diff = np.abs(oracle.to_bytes(rgb).astype(int) - oracle.to_bytes(orgb).astype(int)).max(axis=2)
        assert (diff <= 1).mean() >= 0.999


def test_equal_t_hits_in_different_leaves(oracle):
    """Every triangle twice, at the same place: each hit is a tie between two triangles, and a
    ray may meet the pair in leaves the two walks visit in a different order.  The reference keeps the
    triangle its walk tests last (raytracing.cpp:323-349), the GPU the one with the larger rank
    (breadth-first order of the reference's tree); for triangles on one root-to-leaf path these are the
    same, across leaves they need not be.  What must hold: the same hit (t, hit point) bit for bit,
    and the triangle is one of the twins."""
    rng = np.random.default_rng(5)
    n = 120
    centre = rng.uniform(-1.5, 1.5, (n, 1, 3))
    v = (centre + 0.6 * rng.normal(size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
    vpos = np.concatenate([v, v])
    tri_v = np.arange(6 * n).reshape(-1, 3).astype(np.int32)
    mats = np.zeros((1, 16), np.float32)
    mats[0, :10] = [0.8, 0.3, 0.2, 0.5, 0.5, 0.5, 40, 1.0, 2, 1]
    s = oracle.Scene.from_arrays(vpos, tri_v, np.zeros(2 * n, np.int32), mats)
    gs = gpu_scene_from_oracle(s)
    o = rng.uniform(-4, 4, (20000, 3)).astype(np.float32)
    d = rng.uniform(-1, 1, (20000, 3)).astype(np.float32)
    tri, a, b, t, hp = gs.check_hits(o, d)
    rtri, ra, rb, rt_, rhp = s.check_hits(o, d, threads=4)
    hit = rtri >= 0
    assert hit.mean() > 0.2 and np.array_equal(tri >= 0, hit)
    assert np.array_equal(t[hit].view(np.uint32), rt_[hit].view(np.uint32))
    assert np.array_equal(hp[hit].view(np.uint32), rhp[hit].view(np.uint32))
    assert np.array_equal(tri[hit] % n, rtri[hit] % n)          # the same pair
    agree = (tri[hit] == rtri[hit]).mean()
    print(f"twin chosen as by the reference's walk: {agree:.4f}")
    # the same scene as a frame through the wavefront pipeline
    W, H = 96, 96
    corners = _zoom_corners([0.0, 0.0, 9.0], 2.5)
    s.set_lights(np.float32([[0, 0, 6]]))
    s.settings.MSAA = 1
    orgb, odepth, otri = s.render(corners, W, H, threads=8)
    rgb, depth, ftri, _ = gs.render(corners, np.float32([[0, 0, 6]]), pkg().capi.default_params(W, H, 1))
    both = (otri >= 0) & (ftri >= 0)
    assert np.array_equal(otri >= 0, ftri >= 0)
    assert np.array_equal(ftri[both] % n, otri[both] % n)
    assert np.array_equal(depth[both], odepth[both])


def test_frame_without_lights(oracle):
    """No light at all: the ambient term alone (raytracing.cpp:92-95; the shadow share
    ShadowFactor / 0 of raytracing.cpp:102 is never used)."""
    s, g = _scene(oracle, "redballs")
    s.set_lights(np.zeros((0, 3), np.float32))
    gs = gpu_scene_from_oracle(s)
    W, H = 96, 54
    corners = oracle.camera_corners(g["modelview"], W, H)
    s.settings.MSAA = 4
    orgb, odepth, otri = s.render(corners, W, H, threads=8)
    rgb, depth, tri, st = gs.render(corners, np.zeros((0, 3), np.float32), pkg().capi.default_params(W, H, 4))
    assert np.isfinite(rgb).all() and st.rays_shadow == 0
    assert (tri == otri).mean() >= 0.9999
    diff = np.abs(oracle.to_bytes(rgb).astype(int) - oracle.to_bytes(orgb).astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999
