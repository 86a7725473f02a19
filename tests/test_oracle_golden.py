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
