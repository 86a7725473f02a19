"""GPU tests of the frame-level path through the product's own host code:
post-process kernels, tile partition, the 'r' path's files, full-size spot
checks against the committed sparse fixtures, and error handling."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCENES, pkg

pytestmark = pytest.mark.gpu
FLT_MAX = np.float32(3.402823466e+38)


@pytest.fixture(scope="module")
def P():
    return pkg()


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _host_gpu(P, name):
    hs = P.host.HostScene(os.path.join(SCENES, name + ".obj"))
    m, lights = P.host.read_camera_sav(os.path.join(SCENES, "cameras", name + ".sav"))
    return hs, hs.to_gpu(0), m, lights


def test_scene_info(P):
    hs, gs, _, _ = _host_gpu(P, "final")
    i = gs.info()
    # the device tree = the reference's 891 nodes / depth 20, refined below its leaves
    assert i["n_triangles"] == 3846 and i["n_nodes"] >= 891 and 20 <= i["tree_depth"] <= 63
    assert i["n_slots"] >= 3846          # straddling triangles are listed by every leaf they touch
    assert i["device_bytes"] > 3846 * 64
    assert i["render_path"] == 0         # the single persistent launch over ray queues


def test_scene_beyond_65535_nodes_renders_on_the_pipeline(P, oracle, monkeypatch):
    """fin.obj (13444 triangles) refined down to leaves of four triangles: the device tree has more nodes
    than a 16-bit index names; traversal stack entries carry 32-bit nodes, so the scene stays on the
    ray-queue pipeline, with its refinement run to the end.  A reduced frame against the oracle, refraction
    included."""
    monkeypatch.setenv("RT_KD_LEAF", "4")      # (read when the scene is created; the default leaves are larger)
    hs, gs, m, lights = _host_gpu(P, "fin")
    monkeypatch.delenv("RT_KD_LEAF")
    i = gs.info()
    assert i["n_nodes"] > 65535 and i["render_path"] == 0 and i["n_triangles"] == 13444
    W, H = 240, 135
    corners = P.host.camera_corners(m, W, H)
    rgb, depth, tri, st = gs.render(corners, lights, P.capi.default_params(W, H, 4))
    s = oracle.Scene.load(os.path.join(SCENES, "fin.obj"))
    s.set_lights(lights)
    s.settings.MSAA = 4
    orgb, odepth, otri = s.render(corners, W, H, threads=8)
    assert (tri == otri).mean() >= 0.9999
    d = np.abs(oracle.to_bytes(rgb).astype(int) - oracle.to_bytes(orgb).astype(int)).max(axis=2)
    assert (d <= 1).mean() >= 0.999
    assert st.child_rays > 0


@pytest.mark.parametrize("name", ["ball", "redballs", "lego", "final", "fin", "checkercube"])
def test_tree_built_on_the_gpu_equals_the_host_built_one(P, monkeypatch, name):
    """rt_scene_create hands the inner nodes' triangles down and splits large leaves on the
    device; RT_KD_GPU=0 runs the same two steps as host loops.  The device tree must be the same
    one: equal hash over nodes and slots, equal sizes - also with the refinement switched off."""
    hs = P.host.HostScene(os.path.join(SCENES, name + ".obj"))
    for env in ({}, {"RT_KD_REFINE": "0"}):
        infos = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("RT_KD_GPU", mode)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            infos[mode] = hs.to_gpu(0).info()
            for k in env:
                monkeypatch.delenv(k)
        monkeypatch.delenv("RT_KD_GPU")
        g, h = infos["1"], infos["0"]
        assert g["tree_on_gpu"] == 1 and h["tree_on_gpu"] == 0
        for key in ("tree_hash", "n_nodes", "n_slots", "tree_depth", "device_bytes"):
            assert g[key] == h[key], (name, env, key, g[key], h[key])
        print(f"{name} {env}: {g['n_nodes']} nodes; tree steps {g['tree_ms']:.1f} ms on the GPU, "
              f"{h['tree_ms']:.1f} ms on the host; rt_scene_create {g['create_ms']:.1f} / {h['create_ms']:.1f} ms")


@pytest.mark.parametrize("name", ["ball", "redballs", "refractionballs", "lego", "final", "fin", "checkercube"])
def test_reference_tree_built_on_the_gpu(P, monkeypatch, name):
    """rt_build_kd_tree: buildKDtree / splitSpaces / calcBestSplit (KDtree.cpp:15-230) on the device.  The tree
    must be (1) the one the real reference built for the scene - bounds, leaves, the triangles every node
    keeps and their order, in the preorder the reference was dumped in (tests/golden/<scene>.npz, for the five
    scenes that have a fixture) - and (2) the one the host loops build, node for node, also when the C++ host
    is told to take its tree from the device (RT_KD_BUILD_GPU=1)."""
    monkeypatch.delenv("RT_KD_BUILD_GPU", raising=False)
    hs = P.host.HostScene(os.path.join(SCENES, name + ".obj"))
    a = hs.arrays()
    t = P.capi.build_kd_tree(a["vertex_pos"], a["tri_vertex"])
    assert t["depth"] == a["tree_depth"]
    for mine, host in (("bounds", "node_bounds"), ("left", "node_left"), ("right", "node_right"),
                       ("tri_off", "node_tri_off"), ("tri_cnt", "node_tri_cnt"), ("tris", "node_tris")):
        assert np.array_equal(t[mine].view(np.uint32) if mine == "bounds" else t[mine],
                              a[host].view(np.uint32) if mine == "bounds" else a[host]), (name, mine)
    gold = os.path.join(os.path.dirname(__file__), "golden", name + ".npz")
    if os.path.exists(gold):
        g = np.load(gold)
        order, stack = [], [0]
        while stack:                                  # preorder, left first: the order of the reference's dump
            n = stack.pop()
            order.append(n)
            if t["left"][n] >= 0:
                stack += [t["right"][n], t["left"][n]]
        order = np.array(order)
        assert len(order) == len(g["ref_tree_leaf"])
        assert np.array_equal(t["bounds"][order], g["ref_tree_bounds"])
        assert np.array_equal((t["left"][order] < 0).astype(np.int32), g["ref_tree_leaf"])
        assert np.array_equal(t["tri_cnt"][order], g["ref_tree_tri_cnt"])
        mine = np.concatenate([t["tris"][t["tri_off"][n]:t["tri_off"][n] + t["tri_cnt"][n]] for n in order])
        assert np.array_equal(mine, g["ref_tree_tris"])
    monkeypatch.setenv("RT_KD_BUILD_GPU", "1")
    b = P.host.HostScene(os.path.join(SCENES, name + ".obj")).arrays()
    monkeypatch.delenv("RT_KD_BUILD_GPU")
    for key in ("node_bounds", "node_left", "node_right", "node_tri_off", "node_tri_cnt", "node_tris", "tree_depth"):
        assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), (name, key)


def test_normalize_depth_matches_oracle(P, oracle):
    rng = np.random.default_rng(2)
    d = (rng.random(100003, dtype=np.float32) * 0.9 + 0.05).astype(np.float32)
    d[rng.integers(0, d.size, 500)] = FLT_MAX
    for focus in (0.0, 0.37):
        assert np.array_equal(P.capi.normalize_depth(d, focus), oracle.normalize_depth(d, focus))
    once = P.capi.normalize_depth(d, 0.37)
    assert np.array_equal(P.capi.normalize_depth(once, 0.37), oracle.normalize_depth(once, 0.37))
    allmiss = np.full(10, FLT_MAX, np.float32)
    assert (P.capi.normalize_depth(allmiss) == 0).all()


def test_dof_blur_matches_oracle(P, oracle):
    rng = np.random.default_rng(4)
    W, H = 96, 70
    rgb = rng.random((H, W, 3), dtype=np.float32)
    dn = rng.random((H, W), dtype=np.float32)
    dn[10:20, 10:30] = 0.0       # in focus: identity kernel
    dn[40, 50] = 1.0             # farthest pixel: NaN kernel
    got, ref = P.capi.dof_blur(rgb, dn), oracle.blur(rgb, dn, threads=4)
    assert got.shape == ref.shape == (H - 12, W - 12, 3)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(got[34, 44]).all()
    ok = ~np.isnan(ref)
    assert np.abs(got[ok] - ref[ok]).max() < 2e-6        # powf differs by ulps between the two
    assert np.array_equal(got[4:14, 4:24], rgb[10:20, 10:30])
    q, oq = P.capi.quantize(got), oracle.to_bytes(ref)
    assert (np.abs(q.astype(int) - oq.astype(int)) <= 1).all() and (q == oq).mean() > 0.995


def test_quantize_matches_oracle(P, oracle):
    v = np.float32([0, 0.999, 1.0, 0.5, np.nan, 3e9, -3e9, 1 / 255, 0.0039215684])
    assert np.array_equal(P.capi.quantize(v), oracle.to_bytes(v))
    r = np.random.default_rng(1).random(10000, dtype=np.float32)
    assert np.array_equal(P.capi.quantize(r), oracle.to_bytes(r))


def test_tiles_on_device_match_numpy(P):
    import torch
    W, H, tw, th, world = 100, 70, 16, 8, 3
    rng = np.random.default_rng(9)
    rgb = rng.random((H, W, 3), dtype=np.float32)
    depth = rng.random((H, W), dtype=np.float32)
    d_rgb, d_depth = torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda()
    out_rgb, out_depth = torch.zeros_like(d_rgb), torch.zeros_like(d_depth)
    L = P.capi.lib()
    p = P.capi.default_params(W, H)
    p.tile_w, p.tile_h, p.world = tw, th, world
    for r in range(world):
        p.rank = r
        n = L.rt_tile_pixels(C.byref(p), r)
        pay = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
        P.capi.check(L.rt_pack_tiles_device(0, C.byref(p), d_rgb.data_ptr(), d_depth.data_ptr(),
                                            pay.data_ptr(), None))
        torch.cuda.synchronize()
        assert np.array_equal(pay.cpu().numpy(), P.tiles.pack(rgb, depth, tw, th, r, world))
        P.capi.check(L.rt_unpack_tiles_device(0, C.byref(p), r, pay.data_ptr(), out_rgb.data_ptr(),
                                              out_depth.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(out_rgb.cpu().numpy(), rgb) and np.array_equal(out_depth.cpu().numpy(), depth)


@pytest.mark.parametrize("name,msaa", [("refractionballs", 1), ("final", 4)])
def test_partitioned_render_equals_whole_frame(P, name, msaa):
    """Two ranks' tiles rendered one after the other into one frame buffer."""
    import torch
    hs, gs, m, lights = _host_gpu(P, name)
    W, H = 200, 120
    corners = P.host.camera_corners(m, W, H)
    whole = gs.render(corners, lights, P.capi.default_params(W, H, msaa))
    rgb = torch.zeros((H, W, 3), device="cuda")
    depth = torch.zeros((H, W), device="cuda")
    tri = torch.full((H, W), -7, dtype=torch.int32, device="cuda")
    rays = 0
    for r in range(2):
        p = P.capi.default_params(W, H, msaa)
        p.tile_w, p.tile_h, p.rank, p.world = 32, 16, r, 2
        st = gs.render_device(corners, lights, p, rgb.data_ptr(), depth.data_ptr(), tri.data_ptr(),
                              0, stats=True)
        rays += st.rays_closest + st.rays_shadow
        if r == 0:
            assert (tri.cpu().numpy() == -7).sum() > 0      # the other rank's pixels untouched
    assert np.array_equal(rgb.cpu().numpy(), whole[0]) and np.array_equal(depth.cpu().numpy(), whole[1])
    assert np.array_equal(tri.cpu().numpy(), whole[2])
    assert rays == whole[3].rays_closest + whole[3].rays_shadow


def test_uneven_deal_render_pack_unpack(P):
    """Three GPUs' worth of an uneven deal (rank_span), rendered one after the other, packed
    and unpacked through the tile payloads: the frame equals the one-piece render."""
    import torch
    hs, gs, m, lights = _host_gpu(P, "final")
    W, H, msaa = 200, 136, 4
    corners = P.host.camera_corners(m, W, H)
    whole = gs.render(corners, lights, P.capi.default_params(W, H, msaa))
    world, deal = P.tiles.shares(3, 0.375, per_rank=8)
    assert deal[0] == (0, 3) and world == 19
    out_rgb = torch.zeros((H, W, 3), device="cuda")
    out_depth = torch.zeros((H, W), device="cuda")
    L = P.capi.lib()
    for first, span in deal:
        p = P.capi.default_params(W, H, msaa)
        p.tile_w, p.tile_h, p.rank, p.world, p.rank_span = 16, 8, first, world, span
        rgb = torch.zeros((H, W, 3), device="cuda")
        depth = torch.zeros((H, W), device="cuda")
        gs.render_device(corners, lights, p, rgb.data_ptr(), depth.data_ptr(), 0, 0)
        n = L.rt_tile_pixels(C.byref(p), first)
        assert n == P.tiles.n_my_tiles(W, H, 16, 8, first, world, span) * 128
        payload = torch.zeros((n, 4), device="cuda")
        P.capi.check(L.rt_pack_tiles_device(0, C.byref(p), C.c_void_p(rgb.data_ptr()), C.c_void_p(depth.data_ptr()),
                                            C.c_void_p(payload.data_ptr()), None))
        P.capi.check(L.rt_unpack_tiles_device(0, C.byref(p), first, C.c_void_p(payload.data_ptr()),
                                              C.c_void_p(out_rgb.data_ptr()), C.c_void_p(out_depth.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.array_equal(out_rgb.cpu().numpy(), whole[0]) and np.array_equal(out_depth.cpu().numpy(), whole[1])


def test_r_path_files_match_oracle_writers(P, oracle, tmp_path):
    """rt_render's path: init() + camera.sav + doDaRayTracingShizz -> four files."""
    name, W, H, msaa = "final", 160, 100, 4
    obj, sav = os.path.join(SCENES, name + ".obj"), os.path.join(SCENES, "cameras", name + ".sav")
    secs, st = P.host.render_to_files(obj, sav, W, H, msaa, str(tmp_path))
    assert secs > 0 and st.rays_closest > 0
    s = oracle.Scene.load(obj)
    m, lights = oracle.read_camera_sav(sav)
    s.set_lights(lights)
    s.settings.MSAA = msaa
    rgb, depth, _ = s.render(oracle.camera_corners(m, W, H), W, H, threads=8)
    ref = tmp_path / "ref"
    ref.mkdir()
    oracle.write_result_files(str(ref), rgb, depth, 0.0, threads=8)
    for f in ("result.ppm", "result.bmp", "depth.bmp", "blurred.bmp"):
        a = np.frombuffer((tmp_path / f).read_bytes(), np.uint8)
        b = np.frombuffer((ref / f).read_bytes(), np.uint8)
        assert a.size == b.size, f
        hdr = 54 if f.endswith(".bmp") else 15
        assert np.array_equal(a[:hdr], b[:hdr]), f
        d = np.abs(a[hdr:].astype(int) - b[hdr:].astype(int))
        assert (d <= 1).mean() >= 0.999 and (d == 0).mean() >= 0.99, (f, d.max(), (d > 1).sum())


def test_r_path_float_planes_and_row_updates(P, tmp_path):
    """The 'r' path keeps the frame on the GPU and brings back 8-bit planes; the variant that
    brings the float planes back (and rewrites result.bmp row by row first, main.cpp:325, 407)
    must write the same four files, byte for byte - ragged width, so rows carry padding."""
    name, W, H, msaa = "refractionballs", 150, 91, 4
    obj, sav = os.path.join(SCENES, name + ".obj"), os.path.join(SCENES, "cameras", name + ".sav")
    a, b = tmp_path / "bytes", tmp_path / "floats"
    a.mkdir(), b.mkdir()
    P.host.render_to_files(obj, sav, W, H, msaa, str(a))
    P.host.render_to_files_float(obj, sav, W, H, msaa, str(b), row_updates=True)
    for f in ("result.ppm", "result.bmp", "depth.bmp", "blurred.bmp"):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f
    t = P.host.time_r_path(obj, sav, W, H, msaa, str(a), runs=2)
    assert t["frame_call_ms"] > 0 and t["file_writes_ms"] > 0


def test_result_planes_of_frames_with_equal_area_and_different_shape(P):
    """rt_render_result_planes keeps its frame buffers between calls.  Their size depends on the frame's
    shape, not only on its pixel count: 2 x 5000 has no blurred plane, 100 x 100 has one of 88 x 88 -
    a square frame after a strip of the same area must get buffers of its own size, and every plane
    must equal the one a fresh scene renders."""
    hs, gs, m, lights = _host_gpu(P, "ball")
    out = {}
    for W, H in ((2, 5000), (100, 100), (5000, 2), (100, 100), (64, 30), (30, 64)):
        corners = P.host.camera_corners(m, W, H)
        out[(W, H)] = gs.result_planes(corners, lights, P.capi.default_params(W, H, 1))
    for (W, H), (rgb8, depth8, blur8, st) in out.items():
        fresh = hs.to_gpu(0)
        f_rgb8, f_depth8, f_blur8, _ = fresh.result_planes(P.host.camera_corners(m, W, H), lights,
                                                           P.capi.default_params(W, H, 1))
        assert np.array_equal(rgb8, f_rgb8) and np.array_equal(depth8, f_depth8), (W, H)
        assert (blur8 is None) == (f_blur8 is None) == (W <= 12 or H <= 12)
        if blur8 is not None:
            assert np.array_equal(blur8, f_blur8), (W, H)
        fresh.close()
    assert out[(100, 100)][0].std() > 0


@pytest.mark.parametrize("name,W,H,msaa,n_ranks,first", [("refractionballs", 200, 120, 4, 3, 5), ("final", 330, 190, 1, 8, 11),
                                                         ("ball", 96, 64, 4, 2, 16), ("lego", 70, 45, 1, 4, 1)])
def test_shared_frame_rehearsal_with_an_uneven_deal(P, name, W, H, msaa, n_ranks, first):
    """rt_render_result_planes_shared without the transfer: every rank's part of the scene's deal
    (rank 0 owns `first` of 16 shares per GPU) rendered here in turn, packed, laid out as the gather lays
    the payloads out, unpacked into rank 0's frame and post-processed.  The three planes must equal
    those of the undivided frame byte for byte, and the parts must trace the frame's rays between them."""
    hs, gs, m, lights = _host_gpu(P, name)
    corners = P.host.camera_corners(m, W, H)
    p = P.capi.default_params(W, H, msaa)
    rgb8, depth8, blur8, st = gs.result_planes(corners, lights, p)
    r_rgb8, r_depth8, r_blur8, r_st = gs.result_planes_rehearsal(n_ranks, first, corners, lights, p)
    assert np.array_equal(rgb8, r_rgb8) and np.array_equal(depth8, r_depth8) and np.array_equal(blur8, r_blur8)
    assert (r_st.rays_closest, r_st.rays_shadow, r_st.camera_rays) == (st.rays_closest, st.rays_shadow, st.camera_rays)
    assert rgb8.std() > 0
    L = P.capi.lib()
    assert L.rt_scene_set_deal(gs.h, 0) == -1 and L.rt_scene_set_deal(gs.h, 17) == -1
    got = C.c_int32(-1)            # one rank: nothing to measure, an even deal
    corners = np.ascontiguousarray(corners, np.float32)
    P.capi.check(L.rt_calibrate_deal(gs.h, None, 0, 1, corners.ctypes.data_as(C.c_void_p),
                                     np.ascontiguousarray(lights, np.float32).ctypes.data_as(C.c_void_p), len(lights),
                                     C.byref(p), C.byref(got)))
    assert got.value == 16
    assert L.rt_calibrate_deal(gs.h, None, 0, 2, corners.ctypes.data_as(C.c_void_p), None, 0, C.byref(p), C.byref(got)) == -1


def test_focus_pick_then_r_path(P, oracle, tmp_path):
    """Key 'f' over a pixel, then 'r': a focus depth other than 0 makes the two depth
    normalisations differ (imageWriter.h:124-135, 352, 416); files against the oracle's."""
    name, W, H, msaa = "final", 150, 90, 1
    fx, fy = 60.0, 55.0
    obj, sav = os.path.join(SCENES, name + ".obj"), os.path.join(SCENES, "cameras", name + ".sav")
    secs, focus, st = P.host.render_to_files_focus(obj, sav, W, H, msaa, str(tmp_path), fx, fy)
    s = oracle.Scene.load(obj)
    m, lights = oracle.read_camera_sav(sav)
    s.set_lights(lights)
    s.settings.MSAA = msaa
    corners = oracle.camera_corners(m, W, H)
    o, d = oracle.pixel_ray(corners, W, H, fx, fy)
    _, min_t, tri = s.trace_rays(o[None], d[None])
    assert tri[0] >= 0 and focus == min_t[0] and 0 < focus < 1     # bit-equal depth of the picked point
    rgb, depth, _ = s.render(corners, W, H, threads=8)
    ref = tmp_path / "ref"
    ref.mkdir()
    dn, dn2, _ = oracle.write_result_files(str(ref), rgb, depth, focus, threads=8)
    assert not np.array_equal(dn, dn2)
    for f in ("depth.bmp", "blurred.bmp"):
        a = np.frombuffer((tmp_path / f).read_bytes(), np.uint8)
        b = np.frombuffer((ref / f).read_bytes(), np.uint8)
        assert a.size == b.size and np.array_equal(a[:54], b[:54]), f
        dd = np.abs(a[54:].astype(int) - b[54:].astype(int))
        assert (dd <= 1).mean() >= 0.999 and (dd == 0).mean() >= 0.99, (f, dd.max(), (dd > 1).sum())


FC = json.load(open(os.path.join(GOLDEN, "frame_counters.json"))) \
    if os.path.exists(os.path.join(GOLDEN, "frame_counters.json")) else {}


@pytest.mark.parametrize("key", sorted(FC))
def test_full_size_frames_against_sparse_fixture(P, oracle, key):
    """BASELINE's configs at full size: 20000 seeded pixels of the oracle's frame."""
    cfg = FC[key]
    g = np.load(os.path.join(GOLDEN, f"sparse_{key}.npz"))
    hs, gs, m, lights = _host_gpu(P, cfg["scene"])
    W, H = cfg["width"], cfg["height"]
    corners = P.host.camera_corners(m, W, H)
    assert np.allclose(corners, g["corners"], rtol=0, atol=2e-6)
    rgb, depth, tri, st = gs.render(g["corners"], g["lights"], P.capi.default_params(W, H, cfg["msaa"]))
    x, y = g["x"], g["y"]
    assert (tri[y, x] == g["tri"]).mean() >= 0.9999
    q = oracle.to_bytes(rgb[y, x])
    d = np.abs(q.astype(int) - g["rgb8"].astype(int)).max(axis=1)
    assert (d <= 1).mean() >= 0.999, f"{(d > 1).sum()} of {d.size} sampled pixels off by > 1 LSB"
    miss = g["depth"] == FLT_MAX
    assert ((depth[y, x] == FLT_MAX) == miss).mean() >= 0.9999
    both = ~miss & (depth[y, x] != FLT_MAX)
    assert (np.abs(depth[y, x][both] - g["depth"][both]) <= 1e-5 * g["depth"][both]).mean() >= 0.9995
    # size-independent properties of the whole frame
    assert abs((tri >= 0).mean() - cfg["hit_fraction"]) < 1e-4
    assert (rgb >= 0).all() and (rgb <= 1).all()                 # RGBValue clamp
    assert (rgb[tri < 0] == np.float32([0.2, 0.5, 1.0])).all()    # misses are background
    assert (depth[tri < 0] == FLT_MAX).all()
    # the GPU skips the discarded centre-ray subtrees and the shadow queries whose answer
    # cannot change shadowMax: fewer queries than the reference's count, never more; the
    # closest-hit queries (one per ray of the tree) are all traced
    ref_rays = cfg["frame"]["rays_closest"] + cfg["frame"]["rays_shadow"]
    gpu_rays = st.rays_closest + st.rays_shadow
    assert gpu_rays <= ref_rays
    if cfg["msaa"] == 1:
        # (a hit the reference's walk steps past, KDtree.cpp:288-587, changes a ray tree here and there)
        assert abs(st.rays_closest - cfg["frame"]["rays_closest"]) <= 1e-4 * cfg["frame"]["rays_closest"]
        # the shadow queries left out are the ones the oracle counts as unable to change shadowMax
        needed = cfg["frame"]["rays_shadow"] - cfg["frame"]["moot_shadow_rays"]
        assert abs(st.rays_shadow - needed) <= 1e-3 * needed and st.rays_shadow <= cfg["frame"]["rays_shadow"]


@pytest.mark.parametrize("key", sorted(FC))
def test_full_size_frames_against_the_reference(P, oracle, key):
    """BASELINE's configs at full size against the REAL reference: 20000 seeded pixels whose
    camera rays (centre + MSAA offsets) went through the reference's own performRayTracing
    (tools/make_sparse_ref.py, fixture sparse_ref_*.npz).  That build has no image library, so
    both sides render the scene untextured."""
    cfg = FC[key]
    g = np.load(os.path.join(GOLDEN, f"sparse_ref_{key}.npz"))
    hs = P.host.HostScene(os.path.join(SCENES, cfg["scene"] + ".obj"), textures=False)
    gs = hs.to_gpu(0)
    W, H = cfg["width"], cfg["height"]
    rgb, depth, tri, st = gs.render(g["corners"], g["lights"], P.capi.default_params(W, H, cfg["msaa"]))
    x, y = g["x"], g["y"]
    assert (tri[y, x] == g["tri"]).mean() >= 0.9999          # triangle of the pixel-centre ray
    d = np.abs(oracle.to_bytes(rgb[y, x]).astype(int) - g["rgb8"].astype(int)).max(axis=1)
    assert (d <= 1).mean() >= 0.999, f"{(d > 1).sum()} of {d.size} sampled pixels off by > 1 LSB"
    assert (d == 0).mean() >= 0.98
    miss = g["depth"] == FLT_MAX
    assert ((depth[y, x] == FLT_MAX) == miss).mean() >= 0.9999
    both = ~miss & (depth[y, x] != FLT_MAX)
    if cfg["msaa"] == 1:     # the camera ray's own t: bit-exact
        assert (depth[y, x][both] == g["depth"][both]).mean() >= 0.9999
    else:                    # mean of the samples' t, same order of additions
        assert (np.abs(depth[y, x][both] - g["depth"][both]) <= 1e-6 * g["depth"][both]).mean() >= 0.9995


@pytest.mark.parametrize("key", [k for k in sorted(FC) if FC[k]["scene"] in ("lego", "final", "fin")])
def test_full_size_textured_frames_against_the_reference(P, oracle, key):
    """The two textured BASELINE configs at full size, WITH their textures, against the REAL reference:
    20000 seeded pixels whose camera rays went through the reference's performRayTracing with
    Material::getTexture (mesh.cpp:485-553) reading the .rtex pixels + two zero rows
    (tools/make_tex_golden.py, fixture sparse_reftex_*.npz).  Scene through the product's own loader."""
    cfg = FC[key]
    g = np.load(os.path.join(GOLDEN, f"sparse_reftex_{key}.npz"))
    hs, gs, m, lights = _host_gpu(P, cfg["scene"])
    W, H = cfg["width"], cfg["height"]
    rgb, depth, tri, st = gs.render(g["corners"], g["lights"], P.capi.default_params(W, H, cfg["msaa"]))
    x, y = g["x"], g["y"]
    assert (tri[y, x] == g["tri"]).mean() >= 0.9999
    d = np.abs(oracle.to_bytes(rgb[y, x]).astype(int) - g["rgb8"].astype(int)).max(axis=1)
    assert (d <= 1).mean() >= 0.999, f"{(d > 1).sum()} of {d.size} sampled pixels off by > 1 LSB"
    assert (d == 0).mean() >= 0.98
    u = np.load(os.path.join(GOLDEN, f"sparse_ref_{key}.npz"))
    assert (u["rgb8"] != g["rgb8"]).any(axis=1).mean() > 0.01      # textured and untextured frames do differ


def test_full_size_post_process_of_the_headline_frame(P, oracle):
    """The headline frame (final.obj 3840 x 2160, MSAA 4) through the fused post-process on the device
    (rt_postprocess_device: depth normalised in place, second normalisation folded into the blur): the whole
    normalised depth plane against the oracle's normalisation, bit for bit, and 300 seeded pixels of the
    blurred frame against the oracle's 13 x 13 kernel on the frame's own colours, within 2e-6.  (The image
    kernels meet the real reference's vectors at 96 x 54; this is the same code at the size that is timed.)"""
    import ctypes as C

    import torch
    hs, gs, m, lights = _host_gpu(P, "final")
    W, H = 3840, 2160
    corners = P.host.camera_corners(m, W, H)
    p = P.capi.default_params(W, H, 4)
    rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    depth = torch.zeros((H, W), dtype=torch.float32, device="cuda")
    out = torch.zeros((H - 12, W - 12, 3), dtype=torch.float32, device="cuda")
    gs.render_device(corners, lights, p, rgb.data_ptr(), depth.data_ptr(), 0, 0, False)
    torch.cuda.synchronize()
    raw = depth.cpu().numpy()
    L = P.capi.lib()
    P.capi.check(L.rt_postprocess_device(0, W, H, p.focusDepth, p.depthPower, C.c_void_p(rgb.data_ptr()),
                                         C.c_void_p(depth.data_ptr()), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    d1 = depth.cpu().numpy()
    assert np.array_equal(d1, oracle.normalize_depth(raw, p.focusDepth))
    h_rgb, blurred = rgb.cpu().numpy(), out.cpu().numpy()
    d2 = oracle.normalize_depth(d1, p.focusDepth)
    rng = np.random.Generator(np.random.PCG64(2160))
    xs, ys = rng.integers(6, W - 6, 300), rng.integers(6, H - 6, 300)
    worst = 0.0
    for x, y in zip(xs, ys):
        k = oracle.gaussian_map(float(d2[y, x]), p.depthPower)
        acc = np.zeros(3, np.float32)
        win = h_rgb[y - 6:y + 7, x - 6:x + 7]
        for j in range(13):                      # the reference's order of the 169 additions (imageWriter.h:204-229)
            for i in range(13):
                acc = (acc + win[j, i] * k[j, i]).astype(np.float32)
        got = blurred[y - 6, x - 6]
        if np.isnan(acc).any():
            assert np.isnan(got).all() == np.isnan(acc).all()
            continue
        worst = max(worst, float(np.abs(acc - got).max()))
    assert worst < 2e-6, worst


def test_bad_arguments_are_rejected(P):
    hs, gs, m, lights = _host_gpu(P, "ball")
    L = P.capi.lib()
    corners = P.host.camera_corners(m, 64, 64)
    p = P.capi.default_params(64, 64)
    p.msaa = 3
    with pytest.raises(P.capi.RtError, match="msaa"):
        gs.render(corners, lights, p)
    p = P.capi.default_params(1, 64)
    with pytest.raises(P.capi.RtError, match="frame size"):
        gs.render(corners, lights, p)
    with pytest.raises(P.capi.RtError, match="n_lights"):
        gs.trace_rays(np.zeros((1, 3)), np.ones((1, 3)), np.zeros((65, 3), np.float32))
    with pytest.raises(P.capi.RtError, match="k "):
        gs.trace_rays(np.zeros((1, 3)), np.ones((1, 3)), lights, k=6)
    with pytest.raises(P.capi.RtError, match="13x13"):
        P.capi.dof_blur(np.zeros((10, 10, 3), np.float32), np.zeros((10, 10), np.float32))
    # a scene whose node list points outside the triangle array
    a = hs.arrays()
    bad = a["node_tris"].copy()
    bad[0] = 10 ** 6
    with pytest.raises(P.capi.RtError, match="lists triangle"):
        P.capi.Scene(a["vertex_pos"], a["vertex_normal"], a["tri_vertex"], a["tri_texcoord"],
                     a["tri_material"], a["texcoords"], a["materials"], a["textures"],
                     a["node_bounds"], a["node_left"], a["node_right"], a["node_tri_off"],
                     a["node_tri_cnt"], bad)
    # zero rays and zero lights are fine
    rgb, mt, tri, _ = gs.trace_rays(np.zeros((0, 3)), np.zeros((0, 3)), lights)
    assert rgb.shape == (0, 3)
    rgb, _, _, _ = gs.trace_rays([[0, 1.6, 3.6]], [[0, 0.45, 0]], np.zeros((0, 3), np.float32))
    assert np.isfinite(rgb).all()


def test_scene_without_a_tree_renders_background(P):
    """A description with no KD nodes (only the raw C ABI can make one: the host builder always
    emits a root) is given one empty leaf: every ray misses, frames are background, nothing hangs."""
    hs, _, m, lights = _host_gpu(P, "ball")
    a = hs.arrays()
    none_i, none_f = np.zeros(0, np.int32), np.zeros((0, 6), np.float32)
    for tris in (True, False):
        gs = P.capi.Scene(a["vertex_pos"], a["vertex_normal"],
                          a["tri_vertex"] if tris else np.zeros((0, 3), np.int32),
                          a["tri_texcoord"] if tris else np.zeros((0, 3), np.int32),
                          a["tri_material"] if tris else none_i, a["texcoords"], a["materials"],
                          a["textures"], none_f, none_i, none_i, none_i, none_i, none_i)
        assert gs.info()["n_nodes"] == 1
        for msaa in (1, 4):
            rgb, depth, tri, st = gs.render(P.host.camera_corners(m, 64, 48), lights, P.capi.default_params(64, 48, msaa))
            assert (tri == -1).all() and (depth == FLT_MAX).all()
            assert (rgb == np.float32([0.2, 0.5, 1.0])).all() and st.rays_shadow == 0
        t, _, _, _, _ = gs.check_hits([[0, 1.6, 3.6]], [[0, 0.45, 0]])
        assert t[0] == -1
        c, mt, _, _ = gs.trace_rays([[0, 1.6, 3.6]], [[0, 0.45, 0]], lights)
        assert np.array_equal(c[0], np.float32([0.2, 0.5, 1.0])) and mt[0] == FLT_MAX


def test_trace_rays_k_and_prev_ni(P, oracle):
    """performRayTracing's k and prev_Ni arguments (raytracing.cpp:61,166-175)."""
    s = oracle.Scene.load(os.path.join(SCENES, "refractionballs.obj"))
    g = np.load(os.path.join(GOLDEN, "refractionballs.npz"))
    s.set_lights(g["lights"])
    from conftest import gpu_scene_from_oracle
    gs = gpu_scene_from_oracle(s)
    o, d = g["origins"][:3000], g["dests"][:3000]
    full, _, _, _ = gs.trace_rays(o, d, g["lights"], k=0)
    last, _, _, st = gs.trace_rays(o, d, g["lights"], k=4)
    assert st.rays_closest == len(o)                     # k = 4 spawns no children
    assert not np.array_equal(full, last)
    L = oracle.lib()
    ref = np.zeros((len(o), 3), np.float32)
    mt = C.c_float()
    for i in range(0, len(o), 37):
        L.or_perform_ray_tracing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
        L.or_perform_ray_tracing(s.h, o[i].ctypes.data, d[i].ctypes.data, 4, 1.0, C.byref(mt),
                                 ref[i].ctypes.data, None)
        assert np.abs(ref[i] - last[i]).max() < 1e-5


def test_render_paths_agree(P, oracle, monkeypatch):
    """The three frame renderers the library holds: the single persistent launch (default),
    one launch per stage (RT_WF_MULTILAUNCH=1) and the per-sample kernels
    (RT_MEGAKERNEL=1).  The first two accumulate in order-free fixed point and must agree
    bit for bit; the third sums in float."""
    hs, gs, m, lights = _host_gpu(P, "refractionballs")
    W, H = 192, 108
    corners = P.host.camera_corners(m, W, H)
    frames = {}
    for name, env in [("pipeline", {}), ("multilaunch", {"RT_WF_MULTILAUNCH": "1"}),
                      ("megakernel", {"RT_MEGAKERNEL": "1"}), ("region512", {"RT_WF_REGION": "512"})]:
        for k in ("RT_WF_MULTILAUNCH", "RT_MEGAKERNEL", "RT_WF_REGION"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gs = hs.to_gpu(0)                                # the switches are read when a scene is created
        frames[name] = gs.render(corners, lights, P.capi.default_params(W, H, 4))
    for k in ("RT_WF_MULTILAUNCH", "RT_MEGAKERNEL", "RT_WF_REGION"):
        monkeypatch.delenv(k, raising=False)
    a = frames["pipeline"]
    for other in ("multilaunch", "region512"):
        b = frames[other]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert a[3].rays_closest == b[3].rays_closest and a[3].rays_shadow == b[3].rays_shadow
    c = frames["megakernel"]
    assert np.array_equal(a[2], c[2]) and np.array_equal(a[1], c[1])
    d = np.abs(oracle.to_bytes(a[0]).astype(int) - oracle.to_bytes(c[0]).astype(int))
    assert d.max() <= 1
    assert a[3].rays_closest == c[3].rays_closest


def test_pending_counts_in_lds_and_in_the_records_agree(P, monkeypatch):
    """Where a region's records are few (no refraction, at most four lights, hard shadows) the single launch keeps
    the count of a hit's pending shadow queries and the lights found occluded in a byte of LDS; otherwise, and in
    the launch-per-stage path, in a word of the hit's record.  Same frame bit for bit either way - final.obj
    with its three lights and redballs with its four take the LDS path by default."""
    for scene in ("final", "redballs"):
        hs, gs, m, lights = _host_gpu(P, scene)
        assert len(lights) <= 4
        W, H = 224, 128
        corners = P.host.camera_corners(m, W, H)
        frames = {}
        for name, env in [("lds", {}), ("records", {"RT_WF_NO_LDS_PENDING": "1"}), ("multilaunch", {"RT_WF_MULTILAUNCH": "1"})]:
            for k in ("RT_WF_MULTILAUNCH", "RT_WF_NO_LDS_PENDING"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            g2 = hs.to_gpu(0)                            # the launch structure is read when a scene is created
            frames[name] = [g2.render(corners, lights, P.capi.default_params(W, H, msaa)) for msaa in (1, 4)]
        for k in ("RT_WF_MULTILAUNCH", "RT_WF_NO_LDS_PENDING"):
            monkeypatch.delenv(k, raising=False)
        for other in ("records", "multilaunch"):
            for a, b in zip(frames["lds"], frames[other]):
                assert a[3].rays_shadow > 0 and a[3].rays_shadow == b[3].rays_shadow
                assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), (scene, other)
                assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])


def test_where_stacks_and_nodes_live_changes_nothing(P, monkeypatch):
    """The walk keeps the first entries of a thread's stack and a copy of the tree's first levels in LDS; the rest
    of a stack lies in HBM, the rest of the tree is read through the vector L1.  How much of either is in LDS is
    a matter of time only: two stack entries (everything deeper spilled), eight, and no copy of the tree at all
    give the default's frame bit for bit."""
    hs, gs, m, lights = _host_gpu(P, "final")
    W, H = 256, 144
    corners = P.host.camera_corners(m, W, H)
    keys = ("RT_WF_STACK_LDS", "RT_WF_TOP")
    frames = []
    for env in ({}, {"RT_WF_STACK_LDS": "2"}, {"RT_WF_STACK_LDS": "8", "RT_WF_TOP": "0"}, {"RT_WF_TOP": "512"}):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g2 = hs.to_gpu(0)                                # both are read when a scene is set up for its first frame
        frames.append(g2.render(corners, lights, P.capi.default_params(W, H, 4)))
    for k in keys:
        monkeypatch.delenv(k, raising=False)
    a = frames[0]
    assert a[3].rays_closest > 0 and a[3].rays_shadow > 0
    for b in frames[1:]:
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])
        assert a[3].rays_closest == b[3].rays_closest and a[3].rays_shadow == b[3].rays_shadow


def test_region_order_from_the_last_frame_changes_nothing(P, monkeypatch):
    """From the second frame of a shape on, the single launch hands its regions out in the order of the
    durations the frame before measured (k_wf_order_*).  That is a hint about when a region runs, never about
    what it renders: frames 2 and 3 of a scene equal its first frame and the frame of a scene that never
    reorders (RT_WF_FEEDBACK=0), bit for bit, also across a change of shape and for one rank's share of a deal."""
    monkeypatch.delenv("RT_WF_FEEDBACK", raising=False)
    hs, gs, m, lights = _host_gpu(P, "refractionballs")
    monkeypatch.setenv("RT_WF_FEEDBACK", "0")
    plain = hs.to_gpu(0)
    monkeypatch.delenv("RT_WF_FEEDBACK", raising=False)
    for W, H, msaa, world in ((800, 600, 1, 1), (512, 384, 4, 1), (800, 600, 1, 1), (800, 608, 1, 2)):
        corners = P.host.camera_corners(m, W, H)
        p = P.capi.default_params(W, H, msaa)
        if world > 1:
            p.tile_w = p.tile_h = 32
            p.world, p.rank = world, 1
        want = plain.render(corners, lights, p)
        assert want[3].rays_closest > 0
        for it in range(3):
            got = gs.render(corners, lights, p)
            assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), (W, H, msaa, it)
            assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
            assert np.array_equal(got[2], want[2])
            assert got[3].rays_closest == want[3].rays_closest and got[3].rays_shadow == want[3].rays_shadow
        # the order is a permutation of the regions: a frame rendered into planes full of NaN leaves none
        # behind in the pixels this rank owns and equals the plain frame there
        import torch
        rgb = torch.full((H, W, 3), float("nan"), dtype=torch.float32, device="cuda")
        depth = torch.full((H, W), float("nan"), dtype=torch.float32, device="cuda")
        gs.render_device(corners, lights, p, rgb.data_ptr(), depth.data_ptr(), 0, 0, False)
        torch.cuda.synchronize()
        plain_rgb = torch.full((H, W, 3), float("nan"), dtype=torch.float32, device="cuda")
        plain_depth = torch.full((H, W), float("nan"), dtype=torch.float32, device="cuda")
        plain.render_device(corners, lights, p, plain_rgb.data_ptr(), plain_depth.data_ptr(), 0, 0, False)
        torch.cuda.synchronize()
        assert torch.equal(rgb.view(torch.int32), plain_rgb.view(torch.int32))
        assert torch.equal(depth.view(torch.int32), plain_depth.view(torch.int32))
        if world == 1:
            assert not bool(torch.isnan(depth).any())


def test_chunked_frame_and_many_lights(P, monkeypatch):
    """A frame too large for its workspace is rendered in chunks of regions (here forced with
    a 2 MB budget), in both launch structures; 40 lights stretch the per-light queues.  All
    must equal the one-chunk frame bit for bit."""
    hs, gs, m, lights = _host_gpu(P, "final")
    rng = np.random.default_rng(3)
    many = np.concatenate([lights, rng.uniform(-6, 6, (37, 3)).astype(np.float32)])
    W, H = 160, 96
    corners = P.host.camera_corners(m, W, H)
    for k in ("RT_WF_MULTILAUNCH", "RT_WF_BUDGET_MB"):
        monkeypatch.delenv(k, raising=False)
    whole = gs.render(corners, many, P.capi.default_params(W, H, 4))
    assert whole[3].rays_shadow > 0
    for env in ({"RT_WF_BUDGET_MB": "2"}, {"RT_WF_BUDGET_MB": "2", "RT_WF_MULTILAUNCH": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        hs2, gs2, _, _ = _host_gpu(P, "final")           # a scene of its own: the workspace is sized per scene
        part = gs2.render(corners, many, P.capi.default_params(W, H, 4))
        for k in env:
            monkeypatch.delenv(k, raising=False)
        assert np.array_equal(part[0], whole[0]) and np.array_equal(part[1], whole[1])
        assert np.array_equal(part[2], whole[2])
        assert part[3].rays_closest == whole[3].rays_closest and part[3].rays_shadow == whole[3].rays_shadow


def test_bench_two_ranks_rehearsal():
    """bench.py as two ranks sharing this GPU, payloads gathered through host memory (gloo):
    the tile partition, pack, gather and unpack path of the N > 1 bench, end to end; the
    gathered frame must equal the frame one GPU renders alone (bench.py checks it itself)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_CHECK="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "gathered frame equals the single-GPU frame" in r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["roofline"]["kernel"] == "k_wf_pipeline" and "cpu_baseline" not in out


def test_rt_render_program(P, tmp_path):
    """The headless 'r' key as a program of its own (C++ host + librt_hip.so, no Python, no
    torch in the process): same four files as the library path, byte for byte."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "raytracing-project_amd", "csrc", "rt_render")
    assert os.path.exists(exe), "rt_render is built by raytracing-project_amd/build.py"
    obj, sav = os.path.join(SCENES, "lego.obj"), os.path.join(SCENES, "cameras", "lego.sav")
    a, b = tmp_path / "prog", tmp_path / "lib"
    a.mkdir(), b.mkdir()
    env = dict(os.environ)
    libdir = os.path.dirname(exe)
    env["LD_LIBRARY_PATH"] = libdir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, obj, sav, "192", "108", "4", str(a), "0", "100", "60"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "closest" in r.stdout and "Focus depth set to" in r.stderr
    P.host.render_to_files_focus(obj, sav, 192, 108, 4, str(b), 100.0, 60.0)
    for f in ("result.ppm", "result.bmp", "depth.bmp", "blurred.bmp"):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f
    bad = subprocess.run([exe, obj], env=env, capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "usage" in bad.stderr


def test_png_textures_through_the_product_path(P, tmp_path):
    """The scene's textures as PNG files next to the .obj (this repository carries them
    pre-decoded, as .rtex): the loader must pick the PNG, decode it (png_decoder.cpp) and render
    the frame the pre-decoded copy gives.  The PNGs are written here from the .rtex pixels."""
    import shutil
    import struct
    import zlib

    from test_host import _png_chunk
    for name, W, H in (("final", 96, 54), ("lego", 80, 60)):
        src = P.host.HostScene(os.path.join(SCENES, name + ".obj"))
        m, lights = P.host.read_camera_sav(os.path.join(SCENES, "cameras", name + ".sav"))
        d = tmp_path / name
        (d / "textures").mkdir(parents=True)
        for f in os.listdir(SCENES):
            if f.endswith((".obj", ".mtl")) and f.startswith(name):
                shutil.copy(os.path.join(SCENES, f), d / f)
        texs = src.arrays()["textures"]
        assert texs
        rtex = sorted(f for f in os.listdir(os.path.join(SCENES, "textures")) if f.endswith(".rtex"))
        written = 0
        for f in rtex:           # every shipped texture as a PNG; the loader opens the ones the .mtl names
            raw = open(os.path.join(SCENES, "textures", f), "rb").read()
            w, h, zl = struct.unpack("<III", raw[4:16])
            px = np.frombuffer(zlib.decompress(raw[16:16 + zl]), np.uint8).reshape(h, w, 4)
            lines = b"".join(b"\0" + px[y].tobytes() for y in range(h))
            png = (b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                   + _png_chunk(b"IDAT", zlib.compress(lines)) + _png_chunk(b"IEND", b""))
            (d / "textures" / f[:-5]).write_bytes(png)
            written += 1
        assert written == len(rtex)
        via_png = P.host.HostScene(str(d / (name + ".obj")))
        a, b = src.arrays(), via_png.arrays()
        assert len(a["textures"]) == len(b["textures"]) > 0
        for (pa, wa, ha), (pb, wb, hb) in zip(a["textures"], b["textures"]):
            assert (wa, ha) == (wb, hb) and np.array_equal(pa, pb)
        corners = P.host.camera_corners(m, W, H)
        f1 = src.to_gpu(0).render(corners, lights, P.capi.default_params(W, H, 4))
        f2 = via_png.to_gpu(0).render(corners, lights, P.capi.default_params(W, H, 4))
        assert np.array_equal(f1[0], f2[0]) and np.array_equal(f1[1], f2[1])


def test_rccl_layer_with_one_rank(P, tmp_path):
    """What one GPU can exercise of the multi-GPU path in C: RCCL is found and bound, a
    communicator of one rank is made, the gather of one rank is the copy of its payload, and
    `rt_render --gpus 1` (the forked process per GPU) writes the files of the plain run."""
    import subprocess

    import torch
    L = P.capi.lib()
    ident = (C.c_ubyte * 128)()
    P.capi.check(L.rt_rccl_unique_id(ident))
    assert any(ident)
    comm = C.c_void_p()
    P.capi.check(L.rt_rccl_comm_create(ident, 1, 0, 0, C.byref(comm)))
    pay = torch.arange(4096, dtype=torch.float32, device="cuda")
    out = torch.zeros_like(pay)
    n = (C.c_int64 * 1)(pay.numel())
    P.capi.check(L.rt_gather_tiles_rccl(comm, 0, 1, C.c_void_p(pay.data_ptr()), n, C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.equal(pay, out)
    L.rt_rccl_comm_destroy(comm)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "raytracing-project_amd", "csrc", "rt_render")
    obj, sav = os.path.join(SCENES, "redballs.obj"), os.path.join(SCENES, "cameras", "redballs.sav")
    a, b = tmp_path / "forked", tmp_path / "plain"
    a.mkdir(), b.mkdir()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.dirname(exe) + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    for args, d in ((["--gpus", "1"], a), ([], b)):
        r = subprocess.run([exe, *args, obj, sav, "200", "120", "4", str(d)], env=env, capture_output=True, text=True,
                           timeout=120)
        assert r.returncode == 0, r.stderr[-1500:]
        assert "rank 0 of 1" in r.stdout
    for f in ("result.ppm", "result.bmp", "depth.bmp", "blurred.bmp"):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f


@pytest.mark.parametrize("W,H", [(13, 13), (14, 13), (45, 21), (77, 20), (44, 33)])
def test_dof_blur_ragged_sizes(P, oracle, W, H):
    """Frames from the smallest the blur accepts (one output pixel) through sizes that leave
    partial 32x8 blocks on both edges."""
    rng = np.random.default_rng(W * 100 + H)
    rgb = rng.random((H, W, 3), dtype=np.float32)
    dn = rng.random((H, W), dtype=np.float32) * 0.9
    dn[H // 2, W // 2] = 0.0
    got, ref = P.capi.dof_blur(rgb, dn), oracle.blur(rgb, dn, threads=2)
    assert got.shape == ref.shape == (H - 12, W - 12, 3)
    assert np.abs(got - ref).max() < 2e-6
    q, oq = P.capi.quantize(got), oracle.to_bytes(ref)
    assert (np.abs(q.astype(int) - oq.astype(int)) <= 1).all()
