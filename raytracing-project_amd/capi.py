"""ctypes view of the C ABI declared in include/rt_hip.h (librt_hip.so).

Plumbing for tests and bench.py only: the product is the shared library; a
C++ host links to it directly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIB selects another build of the library (kernel tuning experiments)
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "csrc", "librt_hip.so")

RT_MAX_LIGHTS = 64


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


class RtMaterial(C.Structure):
    _fields_ = [("Kd", C.c_float * 3), ("Ks", C.c_float * 3), ("Ns", C.c_float), ("Ni", C.c_float),
                ("illum", C.c_int32), ("has_Ni", C.c_int32), ("texture", C.c_int32),
                ("reserved", C.c_int32 * 5)]


class RtTexture(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class RtKdNode(C.Structure):
    _fields_ = [("bounds", C.c_float * 6), ("left", C.c_int32), ("right", C.c_int32),
                ("tri_offset", C.c_int32), ("tri_count", C.c_int32)]


class RtSceneDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("vertex_pos", C.c_void_p), ("vertex_normal", C.c_void_p),
                ("n_triangles", C.c_int32), ("tri_vertex", C.c_void_p), ("tri_texcoord", C.c_void_p),
                ("tri_material", C.c_void_p),
                ("n_texcoords", C.c_int32), ("texcoords", C.c_void_p),
                ("n_materials", C.c_int32), ("materials", C.c_void_p),
                ("n_textures", C.c_int32), ("textures", C.c_void_p),
                ("n_nodes", C.c_int32), ("nodes", C.c_void_p),
                ("n_node_tris", C.c_int32), ("node_tris", C.c_void_p)]


class RtParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("msaa", C.c_int32),
                ("backgroundlighting", C.c_float), ("diffusePower", C.c_float),
                ("ShadowFactor", C.c_float), ("depthPower", C.c_float),
                ("backgroundColor", C.c_float * 3), ("focusDepth", C.c_float),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("rank", C.c_int32),
                ("world", C.c_int32), ("rank_span", C.c_int32), ("shadowSamples", C.c_int32),
                ("shadowRadius", C.c_float)]


class RtStats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("tri_tests", C.c_uint64),
                ("node_visits", C.c_uint64), ("camera_rays", C.c_uint64), ("kernel_ms", C.c_float),
                ("reserved", C.c_float), ("shaded_hits", C.c_uint64), ("child_rays", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


class RtKernelTimes(C.Structure):
    _fields_ = [("trace_closest_ms", C.c_double), ("trace_any_ms", C.c_double), ("frame_ms", C.c_double),
                ("trace_closest_launches", C.c_int32), ("trace_any_launches", C.c_int32),
                ("frames", C.c_int32), ("reserved", C.c_int32)]


class RtSceneInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_triangles", C.c_int32), ("tree_depth", C.c_int32),
                ("n_slots", C.c_int32), ("device_bytes", C.c_uint64), ("tree_hash", C.c_uint64),
                ("create_ms", C.c_float), ("tree_ms", C.c_float), ("tree_on_gpu", C.c_int32),
                ("render_path", C.c_int32)]


EXPORTS = [
    "rt_default_params", "rt_device_count", "rt_last_error", "rt_scene_create", "rt_scene_destroy",
    "rt_scene_info", "rt_scene_set_timing", "rt_scene_collect_timing", "rt_check_hits", "rt_trace_rays", "rt_render_frame", "rt_render_frame_device", "rt_render_result_planes",
    "rt_normalize_depth", "rt_normalize_depth_device", "rt_normalize_depth_device_noalloc", "rt_dof_blur", "rt_dof_blur_device",
    "rt_postprocess_device", "rt_quantize", "rt_tile_pixels", "rt_pack_tiles_device", "rt_unpack_tiles_device",
    "rt_rccl_unique_id", "rt_rccl_comm_create", "rt_rccl_comm_destroy", "rt_gather_tiles_rccl",
    "rt_render_result_planes_shared", "rt_render_result_planes_rehearsal", "rt_deal_params", "rt_scene_set_deal", "rt_scene_set_region_order",
    "rt_calibrate_deal", "rt_build_kd_tree",
]

_lib = None


def lib():
    """Loads librt_hip.so; there is no fallback - a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'`")
    # torch ships its own HIP runtime under the same soname as /opt/rocm's.  Two copies
    # in one process do not both see the GPU, so when torch is going to be used in this
    # process (bench.py, tests that allocate device memory with it) it has to be loaded
    # first; librt_hip.so then binds to the runtime that is already there.
    if os.environ.get("RT_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    P, I32, I64, F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.rt_default_params.argtypes = [C.POINTER(RtParams), I32, I32]
    L.rt_default_params.restype = None
    L.rt_device_count.restype = C.c_int
    L.rt_last_error.restype = C.c_char_p
    L.rt_scene_create.argtypes = [C.POINTER(RtSceneDesc), C.c_int, C.POINTER(P)]
    L.rt_scene_destroy.argtypes = [P]
    L.rt_scene_destroy.restype = None
    L.rt_scene_info.argtypes = [P, C.POINTER(RtSceneInfo)]
    L.rt_scene_set_timing.argtypes = [P, C.c_int]
    L.rt_scene_collect_timing.argtypes = [P, C.POINTER(RtKernelTimes)]
    L.rt_check_hits.argtypes = [P, I64, P, P, P, P, P, P, P]
    L.rt_trace_rays.argtypes = [P, P, I32, C.POINTER(RtParams), I32, F, I64, P, P, P, P, P,
                                C.POINTER(RtStats)]
    L.rt_render_frame.argtypes = [P, P, P, I32, C.POINTER(RtParams), P, P, P, C.POINTER(RtStats)]
    L.rt_render_frame_device.argtypes = [P, P, P, I32, C.POINTER(RtParams), P, P, P, P,
                                         C.POINTER(RtStats)]
    L.rt_render_result_planes.argtypes = [P, P, P, I32, C.POINTER(RtParams), C.POINTER(P), C.POINTER(P),
                                          C.POINTER(P), C.POINTER(RtStats)]
    L.rt_normalize_depth.argtypes = [C.c_int, I64, F, P]
    L.rt_normalize_depth_device.argtypes = [C.c_int, I64, F, P, P]
    L.rt_normalize_depth_device_noalloc.argtypes = [C.c_int, I64, F, P, P]
    L.rt_dof_blur.argtypes = [C.c_int, I32, I32, F, P, P, P]
    L.rt_dof_blur_device.argtypes = [C.c_int, I32, I32, F, P, P, P, P]
    L.rt_postprocess_device.argtypes = [C.c_int, I32, I32, F, F, P, P, P, P]
    L.rt_quantize.argtypes = [C.c_int, I64, P, P]
    L.rt_tile_pixels.argtypes = [C.POINTER(RtParams), I32]
    L.rt_tile_pixels.restype = I64
    L.rt_pack_tiles_device.argtypes = [C.c_int, C.POINTER(RtParams), P, P, P, P]
    L.rt_unpack_tiles_device.argtypes = [C.c_int, C.POINTER(RtParams), I32, P, P, P, P]
    L.rt_render_result_planes_shared.argtypes = [P, P, I32, I32, P, P, I32, C.POINTER(RtParams), C.POINTER(P),
                                                 C.POINTER(P), C.POINTER(P), C.POINTER(RtStats)]
    L.rt_render_result_planes_rehearsal.argtypes = [P, I32, P, P, I32, C.POINTER(RtParams), C.POINTER(P),
                                                    C.POINTER(P), C.POINTER(P), C.POINTER(RtStats)]
    L.rt_deal_params.argtypes = [C.POINTER(RtParams), I32, I32, I32, C.POINTER(RtParams)]
    L.rt_scene_set_deal.argtypes = [P, I32]
    L.rt_scene_set_region_order.argtypes = [P, I32]
    L.rt_calibrate_deal.argtypes = [P, P, I32, I32, P, P, I32, C.POINTER(RtParams), C.POINTER(I32)]
    L.rt_build_kd_tree.argtypes = [C.c_int, I32, P, I32, P, I32, P, C.POINTER(I32), P, C.POINTER(I32)]
    L.rt_rccl_unique_id.argtypes = [P]
    L.rt_rccl_comm_create.argtypes = [P, I32, I32, C.c_int, C.POINTER(P)]
    L.rt_rccl_comm_destroy.argtypes = [P]
    L.rt_rccl_comm_destroy.restype = None
    L.rt_gather_tiles_rccl.argtypes = [P, I32, I32, P, C.POINTER(I64), P, P]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RtError(rc, lib().rt_last_error().decode(errors="replace"))


def default_params(width, height, msaa=1):
    p = RtParams()
    lib().rt_default_params(C.byref(p), width, height)
    p.msaa = msaa
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


class Scene:
    """RtScene handle built from flat arrays (the RtSceneDesc of rt_hip.h)."""

    def __init__(self, vertex_pos, vertex_normal, tri_vertex, tri_texcoord, tri_material, texcoords,
                 materials, textures, node_bounds, node_left, node_right, node_tri_off,
                 node_tri_cnt, node_tris, device=0):
        L = lib()
        self._keep = []
        vp, vn = _f32(vertex_pos, (-1, 3)), _f32(vertex_normal, (-1, 3))
        tv = np.ascontiguousarray(tri_vertex, np.int32).reshape(-1, 3)
        tt = np.ascontiguousarray(tri_texcoord, np.int32).reshape(-1, 3)
        tm = np.ascontiguousarray(tri_material, np.int32).reshape(-1)
        tc = _f32(texcoords, (-1, 3))
        mats = _f32(materials, (-1, 16))
        m_arr = (RtMaterial * max(len(mats), 1))()
        for i, r in enumerate(mats):
            m = m_arr[i]
            m.Kd[:] = r[0:3].tolist()
            m.Ks[:] = r[3:6].tolist()
            m.Ns, m.Ni = float(r[6]), float(r[7])
            m.illum, m.has_Ni, m.texture = int(r[8]), int(r[9] != 0), int(r[10])
        t_arr = (RtTexture * max(len(textures), 1))()
        for i, (px, w, h) in enumerate(textures):
            px = np.ascontiguousarray(px, np.uint8)
            self._keep.append(px)
            t_arr[i].rgba = px.ctypes.data
            t_arr[i].width, t_arr[i].height = w, h
        nb = _f32(node_bounds, (-1, 6))
        n_arr = (RtKdNode * max(len(nb), 1))()
        for i in range(len(nb)):
            n = n_arr[i]
            n.bounds[:] = nb[i].tolist()
            n.left, n.right = int(node_left[i]), int(node_right[i])
            n.tri_offset, n.tri_count = int(node_tri_off[i]), int(node_tri_cnt[i])
        nt = np.ascontiguousarray(node_tris, np.int32)
        d = RtSceneDesc()
        d.n_vertices, d.vertex_pos, d.vertex_normal = len(vp), _ptr(vp), _ptr(vn)
        d.n_triangles, d.tri_vertex, d.tri_texcoord, d.tri_material = len(tv), _ptr(tv), _ptr(tt), _ptr(tm)
        d.n_texcoords, d.texcoords = len(tc), _ptr(tc)
        d.n_materials, d.materials = len(mats), C.cast(m_arr, C.c_void_p)
        d.n_textures, d.textures = len(textures), C.cast(t_arr, C.c_void_p)
        d.n_nodes, d.nodes = len(nb), C.cast(n_arr, C.c_void_p)
        d.n_node_tris, d.node_tris = len(nt), _ptr(nt)
        self._keep += [vp, vn, tv, tt, tm, tc, m_arr, t_arr, n_arr, nt]
        self.desc = d
        self.device = device
        h = C.c_void_p()
        check(L.rt_scene_create(C.byref(d), device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().rt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = RtSceneInfo()
        check(lib().rt_scene_info(self.h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in i._fields_}

    def set_timing(self, enable):
        check(lib().rt_scene_set_timing(self.h, 1 if enable else 0))

    def set_region_order(self, from_last_frame):
        check(lib().rt_scene_set_region_order(self.h, 1 if from_last_frame else 0))

    def collect_timing(self):
        t = RtKernelTimes()
        check(lib().rt_scene_collect_timing(self.h, C.byref(t)))
        return t

    def check_hits(self, origins, dests):
        o, d = _f32(origins, (-1, 3)), _f32(dests, (-1, 3))
        n = len(o)
        tri = np.zeros(n, np.int32)
        a, b, t = (np.zeros(n, np.float32) for _ in range(3))
        hp = np.zeros((n, 3), np.float32)
        check(lib().rt_check_hits(self.h, n, _ptr(o), _ptr(d), _ptr(tri), _ptr(a), _ptr(b), _ptr(t),
                                  _ptr(hp)))
        return tri, a, b, t, hp

    def trace_rays(self, origins, dests, lights, params=None, k=0, prev_ni=1.0):
        o, d = _f32(origins, (-1, 3)), _f32(dests, (-1, 3))
        li = _f32(lights, (-1, 3))
        n = len(o)
        p = params or default_params(64, 64)
        rgb = np.zeros((n, 3), np.float32)
        mt = np.zeros(n, np.float32)
        tri = np.zeros(n, np.int32)
        st = RtStats()
        check(lib().rt_trace_rays(self.h, _ptr(li), len(li), C.byref(p), k, prev_ni, n, _ptr(o),
                                  _ptr(d), _ptr(rgb), _ptr(mt), _ptr(tri), C.byref(st)))
        return rgb, mt, tri, st

    def render(self, corners, lights, params):
        cam = _f32(corners, (24,))
        li = _f32(lights, (-1, 3))
        W, H = params.width, params.height
        rgb = np.zeros((H, W, 3), np.float32)
        depth = np.zeros((H, W), np.float32)
        tri = np.zeros((H, W), np.int32)
        st = RtStats()
        check(lib().rt_render_frame(self.h, _ptr(cam), _ptr(li), len(li), C.byref(params), _ptr(rgb),
                                    _ptr(depth), _ptr(tri), C.byref(st)))
        return rgb, depth, tri, st

    def result_planes(self, corners, lights, params):
        """rt_render_result_planes: the 8-bit planes of result.*, depth.bmp and blurred.bmp (copies;
        the blurred plane is None for frames too small to blur)."""
        cam = _f32(corners, (24,))
        li = _f32(lights, (-1, 3))
        W, H = params.width, params.height
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        st = RtStats()
        check(lib().rt_render_result_planes(self.h, _ptr(cam), _ptr(li), len(li), C.byref(params), C.byref(a),
                                            C.byref(b), C.byref(c), C.byref(st)))
        rgb8 = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_uint8)), (H, W, 3)).copy()
        depth8 = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint8)), (H, W)).copy()
        blur8 = None
        if c.value:
            blur8 = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_uint8)), (H - 12, W - 12, 3)).copy()
        return rgb8, depth8, blur8, st

    def result_planes_rehearsal(self, n_ranks, first_rank_shares, corners, lights, params):
        """rt_render_result_planes_rehearsal after rt_scene_set_deal: the shared frame of n_ranks GPUs on this one."""
        check(lib().rt_scene_set_deal(self.h, first_rank_shares))
        cam = _f32(corners, (24,))
        li = _f32(lights, (-1, 3))
        W, H = params.width, params.height
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        st = RtStats()
        check(lib().rt_render_result_planes_rehearsal(self.h, n_ranks, _ptr(cam), _ptr(li), len(li), C.byref(params),
                                                      C.byref(a), C.byref(b), C.byref(c), C.byref(st)))
        rgb8 = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_uint8)), (H, W, 3)).copy()
        depth8 = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint8)), (H, W)).copy()
        blur8 = None
        if c.value:
            blur8 = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_uint8)), (H - 12, W - 12, 3)).copy()
        return rgb8, depth8, blur8, st

    def render_device(self, corners, lights, params, d_rgb, d_depth, d_tri, stream=0, stats=False):
        """Device-pointer form (integers from torch tensors' data_ptr())."""
        cam = _f32(corners, (24,))
        li = _f32(lights, (-1, 3))
        st = RtStats() if stats else None
        check(lib().rt_render_frame_device(self.h, _ptr(cam), _ptr(li), len(li), C.byref(params),
                                           C.c_void_p(d_rgb), C.c_void_p(d_depth),
                                           C.c_void_p(d_tri) if d_tri else None,
                                           C.c_void_p(stream), C.byref(st) if stats else None))
        return st


def build_kd_tree(vertex_pos, tri_vertex, device=0):
    """rt_build_kd_tree: the reference's KD tree for these triangles, built on the GPU.  Returns a dict of
    arrays in node order: bounds (n, 6), left, right, tri_off, tri_cnt, tris, and depth."""
    vp = _f32(vertex_pos).reshape(-1, 3)
    tv = np.ascontiguousarray(tri_vertex, np.int32).reshape(-1, 3)
    nt = len(tv)
    cap = 4 * nt + 64
    nodes = (RtKdNode * cap)()
    n_nodes, depth = C.c_int32(0), C.c_int32(0)
    tris = np.zeros(max(nt, 1), np.int32)
    check(lib().rt_build_kd_tree(device, len(vp), _ptr(vp), nt, tv.ctypes.data_as(C.c_void_p), cap,
                                 C.cast(nodes, C.c_void_p), C.byref(n_nodes), tris.ctypes.data_as(C.c_void_p),
                                 C.byref(depth)))
    n = n_nodes.value
    raw = np.frombuffer(nodes, dtype=np.dtype([("bounds", np.float32, 6), ("left", np.int32), ("right", np.int32),
                                               ("tri_off", np.int32), ("tri_cnt", np.int32)]), count=n)
    return {"bounds": raw["bounds"].copy(), "left": raw["left"].copy(), "right": raw["right"].copy(),
            "tri_off": raw["tri_off"].copy(), "tri_cnt": raw["tri_cnt"].copy(), "tris": tris[:nt], "depth": depth.value}


def normalize_depth(depth, focus=0.0, device=0):
    d = _f32(depth).copy()
    check(lib().rt_normalize_depth(device, d.size, focus, _ptr(d)))
    return d


def dof_blur(rgb, depth_n, depth_power=2.5, device=0):
    rgb = _f32(rgb)
    H, W = rgb.shape[:2]
    dn = _f32(depth_n, (H, W))
    out = np.zeros((max(H - 12, 0), max(W - 12, 0), 3), np.float32)
    check(lib().rt_dof_blur(device, W, H, depth_power, _ptr(rgb), _ptr(dn), _ptr(out)))
    return out


def quantize(v, device=0):
    v = _f32(v)
    out = np.zeros(v.shape, np.uint8)
    check(lib().rt_quantize(device, v.size, _ptr(v), _ptr(out)))
    return out
