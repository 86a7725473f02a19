"""ctypes binding of the CPU oracle (oracle/rt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


class Settings(C.Structure):
    _fields_ = [
        ("backgroundlighting", C.c_float),
        ("diffusePower", C.c_float),
        ("ShadowFactor", C.c_float),
        ("MSAA", C.c_int),
        ("depthPower", C.c_float),
        ("backgroundColor", C.c_float * 3),
        ("focusDepth", C.c_float),
        ("shadowSamples", C.c_int),
        ("shadowRadius", C.c_float),
    ]


COUNTER_FIELDS = [
    "rays_closest", "rays_shadow", "tri_tests", "leaf_steps", "shaded_hits",
    "hit_lights", "child_rays", "camera_rays", "walk_guard",
    "moot_shadow_rays", "moot_tri_tests", "moot_leaf_steps",
]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}

    def flops(self):
        """Algorithmic flops F = 45 T + 40 S + 30 H + 60 Hl + 25 C."""
        return (45 * self.tri_tests + 40 * self.leaf_steps + 30 * self.shaded_hits
                + 60 * self.hit_lights + 25 * self.child_rays)

    def flops_moot(self):
        """The part of flops() spent on shadow queries whose answer cannot change the colour."""
        return 45 * self.moot_tri_tests + 40 * self.moot_leaf_steps


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rt_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    P = C.c_void_p
    fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    L.or_scene_load.restype = P
    L.or_scene_load.argtypes = [C.c_char_p, C.c_int]
    L.or_scene_from_arrays.restype = P
    L.or_scene_from_arrays.argtypes = [C.c_int, fp, C.c_int, ip, ip, ip, C.c_int, fp, C.c_int, fp]
    L.or_scene_free.argtypes = [P]
    L.or_compute_vertex_normals.argtypes = [P]
    L.or_set_lights.argtypes = [P, C.c_int, fp]
    L.or_scene_settings.restype = C.POINTER(Settings)
    L.or_scene_settings.argtypes = [P]
    for name in ("or_num_vertices", "or_num_triangles", "or_num_texcoords", "or_num_materials",
                 "or_num_textures", "or_num_nodes", "or_tree_depth", "or_num_node_tris"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [P]
    for name in ("or_vertex_pos", "or_vertex_nrm", "or_texcoords"):
        getattr(L, name).restype = C.POINTER(C.c_float)
        getattr(L, name).argtypes = [P]
    L.or_node_tris.restype = C.POINTER(C.c_int32)
    L.or_node_tris.argtypes = [P]
    L.or_get_triangles.argtypes = [P, ip, ip, ip]
    L.or_get_materials.argtypes = [P, fp]
    L.or_material_name.restype = C.c_char_p
    L.or_material_name.argtypes = [P, C.c_int]
    L.or_texture.restype = C.POINTER(C.c_ubyte)
    L.or_texture.argtypes = [P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.or_build_kdtree.argtypes = [P]
    L.or_get_nodes.argtypes = [P, fp, ip, ip, ip, ip, ip, ip]
    cp = C.POINTER(Counters)
    L.or_trace_rays.argtypes = [P, C.c_int, fp, fp, fp, fp, ip, C.c_int, cp]
    L.or_check_hits.argtypes = [P, C.c_int, fp, fp, ip, fp, fp, fp, fp, C.c_int, cp]
    L.or_brute_hits.argtypes = [P, C.c_int, fp, fp, ip, fp]
    L.or_guard_rays.argtypes = [fp, C.c_int]
    L.or_guard_rays.restype = C.c_int
    L.or_guard_reset.restype = None
    L.or_camera_corners.argtypes = [np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"),
                                    C.c_int, C.c_int, fp]
    L.or_pixel_ray.argtypes = [fp, C.c_int, C.c_int, C.c_double, C.c_double, fp, fp]
    L.or_render_rows.argtypes = [P, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, ip, C.c_int, cp]
    L.or_render_rows2.argtypes = [P, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, ip, C.c_int, cp, cp]
    L.or_normalize_depth.argtypes = [fp, C.c_int, C.c_float]
    L.or_gaussian_map.argtypes = [C.c_float, C.c_float, fp]
    L.or_blur.argtypes = [fp, fp, C.c_int, C.c_int, C.c_float, fp, C.c_int]
    L.or_to_bytes.argtypes = [fp, C.c_int, np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")]
    for name in ("or_write_bmp", "or_write_ppm"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.c_char_p, fp, C.c_int, C.c_int]
    L.or_write_depth_bmp.restype = C.c_int
    L.or_write_depth_bmp.argtypes = [C.c_char_p, C.c_void_p, fp, C.c_int, C.c_int]
    L.or_write_blurred_bmp.restype = C.c_int
    L.or_write_blurred_bmp.argtypes = [C.c_char_p, fp, C.c_int, C.c_int]
    L.or_write_blurred_bmp2.restype = C.c_int
    L.or_write_blurred_bmp2.argtypes = [C.c_char_p, fp, C.c_int, C.c_int,
                                        np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.c_size_t]
    L.or_update_bmp_rows.restype = C.c_int
    L.or_update_bmp_rows.argtypes = [C.c_char_p, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class Scene:
    """One loaded scene: mesh + normals + KD tree, following the reference's init()
    (raytracing.cpp:23-53)."""

    def __init__(self, handle):
        self.L = lib()
        self.h = handle

    @classmethod
    def load(cls, obj_path, textures=True, build=True):
        L = lib()
        h = L.or_scene_load(os.fsencode(obj_path), 1 if textures else 0)
        if not h:
            raise FileNotFoundError(obj_path)
        s = cls(h)
        L.or_compute_vertex_normals(h)
        if build:
            L.or_build_kdtree(h)
        return s

    @classmethod
    def from_arrays(cls, vpos, tri_v, tri_mat, materials, texcoords=None, tri_t=None, build=True):
        L = lib()
        vpos = _f32(vpos).reshape(-1, 3)
        tri_v = np.ascontiguousarray(tri_v, np.int32).reshape(-1, 3)
        tri_t = np.zeros_like(tri_v) if tri_t is None else np.ascontiguousarray(tri_t, np.int32)
        tri_mat = np.ascontiguousarray(tri_mat, np.int32)
        texcoords = np.zeros((1, 3), np.float32) if texcoords is None else _f32(texcoords)
        materials = _f32(materials).reshape(-1, 16)
        h = L.or_scene_from_arrays(len(vpos), vpos, len(tri_v), tri_v, tri_t, tri_mat,
                                   len(texcoords), texcoords, len(materials), materials)
        s = cls(h)
        L.or_compute_vertex_normals(h)
        if build:
            L.or_build_kdtree(h)
        return s

    def __del__(self):
        try:
            if self.h:
                self.L.or_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    # ---- scene data
    @property
    def settings(self):
        return self.L.or_scene_settings(self.h).contents

    def set_lights(self, lights):
        lights = _f32(lights).reshape(-1, 3)
        self.L.or_set_lights(self.h, len(lights), lights)

    def counts(self):
        L, h = self.L, self.h
        return dict(vertices=L.or_num_vertices(h), triangles=L.or_num_triangles(h),
                    texcoords=L.or_num_texcoords(h), materials=L.or_num_materials(h),
                    textures=L.or_num_textures(h), nodes=L.or_num_nodes(h),
                    depth=L.or_tree_depth(h), node_tris=L.or_num_node_tris(h))

    def vertices(self):
        n = self.L.or_num_vertices(self.h)
        return (_arr(self.L.or_vertex_pos(self.h), 3 * n, np.float32).reshape(n, 3),
                _arr(self.L.or_vertex_nrm(self.h), 3 * n, np.float32).reshape(n, 3))

    def texcoords(self):
        n = self.L.or_num_texcoords(self.h)
        return _arr(self.L.or_texcoords(self.h), 3 * n, np.float32).reshape(n, 3)

    def triangles(self):
        n = self.L.or_num_triangles(self.h)
        v = np.zeros((n, 3), np.int32)
        t = np.zeros((n, 3), np.int32)
        m = np.zeros(n, np.int32)
        self.L.or_get_triangles(self.h, v, t, m)
        return v, t, m

    def materials(self):
        n = self.L.or_num_materials(self.h)
        out = np.zeros((n, 16), np.float32)
        self.L.or_get_materials(self.h, out)
        return out

    def material_names(self):
        return [self.L.or_material_name(self.h, i).decode() for i in
                range(self.L.or_num_materials(self.h))]

    def textures(self):
        out = []
        for i in range(self.L.or_num_textures(self.h)):
            w, h = C.c_int(), C.c_int()
            p = self.L.or_texture(self.h, i, C.byref(w), C.byref(h))
            n = (h.value + 2) * w.value * 4
            out.append((np.ctypeslib.as_array(p, shape=(n,)).copy(), w.value, h.value))
        return out

    def nodes(self):
        n = self.L.or_num_nodes(self.h)
        b = np.zeros((n, 6), np.float32)
        arrs = [np.zeros(n, np.int32) for _ in range(6)]
        self.L.or_get_nodes(self.h, b, *arrs)
        nt = self.L.or_num_node_tris(self.h)
        return dict(bounds=b, left=arrs[0], right=arrs[1], parent=arrs[2], depth=arrs[3],
                    tri_off=arrs[4], tri_cnt=arrs[5],
                    tris=_arr(self.L.or_node_tris(self.h), nt, np.int32))

    # ---- tracing
    def trace_rays(self, origins, dests, threads=1, counters=None):
        o, d = _f32(origins).reshape(-1, 3), _f32(dests).reshape(-1, 3)
        n = len(o)
        rgb = np.zeros((n, 3), np.float32)
        minT = np.zeros(n, np.float32)
        tri = np.zeros(n, np.int32)
        self.L.or_trace_rays(self.h, n, o, d, rgb, minT, tri, threads,
                             C.byref(counters) if counters is not None else None)
        return rgb, minT, tri

    def check_hits(self, origins, dests, threads=1, counters=None):
        o, d = _f32(origins).reshape(-1, 3), _f32(dests).reshape(-1, 3)
        n = len(o)
        tri = np.zeros(n, np.int32)
        a, b, t = (np.zeros(n, np.float32) for _ in range(3))
        hp = np.zeros((n, 3), np.float32)
        self.L.or_check_hits(self.h, n, o, d, tri, a, b, t, hp, threads,
                             C.byref(counters) if counters is not None else None)
        return tri, a, b, t, hp

    def brute_hits(self, origins, dests):
        o, d = _f32(origins).reshape(-1, 3), _f32(dests).reshape(-1, 3)
        tri = np.zeros(len(o), np.int32)
        t = np.zeros(len(o), np.float32)
        self.L.or_brute_hits(self.h, len(o), o, d, tri, t)
        return tri, t

    def render(self, corners, W, H, y0=0, y1=None, threads=1, counters=None, centre=None,
               out=None):
        """Rows [y0, y1) of a frame.  `centre` receives the work of the pixel-centre
        rays alone.  `out` = (rgb, depth, tri) to render several bands into one frame."""
        y1 = H if y1 is None else y1
        corners = _f32(corners).reshape(24)
        if out is None:
            out = (np.zeros((H, W, 3), np.float32), np.zeros((H, W), np.float32),
                   np.full((H, W), -1, np.int32))
        rgb, depth, tri = out
        self.L.or_render_rows2(self.h, corners, W, H, y0, y1, rgb, depth, tri, threads,
                               C.byref(counters) if counters is not None else None,
                               C.byref(centre) if centre is not None else None)
        return rgb, depth, tri


def guard_rays(reset=True):
    """Rays whose leaf walk the step guard stopped since the last reset: (n, 6) origin + destination."""
    out = np.zeros((64, 6), np.float32)
    n = lib().or_guard_rays(out, 64)
    if reset:
        lib().or_guard_reset()
    return out[:min(n, 64)].copy()


def camera_corners(modelview, W, H):
    m = np.ascontiguousarray(modelview, np.float64).reshape(16)
    out = np.zeros(24, np.float32)
    lib().or_camera_corners(m, W, H, out)
    return out


def pixel_ray(corners, W, H, x, y):
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    lib().or_pixel_ray(_f32(corners).reshape(24), W, H, float(x), float(y), o, d)
    return o, d


def normalize_depth(depth, focus=0.0):
    d = _f32(depth).copy().reshape(-1)
    lib().or_normalize_depth(d, d.size, focus)
    return d.reshape(np.shape(depth))


def gaussian_map(intensity, depth_power=2.5):
    m = np.zeros(169, np.float32)
    lib().or_gaussian_map(intensity, depth_power, m)
    return m.reshape(13, 13)


def blur(rgb, depthN, depth_power=2.5, threads=1):
    rgb = _f32(rgb)
    H, W = rgb.shape[:2]
    out = np.zeros((H - 12, W - 12, 3), np.float32)
    lib().or_blur(rgb.reshape(-1), _f32(depthN).reshape(-1), W, H, depth_power, out.reshape(-1),
                  threads)
    return out


def to_bytes(v):
    v = _f32(v)
    out = np.zeros(v.size, np.uint8)
    lib().or_to_bytes(v.reshape(-1), v.size, out)
    return out.reshape(v.shape)


def write_result_files(out_dir, rgb, depth, focus=0.0, depth_power=2.5, threads=1):
    """The files of the 'r' path in the reference's order (main.cpp:325, 404-407, 371-377):
    progress.bmp (blank result.bmp + one per-row update per row), result.ppm, result.bmp,
    depth.bmp (depth normalised once), blurred.bmp (normalised twice, then blurred).
    rgb is the clamped image, depth the raw depth plane.  Returns (depth1, depth2, blurred)."""
    L = lib()
    rgb = _f32(rgb)
    H, W = rgb.shape[:2]
    p = lambda n: os.fsencode(os.path.join(out_dir, n))  # noqa: E731
    L.or_write_bmp(p("progress.bmp"), np.zeros_like(rgb).reshape(-1), W, H)
    for y in range(H):
        L.or_update_bmp_rows(p("progress.bmp"), rgb.reshape(-1), W, 0, y, W, 1)
    L.or_write_ppm(p("result.ppm"), rgb.reshape(-1), W, H)
    L.or_write_bmp(p("result.bmp"), rgb.reshape(-1), W, H)
    d1 = normalize_depth(depth, focus)
    L.or_write_depth_bmp(p("depth.bmp"), None, d1.reshape(-1), W, H)
    d2 = normalize_depth(d1, focus)
    bl = None
    if W > 12 and H > 12:
        bl = blur(rgb, d2, depth_power, threads)
        prev = np.fromfile(os.path.join(out_dir, "depth.bmp"), np.uint8)[54:].copy()
        L.or_write_blurred_bmp2(p("blurred.bmp"), bl.reshape(-1), W, H, prev, prev.size)
    return d1, d2, bl


def read_camera_sav(path):
    """camera.sav as written by the reference (main.cpp:270-280): 16 floats of the
    column-major modelview, the light count, then one light per line."""
    tok = open(path, "rb").read().split()
    # fscanf("%f") into a float, then widened (main.cpp:285-288)
    m = np.array([float(x) for x in tok[:16]], np.float32).astype(np.float64)
    n = int(tok[16])
    lights = np.array([float(x) for x in tok[17:17 + 3 * n]], np.float32).reshape(n, 3)
    return m, lights
