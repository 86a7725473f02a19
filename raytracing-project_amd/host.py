"""ctypes view of librt_host.so, the C++ host side (scene loader, KD build, camera,
image writers, the 'r' render path).  Plumbing for tests and bench.py."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librt_host.so")

EXPORTS = ["rth_scene_load", "rth_scene_free", "rth_scene_desc", "rth_tree_depth",
           "rth_material_name", "rth_read_camera_sav", "rth_camera_corners", "rth_write_ppm",
           "rth_write_bmp", "rth_write_depth_bmp", "rth_write_blurred_bmp", "rth_render_to_files",
           "rth_last_error", "rth_decode_png", "rth_render_to_files_focus", "rth_write_result_files", "rth_render_to_files_float",
           "rth_time_r_path"]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'`")
    capi.lib()  # librt_hip.so first (rpath also finds it)
    L = C.CDLL(LIB_PATH)
    P = C.c_void_p
    L.rth_scene_load.restype = P
    L.rth_scene_load.argtypes = [C.c_char_p, C.c_int]
    L.rth_scene_free.argtypes = [P]
    L.rth_scene_free.restype = None
    L.rth_scene_desc.restype = C.POINTER(capi.RtSceneDesc)
    L.rth_scene_desc.argtypes = [P]
    L.rth_tree_depth.argtypes = [P]
    L.rth_material_name.restype = C.c_char_p
    L.rth_material_name.argtypes = [P, C.c_int]
    L.rth_read_camera_sav.argtypes = [C.c_char_p, P, P, C.c_int]
    L.rth_camera_corners.argtypes = [P, C.c_int, C.c_int, P]
    L.rth_camera_corners.restype = None
    for n in ("rth_write_ppm", "rth_write_bmp", "rth_write_depth_bmp", "rth_write_blurred_bmp"):
        getattr(L, n).argtypes = [C.c_char_p, P, C.c_int, C.c_int]
    L.rth_write_result_files.argtypes = [C.c_char_p, P, P, P, C.c_int, C.c_int]
    L.rth_render_to_files.restype = C.c_double
    L.rth_render_to_files.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                      C.c_int, C.POINTER(capi.RtStats)]
    L.rth_render_to_files_float.restype = C.c_double
    L.rth_render_to_files_float.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                            C.c_int]
    L.rth_time_r_path.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                  C.POINTER(C.c_double)]
    L.rth_last_error.restype = C.c_char_p
    L.rth_render_to_files_focus.restype = C.c_double
    L.rth_render_to_files_focus.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                            C.c_int, C.c_double, C.c_double, C.POINTER(C.c_float), P]
    L.rth_decode_png.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), P, C.c_size_t]
    _lib = L
    return L


def _view(ptr, n, ctype, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).astype(dtype, copy=True)


class HostScene:
    """A scene loaded and indexed by the C++ host code (Mesh::loadMesh,
    computeVertexNormals, buildKDtree)."""

    def __init__(self, obj_path, textures=True):
        self.h = lib().rth_scene_load(os.fsencode(obj_path), 1 if textures else 0)
        if not self.h:
            raise FileNotFoundError(obj_path)
        self.desc = lib().rth_scene_desc(self.h).contents

    def close(self):
        if getattr(self, "h", None):
            lib().rth_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def arrays(self):
        d = self.desc
        nv, nt, ntc, nm, nn = d.n_vertices, d.n_triangles, d.n_texcoords, d.n_materials, d.n_nodes
        mats = np.zeros((nm, 16), np.float32)
        m_arr = C.cast(d.materials, C.POINTER(capi.RtMaterial))
        for i in range(nm):
            m = m_arr[i]
            mats[i, 0:3], mats[i, 3:6] = list(m.Kd), list(m.Ks)
            mats[i, 6:11] = [m.Ns, m.Ni, m.illum, m.has_Ni, m.texture]
        n_arr = C.cast(d.nodes, C.POINTER(capi.RtKdNode))
        bounds = np.zeros((nn, 6), np.float32)
        left, right, off, cnt = (np.zeros(nn, np.int32) for _ in range(4))
        for i in range(nn):
            n = n_arr[i]
            bounds[i] = list(n.bounds)
            left[i], right[i], off[i], cnt[i] = n.left, n.right, n.tri_offset, n.tri_count
        texs = []
        t_arr = C.cast(d.textures, C.POINTER(capi.RtTexture))
        for i in range(d.n_textures):
            t = t_arr[i]
            texs.append((_view(t.rgba, t.width * t.height * 4, C.c_ubyte, np.uint8), t.width, t.height))
        return dict(
            vertex_pos=_view(d.vertex_pos, 3 * nv, C.c_float, np.float32).reshape(nv, 3),
            vertex_normal=_view(d.vertex_normal, 3 * nv, C.c_float, np.float32).reshape(nv, 3),
            tri_vertex=_view(d.tri_vertex, 3 * nt, C.c_int32, np.int32).reshape(nt, 3),
            tri_texcoord=_view(d.tri_texcoord, 3 * nt, C.c_int32, np.int32).reshape(nt, 3),
            tri_material=_view(d.tri_material, nt, C.c_int32, np.int32),
            texcoords=_view(d.texcoords, 3 * ntc, C.c_float, np.float32).reshape(ntc, 3),
            materials=mats, textures=texs,
            material_names=[lib().rth_material_name(self.h, i).decode() for i in range(nm)],
            node_bounds=bounds, node_left=left, node_right=right, node_tri_off=off,
            node_tri_cnt=cnt, node_tris=_view(d.node_tris, d.n_node_tris, C.c_int32, np.int32),
            tree_depth=lib().rth_tree_depth(self.h))

    def to_gpu(self, device=0):
        """RtScene on `device`, created straight from the host arrays."""
        return GpuScene(self, device)


class GpuScene(capi.Scene):
    def __init__(self, host_scene, device=0):  # noqa: super().__init__ deliberately not called
        self._keep = [host_scene]
        self.device = device
        h = C.c_void_p()
        capi.check(capi.lib().rt_scene_create(C.byref(host_scene.desc), device, C.byref(h)))
        self.h = h


def decode_png(data):
    """PNG bytes -> (H, W, 4) uint8 RGBA, the host library's decoder (png_decoder.cpp)."""
    w, h = C.c_int(0), C.c_int(0)
    if lib().rth_decode_png(data, len(data), C.byref(w), C.byref(h), None, 0) != 0:
        raise ValueError(lib().rth_last_error().decode(errors="replace"))
    out = np.zeros((h.value, w.value, 4), np.uint8)
    if lib().rth_decode_png(data, len(data), None, None, out.ctypes.data, out.size) != 0:
        raise ValueError(lib().rth_last_error().decode(errors="replace"))
    return out


def read_camera_sav(path):
    m = np.zeros(16, np.float64)
    lights = np.zeros((capi.RT_MAX_LIGHTS, 3), np.float32)
    n = lib().rth_read_camera_sav(os.fsencode(path), m.ctypes.data, lights.ctypes.data,
                                  capi.RT_MAX_LIGHTS)
    if n < 0:
        raise FileNotFoundError(path)
    return m, lights[:n].copy()


def camera_corners(modelview, W, H):
    m = np.ascontiguousarray(modelview, np.float64).reshape(16)
    out = np.zeros(24, np.float32)
    lib().rth_camera_corners(m.ctypes.data, W, H, out.ctypes.data)
    return out


def _write(fn, path, arr, W, H):
    a = np.ascontiguousarray(arr, np.float32)
    if not getattr(lib(), fn)(os.fsencode(path), a.ctypes.data, W, H):
        raise OSError(f"{fn}({path}) failed")


def write_ppm(path, rgb):
    _write("rth_write_ppm", path, rgb, rgb.shape[1], rgb.shape[0])


def write_bmp(path, rgb):
    _write("rth_write_bmp", path, rgb, rgb.shape[1], rgb.shape[0])


def write_depth_bmp(path, depth_n):
    _write("rth_write_depth_bmp", path, depth_n, depth_n.shape[1], depth_n.shape[0])


def write_blurred_bmp(path, blurred):
    _write("rth_write_blurred_bmp", path, blurred, blurred.shape[1] + 12, blurred.shape[0] + 12)


def write_result_files(out_dir, rgb, depth_n, blurred):
    """progress.bmp (blank file + per-row updates), result.ppm, result.bmp, depth.bmp and
    blurred.bmp written by one Image in the order of the reference's 'r' path."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    H, W = rgb.shape[:2]
    dn = np.ascontiguousarray(depth_n, np.float32)
    bl = np.ascontiguousarray(blurred, np.float32) if blurred is not None else None
    if not lib().rth_write_result_files(os.fsencode(out_dir), rgb.ctypes.data, dn.ctypes.data,
                                        bl.ctypes.data if bl is not None else None, W, H):
        raise OSError(f"rth_write_result_files({out_dir}) failed")


def render_to_files(obj_path, camera_sav, W, H, msaa, out_dir, device=0):
    st = capi.RtStats()
    secs = lib().rth_render_to_files(os.fsencode(obj_path), os.fsencode(camera_sav), W, H, msaa,
                                     os.fsencode(out_dir), device, C.byref(st))
    if secs < 0:
        raise RuntimeError(lib().rth_last_error().decode(errors="replace"))
    return secs, st


def render_to_files_float(obj_path, camera_sav, W, H, msaa, out_dir, device=0, row_updates=False):
    """The 'r' path through the float planes (as the reference's Image holds them)."""
    secs = lib().rth_render_to_files_float(os.fsencode(obj_path), os.fsencode(camera_sav), W, H, msaa,
                                           os.fsencode(out_dir), device, 1 if row_updates else 0)
    if secs < 0:
        raise RuntimeError(lib().rth_last_error().decode(errors="replace"))
    return secs


def time_r_path(obj_path, camera_sav, W, H, msaa, out_dir, device=0, runs=2):
    """Wall time of RayTracer::doDaRayTracingShizz on a resident scene (last of `runs`)."""
    out = (C.c_double * 2)()
    rc = lib().rth_time_r_path(os.fsencode(obj_path), os.fsencode(camera_sav), W, H, msaa, os.fsencode(out_dir),
                               device, runs, out)
    if rc != 0:
        raise RuntimeError(lib().rth_last_error().decode(errors="replace"))
    return {"frame_call_ms": round(out[0] * 1e3, 2), "file_writes_ms": round(out[1] * 1e3, 2),
            "total_ms": round((out[0] + out[1]) * 1e3, 2),
            "what": "frame_call = GPU frame + depth passes + blur + 8-bit planes to pinned host memory; "
                    "file_writes = result.ppm, result.bmp, depth.bmp, blurred.bmp by the host writers"}


def render_to_files_focus(obj_path, camera_sav, W, H, msaa, out_dir, fx, fy, device=0):
    """The 'r' path after key 'f' at pixel (fx, fy): returns (seconds, focus depth, RtStats)."""
    st = capi.RtStats()
    focus = C.c_float(0)
    secs = lib().rth_render_to_files_focus(os.fsencode(obj_path), os.fsencode(camera_sav), W, H, msaa,
                                           os.fsencode(out_dir), device, fx, fy, C.byref(focus), C.byref(st))
    if secs < 0:
        raise RuntimeError(lib().rth_last_error().decode(errors="replace"))
    return secs, focus.value, st
