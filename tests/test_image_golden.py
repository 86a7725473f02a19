"""The image half of the hot path against vectors made by the reference's own
imageWriter.h (compiled as oracle/_ref/refimage, run by
tools/make_image_golden.py; fixtures tests/golden/image_*.npz):

  RGBValue clamp             imageWriter.h:23-44
  normailzeDepthRange x1/x2  imageWriter.h:113-135 (focus 0 and != 0)
  gauseanMap                 imageWriter.h:137-181, the 't'-key list of main.cpp:238-245
  blurrAllLines              imageWriter.h:210-237
  writers + per-row update   imageWriter.h:253-488, in the order of main.cpp:325, 404-407, 371-377

CPU tests pin the oracle and the host writers (bit / byte exact); GPU tests
compare the HIP kernels with the same vectors through the C ABI."""
import os

import numpy as np
import pytest

from conftest import load_golden, pkg

CASES = ["aligned_64x48_focus0", "ragged_63x47_focus037", "final_96x54_focus0", "final_90x50_focus05"]
FILES = ["progress.bmp", "result.ppm", "result.bmp", "depth.bmp", "blurred.bmp"]


@pytest.fixture(scope="module", params=CASES)
def g(request):
    return load_golden("image_" + request.param)


def same_or_both_nan(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def file_bytes(d, name):
    return np.fromfile(os.path.join(d, name), np.uint8)


# ---------------------------------------------------------------- CPU: oracle == reference
def test_oracle_depth_normalisation_equals_reference(g, oracle):
    focus = float(g["focus"])
    d1 = oracle.normalize_depth(g["depth_in"], focus)
    assert np.array_equal(d1, g["depth1"])
    assert np.array_equal(oracle.normalize_depth(d1, focus), g["depth2"])
    if focus != 0.0:   # the second pass is not the identity once a focus is set
        assert not np.array_equal(g["depth1"], g["depth2"])


def test_oracle_gaussian_maps_equal_reference(g, oracle):
    for t, ref in zip(g["t_key_list"], g["maps"]):
        assert same_or_both_nan(oracle.gaussian_map(float(t)), ref), float(t)
    assert np.isnan(g["maps"][6]).all()           # intensity 1.0: 0/0 for every tap
    assert abs(float(g["maps"][3].sum()) - 1.0) < 1e-5   # normalised kernels


def test_oracle_blur_equals_reference(g, oracle):
    assert np.array_equal(np.clip(g["rgb_in"], 0, 1), g["image"])     # RGBValue clamp
    assert same_or_both_nan(oracle.blur(g["image"], g["depth2"], threads=4), g["blurr"])
    assert np.isnan(g["blurr"]).any()             # the farthest pixel's kernel is 0/0


def test_oracle_files_equal_reference(g, oracle, tmp_path):
    oracle.write_result_files(str(tmp_path), g["image"], g["depth_in"], float(g["focus"]), threads=4)
    for name in FILES:
        assert np.array_equal(file_bytes(tmp_path, name), g[name.replace(".", "_")]), name


# ---------------------------------------------------------------- CPU: host writers == reference
def test_host_files_equal_reference(g, tmp_path):
    pkg().host.write_result_files(str(tmp_path), g["image"], g["depth1"], g["blurr"])
    for name in FILES:
        assert np.array_equal(file_bytes(tmp_path, name), g[name.replace(".", "_")]), name


# ---------------------------------------------------------------- GPU kernels == reference
@pytest.mark.gpu
def test_gpu_depth_normalisation_equals_reference(g):
    capi = pkg().capi
    focus = float(g["focus"])
    d1 = capi.normalize_depth(g["depth_in"], focus)
    assert np.array_equal(d1, g["depth1"])
    assert np.array_equal(capi.normalize_depth(d1, focus), g["depth2"])


@pytest.mark.gpu
def test_gpu_blur_and_bytes_equal_reference(g):
    capi = pkg().capi
    got, ref = capi.dof_blur(g["image"], g["depth2"]), g["blurr"]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    # the kernel's exp2-based powf is within 2 ulp of glibc's; weights are <= 1, colours in [0, 1]
    assert np.abs(got[ok] - ref[ok]).max() < 2e-6
    H, W = g["depth_in"].shape
    # bytes of the files: result.ppm = quantised image; blurred.bmp from the GPU's blur through the
    # host writer, at most 1 LSB from the reference's file on at least 99.9 % of the bytes
    ppm = g["result_ppm"]
    assert np.array_equal(capi.quantize(g["image"]).reshape(-1), ppm[ppm.size - 3 * W * H:])


@pytest.mark.gpu
def test_gpu_r_path_files_equal_reference(g, tmp_path):
    """Depth normalisation x2 and the blur on the GPU, files by the host writers."""
    P = pkg()
    focus = float(g["focus"])
    d1 = P.capi.normalize_depth(g["depth_in"], focus)
    d2 = P.capi.normalize_depth(d1, focus)
    P.host.write_result_files(str(tmp_path), g["image"], d1, P.capi.dof_blur(g["image"], d2))
    for name in FILES[:4]:
        assert np.array_equal(file_bytes(tmp_path, name), g[name.replace(".", "_")]), name
    a, b = file_bytes(tmp_path, "blurred.bmp").astype(int), g["blurred_bmp"].astype(int)
    assert a.size == b.size and np.array_equal(a[:54], b[:54])
    assert np.abs(a - b).max() <= 1 and (a == b).mean() >= 0.999


@pytest.mark.gpu
def test_gpu_fused_postprocess_equals_reference_and_the_separate_calls(g):
    """rt_postprocess_device: first normalisation in place, second one folded into the blur."""
    import ctypes as C

    import torch
    P = pkg()
    L = P.capi.lib()
    focus = float(g["focus"])
    H, W = g["depth_in"].shape
    rgb = torch.from_numpy(g["image"]).cuda()
    depth = torch.from_numpy(g["depth_in"]).cuda()
    out = torch.zeros((H - 12, W - 12, 3), dtype=torch.float32, device="cuda")
    P.capi.check(L.rt_postprocess_device(0, W, H, focus, 2.5, C.c_void_p(rgb.data_ptr()), C.c_void_p(depth.data_ptr()),
                                         C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.array_equal(depth.cpu().numpy(), g["depth1"])
    got, ref = out.cpu().numpy(), g["blurr"]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.abs(got[ok] - ref[ok]).max() < 2e-6
    d2 = P.capi.normalize_depth(P.capi.normalize_depth(g["depth_in"], focus), focus)
    sep = P.capi.dof_blur(g["image"], d2)
    assert np.array_equal(got[ok], sep[ok])
    # without a blur buffer only the depth plane is produced
    depth = torch.from_numpy(g["depth_in"]).cuda()
    P.capi.check(L.rt_postprocess_device(0, W, H, focus, 2.5, None, C.c_void_p(depth.data_ptr()), None, None))
    torch.cuda.synchronize()
    assert np.array_equal(depth.cpu().numpy(), g["depth1"])
