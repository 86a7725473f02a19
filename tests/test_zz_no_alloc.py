"""rt_normalize_depth_device_noalloc (the entry point that allocates nothing) against
rt_normalize_depth_device, through the C ABI: two calls in a row on one stream, the second with
a smaller input, bit-exact; sizes beyond the stated maximum are rejected."""
import ctypes as C

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
FLT_MAX = np.float32(3.402823466e+38)


def _depth(n, seed):
    rng = np.random.default_rng(seed)
    d = (rng.random(n, dtype=np.float32) * 0.8 + 0.1).astype(np.float32)
    d[rng.integers(0, n, max(1, n // 50))] = FLT_MAX
    return d


def test_noalloc_normalisation_equals_the_allocating_entry_point():
    import torch
    P = pkg()
    L = P.capi.lib()
    st = torch.cuda.current_stream().cuda_stream
    for n, focus, seed in ((1920 * 1080, 0.0, 5), (70001, 0.37, 6)):   # the second call is the smaller one
        d = _depth(n, seed)
        a, b = torch.from_numpy(d).cuda(), torch.from_numpy(d).cuda()
        P.capi.check(L.rt_normalize_depth_device(0, n, focus, C.c_void_p(a.data_ptr()), C.c_void_p(st)))
        P.capi.check(L.rt_normalize_depth_device_noalloc(0, n, focus, C.c_void_p(b.data_ptr()), C.c_void_p(st)))
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        assert float(b.max()) == 1.0 and float(b.min()) == 0.0
    # same stream, back to back without a synchronisation in between: the second normalisation
    # (as the 'r' path runs it) reuses the array the first one used
    d = _depth(300007, 7)
    a, b = torch.from_numpy(d).cuda(), torch.from_numpy(d).cuda()
    for _ in range(2):
        P.capi.check(L.rt_normalize_depth_device(0, d.size, 0.41, C.c_void_p(a.data_ptr()), C.c_void_p(st)))
        P.capi.check(L.rt_normalize_depth_device_noalloc(0, d.size, 0.41, C.c_void_p(b.data_ptr()), C.c_void_p(st)))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # after the calls above (the warm-up) the entry point takes no device memory: the free memory the
    # driver reports is the same before and after a run of calls
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        P.capi.check(L.rt_normalize_depth_device_noalloc(0, d.size, 0.41, C.c_void_p(b.data_ptr()), C.c_void_p(st)))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0


def test_noalloc_rejects_what_it_cannot_serve():
    P = pkg()
    L = P.capi.lib()
    too_many = (1 << 28) + 1
    assert L.rt_normalize_depth_device_noalloc(0, too_many, 0.0, C.c_void_p(16), None) == -1
    assert b"exceeds the maximum" in L.rt_last_error()
    assert L.rt_normalize_depth_device_noalloc(0, -1, 0.0, C.c_void_p(16), None) == -1
    assert L.rt_normalize_depth_device_noalloc(0, 5, 0.0, None, None) == -1
    assert L.rt_normalize_depth_device_noalloc(0, 0, 0.0, None, None) == 0
    assert L.rt_normalize_depth_device_noalloc(99, 5, 0.0, C.c_void_p(16), None) == -1
