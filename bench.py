#!/usr/bin/env python3
"""Headline benchmark: frames of final.obj at 3840x2160, MSAA 4, depth 5, with the
depth-of-field pass, on N GPUs of one node (BASELINE.json configs[4]).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one frame: every rank traces its screen tiles (32x32, dealt
round-robin), the tile payloads are gathered on rank 0 (RCCL), rank 0 runs the
depth normalisation (twice, as the reference does) and the blur.  Scene,
camera and lights are resident on the GPUs before the timed region; the frame
stays in HBM.  Rank 0 prints one JSON line.

value = rays traced by all ranks per second (closest-hit + shadow queries the
GPUs executed).  roofline.achieved = algorithmic FLOP of the dominant kernel
(k_wf_pipeline, the persistent launch that renders a rank's part of the frame)
divided by its duration from HIP events recorded inside the timed loop; the FLOP
count is F = 45 T + 40 S + 30 H + 60 Hl + 25 C with the counters of the
reference's algorithm (tests/golden/frame_counters.json, produced by
tools/make_counters.py on the CPU oracle), less the work whose result the
reference never uses (shading of the pixel-centre rays, shadow queries that
cannot change shadowMax).  That figure measures the frame against a machine
running the REFERENCE's walk at peak rate; roofline.executed prices, with the
same weights, the tests, node steps, shaded hits and child rays the launch
really executed (its own counters) - that one is the hardware utilisation.

After the timed loop, outside it, the last frame is checked: sparse pixels
against the committed fixture of the CPU oracle, blurred pixels against the
oracle's kernel; a mismatch fails the run ("checked" in the line).  On one GPU
the line also carries a record per other BASELINE config ("configs"), the wall
time of the drop-in 'r' path ("r_path_ms"), an eighth of the frame alone and with two frames in flight
("frame_parts_in_flight": what a rank of eight GPUs would render) and the CPU baseline.  For every N the
K steps are timed twice: with one frame in flight (the line's value and ms_per_step: the time of a
frame) and with two ("two_frames_in_flight": the throughput of consecutive frames), and a third time with
the regions of every launch handed out by index instead of by the durations the frame before measured
("regions_by_index": the first frame of a view).
"""
import argparse
import importlib
import json
import os
import sys
import time

# Frames in flight (several GPUs: two per rank, see below) only overlap when their streams sit on
# different hardware queues.  The runtime's default of four queues per process, shared with every other
# stream the process makes, left two frames in flight running one after the other (an eighth of the
# 4K frame: 1.90 ms per part); with eight queues they overlap (1.24 ms).  Read by the HIP runtime when
# it starts, so it is set before anything touches the GPU - and only if the environment does not set it
# already (a pool that fixes it at four keeps four; the partner probe below copes with that).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE, W, H, MSAA = "final", 3840, 2160, 4
TILE = 32
FP32_PEAK_TFLOPS = 157.3   # MI355X FP32 vector peak (= f32 MFMA peak)
HBM_PEAK_GBS = 8000.0


def cpu_share():
    """Host threads this process may use: the affinity mask, capped by the pool's per-GPU share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))


def cpu_baseline(corners, lights):
    """The oracle (CPU restatement of the reference) on a bounded sample of the same frame,
    by the reference's own protocol (key 'R', main.cpp:261-267: mean of 5 runs) after one
    warm-up run, at two thread counts: the reference's policy
    min(16, hardware threads - 1) (config.h.dist:23, main.cpp:331) and every thread we may
    use.  The sample is 6 bands of rows spread over the image; the per-frame figure is
    extrapolated from it.  Checker code used as a reported baseline, never as the product."""
    from oracle import oracle as O
    share = cpu_share()
    s = O.Scene.load(os.path.join(ROOT, "scenes", SCENE + ".obj"))
    s.set_lights(lights)
    s.settings.MSAA = MSAA
    bands = 6

    def sample(rows, threads):
        c = O.Counters()
        out = None
        t0 = time.perf_counter()
        for b in range(bands):
            y0 = min(int((b + 0.5) * H / bands), H - rows)
            out = s.render(corners, W, H, y0=y0, y1=y0 + rows, threads=threads, counters=c, out=out)
        return c, time.perf_counter() - t0

    policies = [("reference_policy", max(1, min(16, share - 1))), ("all_threads", share)]
    res = {}
    rows = 2
    for name, threads in policies:
        c, dt = sample(2, threads)                   # warm-up, also sizes the sample: ~1.6 s per run
        rows = int(max(2, min(H // bands, 2 * 1.6 / max(dt, 1e-3))))
        times = []
        for _ in range(5):
            c, dt = sample(rows, threads)
            times.append(dt)
        dt = sum(times) / len(times)
        rays = c.rays_closest + c.rays_shadow
        res[name] = {"threads": threads, "mrays_s": round(rays / dt / 1e6, 3),
                     "ms_per_frame_extrapolated": round(dt * H / (bands * rows) * 1e3, 1),
                     "rows": bands * rows, "runs": 5, "mean_run_s": round(dt, 3)}
    best = max(res.values(), key=lambda r: r["mrays_s"])
    return {"value": best["mrays_s"], "unit": "Mrays/s", "cores": best["threads"], "kind": "port",
            "value_counts": "the queries the REFERENCE's walk makes (251 M per frame, discarded centre-ray subtrees and "
                            "moot shadow queries included); the GPU line's value counts the 128 M the GPU traces - "
                            "compare ms per frame (ms_per_frame_extrapolated against ms_per_step), not Mrays/s",
            "host_cpu_count": os.cpu_count(), "cpu_share": share, "extrapolated": True,
            "protocol": "mean of 5 runs after 1 warm-up (main.cpp:261-267), render loop only, no file writes",
            "sample": f"{best['rows']} of {H} rows ({bands} bands) of {SCENE}.obj {W}x{H} MSAA{MSAA}, no DoF, "
                      f"5 x {best['mean_run_s']:.1f} s",
            "ms_per_frame_extrapolated": best["ms_per_frame_extrapolated"], "policies": res}


OTHER_CONFIGS = [("ball", 800, 800, 1), ("redballs", 1920, 1080, 1), ("refractionballs", 1920, 1080, 1),
                 ("lego", 3840, 2160, 4),
                 ("fin", 1920, 1080, 1)]   # not a BASELINE config: the reference's largest scene, with refraction


def r_path_ms(pkg, dev_index):
    """Wall time of the drop-in 'r' path for the headline config through the C++ host
    (RayTracer::doDaRayTracingShizz: frame + post-process on the GPU, planes to the host, four
    files written), second of two runs; the scene is loaded and resident before the clock starts."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        t = pkg.host.time_r_path(os.path.join(ROOT, "scenes", SCENE + ".obj"),
                                 os.path.join(ROOT, "scenes", "cameras", SCENE + ".sav"), W, H, MSAA, tmp,
                                 dev_index, runs=2)
    return t


def frame_parts_in_flight(pkg, hs, corners, lights, dev_index, parts=8, frames=32):
    """What one rank of `parts` GPUs would render, measured on this GPU: its share of the tiles of the
    headline frame, alone and with two such frames in flight on streams of their own (as the timed loop
    keeps them with several GPUs).  Render only: no gather, no post-process."""
    import torch
    capi = pkg.capi
    p = capi.default_params(W, H, MSAA)
    p.tile_w = p.tile_h = TILE
    p.world, p.rank = parts, 0
    res = {"parts": parts, "what": f"rank 0's tiles of a {parts}-GPU deal rendered on one GPU, ms per frame part"
                                   if parts > 1 else "the whole frame, render only, ms per frame: one after the other "
                                   "and with two in flight (throughput; the timed loop on one GPU keeps one in flight)"}
    dev = f"cuda:{dev_index}"
    slots = [{"gs": hs.to_gpu(dev_index), "st": torch.cuda.Stream(),
              "rgb": torch.zeros((H, W, 3), dtype=torch.float32, device=dev),
              "depth": torch.zeros((H, W), dtype=torch.float32, device=dev)} for _ in range(4)]

    def go(which, n):
        for i in range(n):
            sl = slots[which[i % len(which)]]
            sl["gs"].render_device(corners, lights, p, sl["rgb"].data_ptr(), sl["depth"].data_ptr(), 0,
                                   sl["st"].cuda_stream, False)
        torch.cuda.synchronize()

    def ms(which):
        go(which, 2 * len(which))
        t0 = time.perf_counter()
        go(which, frames)
        return round((time.perf_counter() - t0) * 1e3 / frames, 3)

    res["one_in_flight_ms"] = ms([0])
    # whether two launches overlap depends on the hardware queues their streams were given: try the pairs
    pairs = {f"0+{k}": ms([0, k]) for k in (1, 2, 3)}
    res["two_in_flight_ms"] = min(pairs.values())
    res["pairs"] = pairs
    return res


def executed_flop(st, n_lights):
    """The flop weights of the frame figure F (45 per ray/triangle test, 40 per node step, 30 per shaded hit, 60 per
    hit and light, 25 per child ray) on the work a launch really executed (its own counters)."""
    return (45 * st.tri_tests + 40 * st.node_visits + 30 * st.shaded_hits
            + 60 * st.shaded_hits * n_lights + 25 * st.child_rays)


def config_record(pkg, name, w, h, msaa, fc, dev_index, frames=5):
    """One of the other BASELINE configs, render only: ms per frame from HIP events around the
    launch, rays from one counted frame, sparse pixels checked against the oracle's fixture."""
    import torch
    capi, host = pkg.capi, pkg.host
    hs = host.HostScene(os.path.join(ROOT, "scenes", name + ".obj"))
    gs = hs.to_gpu(dev_index)
    m, lights = host.read_camera_sav(os.path.join(ROOT, "scenes", "cameras", name + ".sav"))
    corners = host.camera_corners(m, w, h)
    p = capi.default_params(w, h, msaa)
    dev = torch.device("cuda", dev_index)
    rgb = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    depth = torch.zeros((h, w), dtype=torch.float32, device=dev)
    tri = torch.zeros((h, w), dtype=torch.int32, device=dev)
    st = gs.render_device(corners, lights, p, rgb.data_ptr(), depth.data_ptr(), tri.data_ptr(), 0, True)
    gs.render_device(corners, lights, p, rgb.data_ptr(), depth.data_ptr(), 0, 0, False)
    torch.cuda.synchronize()
    gs.set_timing(True)
    for _ in range(frames):
        gs.render_device(corners, lights, p, rgb.data_ptr(), depth.data_ptr(), 0, 0, False)
    torch.cuda.synchronize()
    kt = gs.collect_timing()
    gs.set_timing(False)
    ms = kt.frame_ms / max(kt.frames, 1)
    # the same launches with the regions handed out by index (what the first frame of a view gets)
    gs.set_region_order(False)
    gs.render_device(corners, lights, p, rgb.data_ptr(), depth.data_ptr(), 0, 0, False)
    torch.cuda.synchronize()
    gs.set_timing(True)
    for _ in range(frames):
        gs.render_device(corners, lights, p, rgb.data_ptr(), depth.data_ptr(), 0, 0, False)
    torch.cuda.synchronize()
    kt0 = gs.collect_timing()
    gs.set_timing(False)
    gs.set_region_order(True)
    rays = st.rays_closest + st.rays_shadow
    key = f"{name}_{w}x{h}_msaa{msaa}"
    info = gs.info()
    rec = {"ms_per_frame": round(ms, 3), "ms_per_frame_regions_by_index": round(kt0.frame_ms / max(kt0.frames, 1), 3),
           "mrays_s": round(rays / ms / 1e3, 1), "rays_per_frame": int(rays),
           "lights": len(lights), "launches_timed": kt.frames,
           "scene_create_ms": round(info["create_ms"], 2), "tree_steps_ms": round(info["tree_ms"], 2),
           "device_nodes": info["n_nodes"], "render_path": info["render_path"]}
    if fc.get(key):
        f_needed = fc[key].get("flops_needed", fc[key]["flops_frame"])
        rec["frac"] = round(f_needed / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)
        rec["frac_executed"] = round(executed_flop(st, len(lights)) / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)
    rec["checked"] = check_sparse(key, rgb.cpu().numpy(), depth.cpu().numpy(), tri.cpu().numpy())
    gs.close()
    return rec


def check_sparse(key, rgb, depth, tri=None):
    """20000 seeded pixels of the full-size frame against two committed fixtures: the CPU oracle's frame
    (tests/golden/sparse_<config>.npz) and the pixels the REAL reference's own performRayTracing composes to
    (sparse_reftex_<config>.npz for the two textured scenes, sparse_ref_<config>.npz for the others, which
    have no texture to load).  8-bit RGB within 1 LSB on >= 99.9 %, misses agree and depth within 1e-5 on
    >= 99.95 %, pixel-centre triangle equal on >= 99.99 %."""
    from oracle import oracle as O
    gold = os.path.join(ROOT, "tests", "golden")
    ref = os.path.join(gold, f"sparse_reftex_{key}.npz")
    if not os.path.exists(ref):
        ref = os.path.join(gold, f"sparse_ref_{key}.npz")
    paths = [q for q in (os.path.join(gold, f"sparse_{key}.npz"), ref) if os.path.exists(q)]
    if not paths:
        return None
    ok = True
    for path in paths:
        g = np.load(path)
        x, y = g["x"], g["y"]
        d = np.abs(O.to_bytes(rgb[y, x]).astype(int) - g["rgb8"].astype(int)).max(axis=1)
        ok = ok and (d <= 1).mean() >= 0.999
        miss = g["depth"] == np.float32(3.402823466e+38)
        got = depth[y, x]
        ok = ok and ((got == np.float32(3.402823466e+38)) == miss).mean() >= 0.9999
        both = ~miss & (got != np.float32(3.402823466e+38))
        ok = ok and (np.abs(got[both] - g["depth"][both]) <= 1e-5 * g["depth"][both]).mean() >= 0.9995
        if tri is not None:
            ok = ok and (tri[y, x] == g["tri"]).mean() >= 0.9999
    return bool(ok)


def check_blur(rgb, depth1, blurred, focus, depth_power, n=300):
    """n seeded pixels of the blurred frame against the oracle's kernel (gauseanMap of the twice
    normalised depth, 169 taps in the reference's order) on the frame's own colour plane."""
    from oracle import oracle as O
    hh, ww = depth1.shape
    d2 = O.normalize_depth(depth1, focus)
    rng = np.random.Generator(np.random.PCG64(1806))
    xs, ys = rng.integers(6, ww - 6, n), rng.integers(6, hh - 6, n)
    worst = 0.0
    for x, y in zip(xs, ys):
        m = O.gaussian_map(float(d2[y, x]), depth_power)
        acc = np.zeros(3, np.float32)
        win = rgb[y - 6:y + 7, x - 6:x + 7]
        for j in range(13):
            for k in range(13):
                acc = (acc + win[j, k] * m[j, k]).astype(np.float32)
        got = blurred[y - 6, x - 6]
        if np.isnan(acc).any() or np.isnan(got).any():
            if not (np.isnan(acc).all() and np.isnan(got).all()):
                return False
            continue
        worst = max(worst, float(np.abs(acc - got).max()))
    return worst < 2e-6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the per-config records and the 'r' path timing (profiling runs)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames rendered concurrently per GPU (each with its own workspace and stream); "
                         "default 1 on one GPU, 2 on several")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    # RT_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the ranks
    # share the visible GPUs and the tile payloads are gathered through host memory.  It checks
    # the partition / pack / gather / unpack path, not RCCL, and its timings mean nothing.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    pkg = importlib.import_module("raytracing-project_amd")
    capi, host = pkg.capi, pkg.host
    import ctypes as C

    hs = host.HostScene(os.path.join(ROOT, "scenes", SCENE + ".obj"))
    m, lights = host.read_camera_sav(os.path.join(ROOT, "scenes", "cameras", SCENE + ".sav"))
    corners = host.camera_corners(m, W, H)
    p = capi.default_params(W, H, MSAA)
    frame_params = capi.default_params(W, H, MSAA)   # the whole frame; p carries this rank's part of the deal
    p.tile_w = p.tile_h = TILE
    L = capi.lib()

    # Frames in flight.  A rank's part of the frame is one persistent launch whose last regions
    # run on a nearly idle GPU (about 1 ms, whatever the size of the part).  With several GPUs
    # that tail is a third of the launch, so consecutive frames are issued on two streams, each
    # with its own scene workspace and frame buffers: the blocks of frame i + 1 start as those of
    # frame i retire, and on rank 0 the post-process of frame i overlaps the render of i + 1.
    # On one GPU the frames run one after the other, so that the launch durations behind the
    # roofline figure are those of a launch that has the GPU to itself.
    # Whether two launches really overlap depends on the hardware queues the runtime gave their streams
    # (not visible to a program): with two frames in flight, four candidates are made and, once the
    # deal is settled, the partner of the first is the one whose launches are seen to overlap its own.
    # The line's value and ms_per_step come from a timed loop with ONE frame in flight, for every N, so that
    # the values the driver divides for its scaling curve are of one kind (the time of a frame); a second
    # timed loop of the same K steps with two frames in flight follows and is reported beside it
    # ("two_frames_in_flight": throughput of consecutive frames).  --frames-in-flight F runs the first loop
    # with F instead (experiments).
    n_flight = args.frames_in_flight or 1
    n_cand = 4
    slots = []
    for i in range(n_cand):
        st_ = torch.cuda.Stream(device=dev)
        slots.append({"gs": hs.to_gpu(local), "stream": st_,
                      "rgb": torch.zeros((H, W, 3), dtype=torch.float32, device=dev),
                      "depth": torch.zeros((H, W), dtype=torch.float32, device=dev),
                      "blurred": torch.zeros((H - 12, W - 12, 3), dtype=torch.float32, device=dev),
                      "ev": [torch.cuda.Event(enable_timing=True) for _ in range(2)]})
    gs = slots[0]["gs"]
    rgb, depth = slots[0]["rgb"], slots[0]["depth"]
    PER_RANK = 16   # shares per GPU in the deal; rank 0 may own fewer (it also runs the post-process)
    state = {"gather": "none (1 GPU)" if world == 1 else "torch.distributed.gather"}

    # The gather of the tile payloads goes through the library's own RCCL entry point
    # (rt_gather_tiles_rccl: grouped ncclRecv on rank 0 / ncclSend elsewhere, on the frame's
    # stream) with a communicator of its own; the unique id travels over the process group.
    # RT_BENCH_GATHER=torch keeps torch.distributed.gather.
    comm = C.c_void_p()
    if world > 1 and backend == "nccl" and os.environ.get("RT_BENCH_GATHER", "rccl") == "rccl":
        try:
            ident = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                buf = (C.c_ubyte * 128)()
                capi.check(L.rt_rccl_unique_id(buf))
                ident.copy_(torch.tensor(list(buf), dtype=torch.uint8))
            dist.broadcast(ident, src=0)
            raw = (C.c_ubyte * 128)(*ident.cpu().tolist())
            capi.check(L.rt_rccl_comm_create(raw, world, rank, local, C.byref(comm)))
            state["gather"] = "rt_gather_tiles_rccl"
        except Exception as e:  # noqa: BLE001 - reported in the line, the torch gather takes over
            state["gather"] = f"torch.distributed.gather (rt_rccl_comm_create failed: {e})"
            comm = C.c_void_p()

    def set_deal(first_rank_shares):
        """The library's deal (rt_deal_params): tiles go over first_rank_shares + 16 (N - 1) shares, rank 0
        renders the first first_rank_shares of them, every other rank 16."""
        per = []
        for r in range(world):
            pr = capi.RtParams()
            capi.check(L.rt_deal_params(C.byref(frame_params), world, first_rank_shares, r, C.byref(pr)))
            per.append(pr)
        p.tile_w, p.tile_h = per[rank].tile_w, per[rank].tile_h
        p.rank, p.world, p.rank_span = per[rank].rank, per[rank].world, per[rank].rank_span
        n_slots = L.rt_tile_pixels(C.byref(p), p.rank)
        state.update(deal=(per[0].world, [(q.rank, q.rank_span) for q in per]), first_rank_shares=first_rank_shares)
        per_rank = [int(L.rt_tile_pixels(C.byref(q), q.rank)) * 4 for q in per]
        state["floats_per_rank"] = (C.c_int64 * world)(*per_rank)
        state["offsets"] = [sum(per_rank[:r]) for r in range(world)]
        for sl in slots:
            capi.check(L.rt_scene_set_deal(sl["gs"].h, first_rank_shares))
            sl["payload"] = torch.zeros((max(n_slots, 1), 4), dtype=torch.float32, device=dev)[:n_slots]
            sl["gathered"] = torch.zeros(max(sum(per_rank), 4) if rank == 0 else 4, dtype=torch.float32,
                                         device=dev)

    set_deal(PER_RANK)

    flight = {"n": n_flight}

    def frame(i=0, stats=False):
        sl = slots[i % flight["n"]]
        with torch.cuda.stream(sl["stream"]):
            stream = sl["stream"].cuda_stream
            d_rgb, d_depth = C.c_void_p(sl["rgb"].data_ptr()), C.c_void_p(sl["depth"].data_ptr())

            def unpack(src, buf):
                ps = capi.RtParams.from_buffer_copy(p)      # the source rank's shares
                ps.rank, ps.rank_span = state["deal"][1][src]
                capi.check(L.rt_unpack_tiles_device(local, C.byref(ps), ps.rank, C.c_void_p(buf.data_ptr()),
                                                    d_rgb, d_depth, C.c_void_p(stream)))

            st = sl["gs"].render_device(corners, lights, p, sl["rgb"].data_ptr(), sl["depth"].data_ptr(), 0,
                                        stream, stats)
            if world > 1:
                payload = sl["payload"]
                capi.check(L.rt_pack_tiles_device(local, C.byref(p), d_rgb, d_depth,
                                                  C.c_void_p(payload.data_ptr()), C.c_void_p(stream)))
                if backend == "gloo":
                    pkg.distributed.gather_frame(payload.cpu(), W, H, TILE, TILE, rank, world, dist,
                                                 lambda src, buf: unpack(src, buf.to(dev)), state["deal"])
                elif state["gather"] == "rt_gather_tiles_rccl":
                    capi.check(L.rt_gather_tiles_rccl(comm, rank, world, C.c_void_p(payload.data_ptr()),
                                                      state["floats_per_rank"],
                                                      C.c_void_p(sl["gathered"].data_ptr()), C.c_void_p(stream)))
                    if rank == 0:
                        for src in range(world):
                            n_f = state["floats_per_rank"][src]
                            if n_f:
                                unpack(src, sl["gathered"][state["offsets"][src]:state["offsets"][src] + n_f])
                else:
                    pkg.distributed.gather_frame(payload, W, H, TILE, TILE, rank, world, dist, unpack,
                                                 state["deal"])
            if rank == 0:
                sl["ev"][0].record()
                # writeDepthBMP and writeBlurredBMP each normalise (imageWriter.h:352,416): the first
                # in place, the second as the blur reads the plane
                capi.check(L.rt_postprocess_device(local, W, H, p.focusDepth, p.depthPower, d_rgb, d_depth,
                                                   C.c_void_p(sl["blurred"].data_ptr()), C.c_void_p(stream)))
                sl["ev"][1].record()
        return st

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def bcast_int(v):
        t = torch.tensor([int(v)], dtype=torch.int64, device="cpu" if backend == "gloo" else dev)
        dist.broadcast(t, src=0)
        return int(t.item())

    def timed_frames(n):
        """(this rank's render ms, rank 0's post-process ms) averaged over n frames."""
        gs.set_timing(True)
        post = 0.0
        for _ in range(n):
            frame()
            torch.cuda.synchronize()
            if rank == 0:
                post += slots[0]["ev"][0].elapsed_time(slots[0]["ev"][1]) / n
        kt_ = gs.collect_timing()
        gs.set_timing(False)
        return kt_.frame_ms / max(kt_.frames, 1), post

    # Static load balance, settled before the timed region: rank 0 also unpacks the gathered
    # tiles and runs the depth passes and the blur, so it is dealt fewer tiles.  With a render
    # time of A * (fraction of the frame) + tail per rank, the frame is balanced when rank 0's
    # fraction is f0 = (1 - (N - 1) * t_post / A) / N; A comes from rank 0's render time under
    # two deals (equal shares, half shares).
    calib = None
    if world > 1 and os.environ.get("RT_BENCH_EVEN") != "1" and state["gather"] == "rt_gather_tiles_rccl":
        # the library's own calibration (rt_calibrate_deal, what rt_render --gpus N runs): a collective call
        got = C.c_int32(0)
        cam_ = np.ascontiguousarray(corners, np.float32)
        li_ = np.ascontiguousarray(lights, np.float32)
        capi.check(L.rt_calibrate_deal(gs.h, comm, rank, world, cam_.ctypes.data_as(C.c_void_p),
                                       li_.ctypes.data_as(C.c_void_p), len(li_), C.byref(frame_params),
                                       C.byref(got)))
        set_deal(int(got.value))
        calib = {"by": "rt_calibrate_deal", "first_rank_shares": int(got.value)}
    elif world > 1 and os.environ.get("RT_BENCH_EVEN") != "1":
        # without a communicator of the library's (rehearsal over gloo, torch's gather): the same rule here
        frame()                                        # first-touch allocations
        t_eq, t_post = timed_frames(2)
        f_eq = PER_RANK / state["deal"][0]
        set_deal(PER_RANK // 2)
        frame()
        t_half, _ = timed_frames(2)
        f_half = state["first_rank_shares"] / state["deal"][0]
        shares0 = PER_RANK
        if rank == 0:
            slope = (t_eq - t_half) / (f_eq - f_half)  # ms per unit fraction of the frame
            if slope > 0:
                f0 = (1.0 - (world - 1) * (t_post + 0.1) / slope) / world   # + 0.1 ms for the unpack kernels
                f0 = min(max(f0, 0.2 / world), 1.0 / world)
                shares0 = int(round(PER_RANK * (world - 1) * f0 / (1.0 - f0)))
            calib = {"by": "bench.py", "render_ms_equal": round(t_eq, 3), "render_ms_half": round(t_half, 3),
                     "post_ms": round(t_post, 3)}
        shares0 = bcast_int(min(max(shares0, 1), PER_RANK))
        set_deal(shares0)

    if state["gather"] == "rt_gather_tiles_rccl":
        # both gathers once, outside the timed region: the frames rank 0 rebuilds must be equal
        frame()
        torch.cuda.synchronize()
        mine = (slots[0]["rgb"].clone(), slots[0]["depth"].clone()) if rank == 0 else None
        state["gather"] = "torch.distributed.gather"
        frame()
        torch.cuda.synchronize()
        same = 1
        if rank == 0:
            same = int(torch.equal(mine[0], slots[0]["rgb"]) and torch.equal(mine[1], slots[0]["depth"]))
        same = bcast_int(same)
        state["gather"] = "rt_gather_tiles_rccl" if same else \
            "torch.distributed.gather (rt_gather_tiles_rccl rebuilt a different frame)"

    flight_probe = None
    if not args.no_extras:
        def part_ms(which, n=12):
            for rep_ in range(2):                       # the first pass warms up
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                for i in range(n):
                    sl = slots[which[i % len(which)]]
                    sl["gs"].render_device(corners, lights, p, sl["rgb"].data_ptr(), sl["depth"].data_ptr(), 0,
                                           sl["stream"].cuda_stream, False)
                torch.cuda.synchronize()
            return (time.perf_counter() - t_) * 1e3 / n
        alone = part_ms([0])
        pairs = {k: part_ms([0, k]) for k in range(1, n_cand)}
        best = min(pairs, key=pairs.get)
        flight_probe = {"render_alone_ms": round(alone, 3),
                        "with_partner_ms": {str(k): round(v, 3) for k, v in pairs.items()}, "partner": best}
        slots = [slots[0], slots[best]]                 # the others' buffers are released with them
        torch.cuda.empty_cache()
    else:
        slots = slots[:2]
        torch.cuda.empty_cache()

    # one counted frame: rays this rank traces per frame (deterministic)
    st = frame(0, stats=True)
    torch.cuda.synchronize()
    my_rays = st.rays_closest + st.rays_shadow
    my = torch.tensor([float(my_rays), float(st.tri_tests), float(st.node_visits)],
                      dtype=torch.float64, device=dev)
    my_executed_flop = executed_flop(st, len(lights))      # this rank's launch
    if world > 1:
        if backend == "gloo":
            my_c = my.cpu()
            dist.all_reduce(my_c)
            my = my_c
        else:
            dist.all_reduce(my)
    rays_per_frame, tests_per_frame, nodes_per_frame = (float(x) for x in my.tolist())

    def timed_loop(n_in_flight):
        """W untimed steps, then exactly K steps between barriers: seconds (max over ranks) and the HIP-event
        times of this rank's launches inside the timed region."""
        flight["n"] = n_in_flight
        for i in range(args.warmup):
            frame(i)
        for sl in slots:
            sl["gs"].set_timing(True)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            frame(i)
        barrier()
        dt = time.perf_counter() - t0
        kt_ = capi.RtKernelTimes()
        for sl in slots:
            k_ = sl["gs"].collect_timing()
            kt_.frame_ms += k_.frame_ms
            kt_.frames += k_.frames
            sl["gs"].set_timing(False)
        el = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            if backend == "gloo":
                el = el.cpu()
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), kt_

    elapsed, kt = timed_loop(n_flight)
    other_flight = 2 if n_flight == 1 else 1
    if args.no_extras:      # profiling runs: the timed loop alone, so that the launches a profiler sees are its launches
        elapsed_other, kt_other = elapsed, kt
    else:
        elapsed_other, kt_other = timed_loop(other_flight)
    flight["n"] = n_flight
    # The library starts the regions a launch would otherwise wait for first, from the durations the frame
    # before measured (rt_scene_set_region_order).  The timed loop above runs that way - it is what a caller
    # gets from the second frame of a view on.  The same K steps again with the regions by index: what the
    # first frame of a view costs.
    by_index = None
    if not args.no_extras:
        for sl in slots:
            sl["gs"].set_region_order(False)
        el_i, kt_i = timed_loop(n_flight)
        for sl in slots:
            sl["gs"].set_region_order(True)
        frame(0)
        frame(1)
        torch.cuda.synchronize()
        by_index = {"ms_per_step": round(el_i / args.steps * 1e3, 3),
                    "k_wf_pipeline_ms": round(kt_i.frame_ms / max(kt_i.frames, 1), 3),
                    "what": "the same K steps with every launch handing its regions out by index (no durations from a "
                            "frame before): the first frame of a view; the line's value and ms_per_step are frames "
                            "that follow a frame of the same view"}
    # the second slot holds a whole frame of its own once a loop has given it one (bit patterns: the blurred
    # plane holds the reference's NaN pixel)
    slot1_used = (not args.no_extras or n_flight > 1) and (args.steps >= 2 or args.warmup >= 2)
    two_equal_one = True
    if rank == 0 and slot1_used:
        two_equal_one = all(bool(torch.equal(slots[0][k_].view(torch.int32), slots[1][k_].view(torch.int32)))
                            for k_ in ("rgb", "depth", "blurred"))

    if rank == 0 and world > 1 and os.environ.get("RT_BENCH_CHECK") == "1":
        # rehearsal check: the gathered frame equals the frame one GPU renders alone
        p1 = capi.default_params(W, H, MSAA)
        rgb1 = torch.zeros_like(rgb)
        depth1 = torch.zeros_like(depth)
        gs.render_device(corners, lights, p1, rgb1.data_ptr(), depth1.data_ptr(), 0, 0, False)
        torch.cuda.synchronize()
        same = all(bool((rgb1 == sl["rgb"]).all().item()) for sl in slots[:2 if slot1_used else 1])
        print(f"partition check: gathered frame {'equals' if same else 'DIFFERS FROM'} the single-GPU frame",
              file=sys.stderr, flush=True)
        if not same:
            raise SystemExit(2)
    checked = None
    if rank == 0:
        # outside the timed region: the frame the loop produced last, against the oracle's fixtures
        torch.cuda.synchronize()
        last = slots[0]      # (the second loop's frames alternate between the two slots: compared above)
        h_rgb, h_depth1 = last["rgb"].cpu().numpy(), last["depth"].cpu().numpy()
        raw = torch.zeros_like(last["depth"])               # the depth plane before normalisation
        p1 = capi.default_params(W, H, MSAA)
        tmp_rgb = torch.zeros_like(last["rgb"])
        gs.render_device(corners, lights, p1, tmp_rgb.data_ptr(), raw.data_ptr(), 0, 0, False)
        torch.cuda.synchronize()
        same_rgb = bool((tmp_rgb == last["rgb"]).all().item())
        ok_sparse = check_sparse(f"{SCENE}_{W}x{H}_msaa{MSAA}", h_rgb, raw.cpu().numpy())
        from oracle import oracle as O
        ok_depth = bool(np.array_equal(O.normalize_depth(raw.cpu().numpy(), p.focusDepth), h_depth1))
        ok_blur = check_blur(h_rgb, h_depth1, last["blurred"].cpu().numpy(), p.focusDepth, p.depthPower)
        checked = bool(same_rgb and ok_sparse and ok_depth and ok_blur and two_equal_one)
        if not checked:
            print(f"bench check FAILED: frame equals a fresh render {same_rgb}, sparse pixels {ok_sparse}, "
                  f"depth normalisation {ok_depth}, blur {ok_blur}, both slots hold the same frame {two_equal_one}",
                  file=sys.stderr, flush=True)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = rays_per_frame * args.steps / elapsed / 1e6
        out = {
            "metric": "Mrays/s (all bounces), final.obj 3840x2160 MSAA4 + DoF; ms/frame in ms_per_step",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "checked": checked,
            # a line from one GPU says nothing about scaling; no multi-GPU node has run this bench yet
            "scaling_measured": world > 1,
            "frames_in_flight": n_flight,
            ("two_frames_in_flight" if other_flight == 2 else "one_frame_in_flight"): None if args.no_extras else {
                "ms_per_step": round(elapsed_other / args.steps * 1e3, 3),
                "value": round(rays_per_frame * args.steps / elapsed_other / 1e6, 2),
                "k_wf_pipeline_ms": round(kt_other.frame_ms / max(kt_other.frames, 1), 3),
                "what": "the same K steps timed again with %d frame%s in flight per GPU (streams, workspaces and "
                        "frame buffers of their own): %s; the line's value and ms_per_step are the loop with %d"
                        % (other_flight, "s" if other_flight > 1 else "",
                           "throughput of consecutive frames, the launch of frame i + 1 fills the GPU as the last "
                           "regions of frame i run out" if other_flight == 2 else "the time of one frame", n_flight)},
            "region_order": "every launch of the timed loop starts the regions it would otherwise wait for first, from the "
                            "durations the frame before left in the scene (rt_scene_set_region_order, the library's default); "
                            "nothing of a frame's result is carried over - every ray of every frame is traced",
            "regions_by_index": by_index,
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "data": "scenes/final.obj + scenes/cameras/final.sav (the reference's own scene; no "
                    "published camera exists, this one is fixed in the repo)",
            "config": {"workload": f"{SCENE}.obj {W}x{H} MSAA{MSAA} + DoF, {len(lights)} lights, "
                                   "reflection/refraction depth 5",
                       "partition": (f"{TILE}x{TILE} tiles dealt over {state['deal'][0]} shares: rank 0 renders "
                                     f"{state['first_rank_shares']}, every other rank {PER_RANK} (rank 0 also runs the "
                                     "post-process); gather to rank 0: " + state["gather"]) if world > 1
                       else f"{TILE}x{TILE} tiles, 1 GPU",
                       "rays_per_frame": int(rays_per_frame),
                       "tri_tests_per_ray": round(tests_per_frame / rays_per_frame, 2),
                       "nodes_per_ray": round(nodes_per_frame / rays_per_frame, 2)},
        }
        # roofline of the dominant kernel: k_wf_pipeline, the single persistent launch that renders
        # this rank's part of the frame (camera rays, KD traversal + ray/triangle tests, shading,
        # pixel resolve).  Its algorithmic work is F = 45 T + 40 S + 30 H + 60 Hl + 25 C with the
        # counters of the REFERENCE's own algorithm for this frame (counted by the CPU oracle,
        # fixture tests/golden/frame_counters.json); its duration comes from HIP events recorded
        # on the launch stream around every launch inside the timed loop (rank 0).
        fc_path = os.path.join(ROOT, "tests", "golden", "frame_counters.json")
        key = f"{SCENE}_{W}x{H}_msaa{MSAA}"
        roof = None
        if os.path.exists(fc_path) and kt.frames > 0:
            fc = json.load(open(fc_path)).get(key)
            if fc:
                # flops_frame is the reference's walk as it stands.  Part of it never reaches the
                # image: the reference shades every pixel-centre ray and discards the colour
                # (main.cpp:439-445), and it traces shadow queries whose answer cannot change
                # shadowMax (raytracing.cpp:127-143).  flops_needed leaves both out; the headline
                # figure uses it, the figure for the full walk is reported next to it.
                f_needed = fc.get("flops_needed", fc["flops_frame"])
                # rank 0's launch covers its own fraction of the frame (the HIP events are rank 0's)
                frac0 = state["first_rank_shares"] / state["deal"][0] if world > 1 else 1.0
                f_kernel = f_needed * frac0
                kern_ms = kt.frame_ms / kt.frames
                ach = f_kernel / (kern_ms * 1e-3) / 1e12
                traffic = None
                tp = os.path.join(ROOT, "profiles", "traffic.json")
                if os.path.exists(tp):
                    traffic = json.load(open(tp)).get("k_wf_pipeline_hbm_bytes_per_launch")
                hbm_bytes = W * H * 16 + W * H * 16 + (W - 12) * (H - 12) * 12
                exe = my_executed_flop
                roof = {"bound": "mfma", "bound_is_vector_fp32": True, "bound_detail": "fp32 vector ALU; the path has no matrix-shaped work, "
                        "so the ceiling is the FP32 rate, which is the same 157.3 TFLOP/s for vector FMA "
                        "and dense f32 MFMA", "kernel": "k_wf_pipeline", "achieved": round(ach, 3),
                        "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_PEAK_TFLOPS, 4),
                        "traffic": traffic,
                        "traffic_is": "bytes between the L2s and the memory side of the fabric per launch (FETCH_SIZE doubled "
                                      "per the gfx950 correction + WRITE_SIZE, profiles/traffic.json): Infinity-Cache hits are "
                                      "counted in it, no counter of this stack separates them - an upper bound of the HBM bytes",
                        "kernel_ms": round(kern_ms, 3),
                        "launches_per_frame": round(kt.frames / args.steps, 2),
                        "frames_in_flight": n_flight,
                        "algorithmic_gflop_per_launch": round(f_kernel / 1e9, 2),
                        "frac_reference_walk": round(
                            fc["flops_frame"] * frac0 / (kern_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                        "frac_reference_walk_note": "the reference's whole walk (F = 45 T + 40 S + 30 H + 60 Hl + 25 C over its counters, 105 tests per ray "
                        "in its tree) priced at this launch's duration: a yardstick of the frame against the "
                        "reference's algorithm, above 1 once the launch is faster than a peak-rate machine running "
                        "that algorithm; the launch itself executes the work under 'executed'",
                        "executed": {"gflop_per_launch": round(exe / 1e9, 2),
                                     "achieved": round(exe / (kern_ms * 1e-3) / 1e12, 3),
                                     "frac": round(exe / (kern_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                                     "what": "45 x tests + 40 x node steps + 30 x shaded hits + 60 x hit-light "
                                             "pairs + 25 x child rays as this launch executed them (its own "
                                             "counters): the hardware utilisation; frac above is the frame "
                                             "measured against the reference's walk"},
                        "note": "no MFMA on this path: the bound is the FP32 vector peak (= the f32 MFMA "
                                "peak, 157.3 TFLOP/s); algorithmic flop follow the reference's own walk "
                                "(T = %d ray/triangle tests, S = %d leaf steps per frame)"
                                % (fc["frame"]["tri_tests"], fc["frame"]["leaf_steps"])
                                + ("" if n_flight == 1 else "; with %d frames in flight a launch shares the GPU with "
                                   "the next one, its HIP-event duration is longer than its share of the GPU and "
                                   "this figure understates the kernel" % n_flight),
                        "frame": {"ms_incl_dof_and_gather": round(ms, 3),
                                  "achieved_tflops": round(f_needed / (ms * 1e-3) / 1e12, 3),
                                  "frac": round(f_needed / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS / world, 4),
                                  "reference_rays": fc["frame"]["rays_closest"] + fc["frame"]["rays_shadow"]},
                        "hbm": {"algorithmic_bytes_per_frame": hbm_bytes,
                                "achieved_gbs": round(hbm_bytes / (ms * 1e-3) / 1e9, 2),
                                "peak_gbs": HBM_PEAK_GBS,
                                "frac": round(hbm_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
        if calib:
            out["config"]["load_balance_calibration"] = calib
        out["roofline"] = roof
        out["kernel_ms"] = {"k_wf_pipeline": round(kt.frame_ms / max(kt.frames, 1), 3)}
        if n_flight == 1:
            out["kernel_ms"]["dof_and_gather"] = round(ms - kt.frame_ms / max(kt.frames, 1), 3)
        out["config"]["frames_in_flight"] = n_flight
        if flight_probe:
            out["config"]["frames_in_flight_probe"] = flight_probe
        info = gs.info()
        out["scene_create_ms"] = {"rt_scene_create": round(info["create_ms"], 2),
                                  "tree_hand_down_and_refine": round(info["tree_ms"], 2),
                                  "on_gpu": bool(info["tree_on_gpu"]), "device_nodes": info["n_nodes"]}
        if world == 1 and not args.no_extras:
            fc_all = json.load(open(fc_path)) if os.path.exists(fc_path) else {}
            out["configs"] = {f"{n_}_{w_}x{h_}_msaa{m_}": config_record(pkg, n_, w_, h_, m_, fc_all, local)
                              for n_, w_, h_, m_ in OTHER_CONFIGS}
            out["r_path_ms"] = r_path_ms(pkg, local)
            out["frame_parts_in_flight"] = frame_parts_in_flight(pkg, hs, corners, lights, local)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(corners, lights)
        print(json.dumps(out), flush=True)
        if checked is False or any(c.get("checked") is False for c in out.get("configs", {}).values()):
            raise SystemExit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
