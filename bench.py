#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path: Mpaths/s and time-to-image.

  python bench.py --gpus N --steps K --warmup W [--config cfg1|cfg2|cfg3|cfg4|cfg5]
  N > 1 without a launcher (no WORLD_SIZE in the environment): this script starts `python -m torch.distributed.run
  --nproc-per-node N` on itself, one rank per GPU, before anything touches the GPU, and passes the ranks' exit code on;
  under an external `torch.distributed.run` it runs as the rank it is given.

Default workload = cfg4, the configuration the metric is quoted on: the final battle scene of main()
(raytracer.cpp:1222-1435) at 1024 x 1024. The other configurations of BASELINE.json can be selected for
profiling; they share every line of this script.

A step = every rank renders --spp-step samples of every pixel of the frame (1 path = 1 camera sample = one
Scene::getColor(r,0,false) call, raytracer.cpp:1485). Ranks own disjoint global sample indices (spp sharding;
weak scaling: per-rank work per step is fixed). Inside the timed region, after the K steps, the FP64 radiance
accumulators are summed onto rank 0 (RCCL reduce over xGMI) and rank 0 resolves the 8-bit image, so `value`
covers everything between "scene resident in HBM" and "image in host memory". Nothing waits between the steps:
the kernels' durations are read from per-call HIP events after the region.

After the timed region the same process measures time-to-image, the second half of the metric, as a separate
strong-scaling render: the configuration's full sample count (1000 for cfg4) split over the ranks, then reduce and
resolve; reported with the set-up (scene construction, BVH build, upload) that precedes any render.

The OBJ/JPG assets of the reference are not redistributable and are absent on the GPU box: procedural stand-ins of
the same triangle counts, object-space extents and transform chains are used (ray-tracing_amd/scenes.py).
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_VECTOR_TFLOPS = 78.6   # MI355X vector FP64 with every instruction an FMA = half the 157.3 TFLOP/s FP32 vector peak
PEAK_FP32_VECTOR_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
FETCH_GATHER_FACTOR = 1.39       # bytes fetched per byte FETCH_SIZE reports, for gathers of 128-byte records (tools/calib_fetch.hip)

# name -> (scene builder name, builder kwargs, frame size, samples of a step, samples of the finished image, CPU baseline samples, description)
CONFIGS = {
    "cfg1": ("sphere_room_deterministic", {}, 512, 1, 1, 1, "cfg1 sphere room, point light, hard shadows, mirror + refraction, no jitter"),
    "cfg2": ("sphere_room", {}, 512, 100, 100, 100, "cfg2 sphere room, area light, indirect, random Fresnel, antialiasing"),
    "cfg3": ("mesh_room", {"n_tris": 54320, "tex_size": 1024}, 512, 100, 100, 24, "cfg3 textured mesh (54 320-triangle stand-in for the dog) on a mirror floor"),
    "cfg4": ("battle", {}, 1024, 64, 1000, 12, "cfg4 battle scene (6 spheres + 12 procedural stand-in meshes, textures, DoF sigma 0.01)"),
    "cfg5": ("stress", {}, 2048, 64, 256, 2, "cfg5 synthetic stress (5 M-triangle mesh + 1000 spheres)"),
}


def algorithmic_flops(c, n_spheres):
    """Flops of the algorithm as the reference states it, from a counted run with the reference's traversal (add/sub/mul/div/
    sqrt/compare = 1; cos/sin/log are counted separately as transcendentals): per ray n_spheres*20 + 25 for the sphere that is
    hit, 24 per box test, 54 per triangle test; per path 60 for the camera sample; 150 per diffuse vertex, 40 per specular
    vertex."""
    sphere_work = c["rays"] * n_spheres * 20  # every ray tests every sphere (raytracer.cpp:934-938)
    flops = (sphere_work + c["rays"] * 25 + c["box_tests"] * 24 + c["tris"] * 54 + c["paths"] * 60 +
             c["diffuse"] * 150 + c["specular"] * 40)
    transc = c["paths"] * 8 + c["diffuse"] * 4
    return flops, transc


def production_flops(c):
    """(FP64 flops, FP32 flops) of a counted run of the production traversal, by the same rule. Single precision: every box
    test of the filter tree, of the tree over a long sphere list, of the mesh roots and of the second pass's records (24
    each). Double precision: exact sphere tests (20) + 25 per ray for the sphere that is hit, triangle tests including the
    second evaluation of a confirmed winner (54), box tests in the reference's double-precision form - undecided
    single-precision tests and the confirmation of a filter-tree winner - (24), camera sample (60), diffuse (150) and
    specular (40) vertices."""
    f32 = (c["box_tests"] + c["sphere_nodes"]) * 24
    f64 = (c["sphere_tests"] * 20 + c["rays"] * 25 + c["tris"] * 54 + c["box_exact"] * 24 + c["paths"] * 60 +
           c["diffuse"] * 150 + c["specular"] * 40)
    return f64, f32


def algorithmic_bytes(c, node_bytes):
    """node_bytes per BVH record visited, 80 B per triangle tested, 128 B of attributes per mesh hit (upper bound: one
    per mesh call), 24 B accumulate per path"""
    return c["nodes"] * node_bytes + c["tris"] * 80 + c["mesh_calls"] * 128 + c["paths"] * 24


def lib_sha16(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(n, argv, run=None, port=None):
    """Start one process per GPU on this script (the parallelism a caller of the reference gets from its one OpenMP line,
    raytracer.cpp:1456) and return their exit code. Called before torch is imported: this process never touches the GPU."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)
    return (run or subprocess.run)(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS))
    ap.add_argument("--spp-step", type=int, default=0, help="samples per pixel per rank per step (default: the configuration's; one kernel launch per 64)")
    ap.add_argument("--size", type=int, default=0, help="frame size (default: the configuration's)")
    ap.add_argument("--total-spp", type=int, default=0, help="samples of the time-to-image render, split over the ranks (default: the configuration's)")
    ap.add_argument("--no-time-to-image", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the extra leg that times the single-precision render (RT_F_FAST)")
    ap.add_argument("--cpu-spp", type=int, default=0, help="samples per pixel of the CPU baseline's bounded sample (default: the configuration's, 10-20 s of CPU work)")
    ap.add_argument("--out-png", default="")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import ctypes
    import torch
    import torch.distributed as dist

    # RT_BENCH_BACKEND=gloo is for rehearsing the multi-rank path on a box with fewer GPUs than ranks (ranks then share
    # devices, which RCCL refuses); the measured configuration is one GPU per rank over RCCL.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend != "nccl":
        local_rank = local_rank % max(1, n_dev)
    elif world > n_dev:
        if rank == 0:
            print("bench.py: --gpus %d but this node has %d GPU(s) (one rank per GPU over RCCL; RT_BENCH_BACKEND=gloo lets ranks "
                  "share a GPU for a rehearsal)" % (world, n_dev), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    rta = importlib.import_module("ray-tracing_amd")
    scenes = importlib.import_module("ray-tracing_amd.scenes")
    parallel = importlib.import_module("ray-tracing_amd.parallel")

    builder, kw, size, spp_step, spp_image, cpu_spp, what = CONFIGS[args.config]
    W = H = args.size or size
    spp_step = args.spp_step or spp_step
    spp_image = args.total_spp or spp_image
    cpu_spp = args.cpu_spp or cpu_spp

    # ---- set-up: what precedes any render (the reference's own 12 s @ 1 spp, README.md:129, includes it) ----
    # The context (HIP runtime, code object, stream, scratch) does not depend on the scene: it is created by a second
    # thread while this one builds the scene, as a host program would do it.
    import threading
    made = {}

    def make_context():
        ta = time.perf_counter()
        try:
            made["r"] = rta.Renderer(local_rank)
        except Exception as e:  # reported by the main thread
            made["error"] = e
        made["s"] = time.perf_counter() - ta

    t0 = time.perf_counter()
    th = threading.Thread(target=make_context)
    th.start()
    out = getattr(scenes, builder)(W=W, **kw)   # scene construction: procedural meshes, transforms, buildBVH per mesh
    scene, cam, flags = out[0], out[1], out[2]
    t1 = time.perf_counter()
    th.join()
    if "error" in made:
        raise made["error"]
    r = made["r"]
    t2 = time.perf_counter()
    r.upload(scene)                              # flatten + derived trees + copy to HBM
    t3 = time.perf_counter()
    setup = {"scene_build_s": round(t1 - t0, 3), "context_s": round(made["s"], 3), "context_overlapped": True, "upload_s": round(t3 - t2, 3),
             "total_s": round(t3 - t0, 3)}
    info = scene.flat_info()

    accum = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    acc_ptr = accum.data_ptr()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(index):
        # global sample indices of this rank in this step: disjoint over (step, rank)
        off = parallel.shard_offset(index, world, rank, spp_step)
        r.render(cam, rta.make_params(W, H, spp_step, spp_offset=off, flags=flags), acc_ptr)

    for w in range(args.warmup):
        step(w)
    r.sync()
    if rank == 0 and args.warmup > 0:
        r.resolve(W, H, spp_step, acc_ptr)  # the tone-map kernel and its device buffer are part of the warm-up too
    if world > 1 and args.warmup > 0:
        # the timed region ends with one sum-reduction of the accumulator: let RCCL set up its connections for
        # that collective and size outside it, as the warm-up steps do for the kernel
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    accum.zero_()
    r.render_time_ms(reset=True)
    barrier()

    t_start = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    r.sync()
    if world > 1:
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()  # the resolve below runs on the renderer's own stream: the sum must have landed
    image = None
    if rank == 0:
        image = r.resolve(W, H, spp_step * args.steps * world, acc_ptr)
    barrier()
    elapsed = time.perf_counter() - t_start
    konly = ctypes.c_double(0.0)
    rta.lib.rt_debug_kernel_time_ms(r._h, ctypes.byref(konly), None)  # HIP events around the render kernel of every launch alone
    call_ms, kernel_launches = r.render_time_ms(reset=True)  # the same around every call: camera-ray kernel, render kernel, addition of the unit sums
    kernel_ms = konly.value

    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    paths_total = float(W) * H * spp_step * args.steps * world
    value = paths_total / elapsed / 1e6

    # ---- time-to-image: the configuration's full sample count, split over the ranks (strong scaling) ----
    tti = None
    if not args.no_time_to_image:
        lo, hi = parallel.split_samples(spp_image, rank, world)
        accum.zero_()
        barrier()
        ta = time.perf_counter()
        if hi > lo:
            r.render(cam, rta.make_params(W, H, hi - lo, spp_offset=lo, flags=flags), acc_ptr)
        r.sync()
        if world > 1:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        if rank == 0:
            image = r.resolve(W, H, spp_image, acc_ptr)
        barrier()
        tb = time.perf_counter() - ta
        tm = torch.tensor([tb, setup["total_s"]], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tb, su = float(tm[0].item()), float(tm[1].item())
        png_s = None
        if rank == 0 and args.out_png:
            tp = time.perf_counter()
            rta.write_png(args.out_png, image)
            png_s = round(time.perf_counter() - tp, 3)
        tti = {"spp": spp_image, "render_reduce_resolve_s": round(tb, 4), "setup_s": round(su, 3), "total_s": round(tb + su, 3),
               "Mpaths_per_s": round(float(W) * H * spp_image / tb / 1e6, 2), "png_write_s": png_s,
               "note": "set-up = scene construction (procedural stand-in meshes, transforms, buildBVH per mesh) with the context created beside it, then flatten + rt_upload_scene, max over ranks; "
                       "render = %d samples per pixel split over %d rank(s), reduce to rank 0, tone map, image in host memory" % (spp_image, world)}

    # ---- extra leg, never `value`: the single-precision render (RT_F_FAST) of the same steps, reduce and resolve ----
    fast = None
    if not args.no_fast_mode and flags == 0 and info["n_spheres"] <= 64:
        fsteps = max(1, min(args.steps, 3))
        r.render(cam, rta.make_params(W, H, spp_step, spp_offset=0, flags=rta.RT_F_FAST), acc_ptr)  # warm-up: its triangle copy is made on first use
        r.sync()
        accum.zero_()
        r.render_time_ms(reset=True)
        barrier()
        ta = time.perf_counter()
        for k in range(fsteps):
            r.render(cam, rta.make_params(W, H, spp_step, spp_offset=parallel.shard_offset(k, world, rank, spp_step), flags=rta.RT_F_FAST), acc_ptr)
        r.sync()
        if world > 1:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        if rank == 0:
            r.resolve(W, H, spp_step * fsteps * world, acc_ptr)
        barrier()
        tf = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tf, op=dist.ReduceOp.MAX)
        fk_ms, fk_n = r.render_time_ms(reset=True)
        fast = {"Mpaths_per_s": round(float(W) * H * spp_step * fsteps * world / float(tf.item()) / 1e6, 3), "steps": fsteps,
                "ms_per_step": round(float(tf.item()) / fsteps * 1e3, 3), "avg_launch_ms": round(fk_ms / max(1, fk_n), 3), "dtype": "f32 (sphere tests and accumulator f64)",
                "note": "rt_params.flags = RT_F_FAST: the same estimator, random stream and traversal in single precision with FMA contraction; no bit-for-bit "
                        "promise, held to the converged-image bars (tests/test_gpu_fast.py); an extra figure, not the metric"}

    if rank == 0:
        # Counted runs (separate, untimed). "production": the counted variant of the render kernel walking exactly what the
        # timed kernel walks (filter tree with single-precision box tests, any-hit shadow rays, rays without effect left out,
        # second pass on the reference's trees for unconfirmed rays) - the numerator of roofline.achieved / frac.
        # "reference": the algorithm as the reference states it (every node whose box the ray's line crosses,
        # RT_F_NO_PRUNE, on the reference's trees; equal to the CPU checker's counters).
        cnt_spp = 2
        scratch = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
        counted = {}
        for label, extra, own in (("reference", rta.RT_F_NO_PRUNE, 0), ("production", 0, 1)):
            rta.lib.rt_debug_count_own(r._h, own)
            r.reset_counters()
            r.render(cam, rta.make_params(W, H, cnt_spp, spp_offset=0, flags=flags | rta.RT_F_COUNTERS | extra), scratch.data_ptr())
            r.sync()
            counted[label] = r.counters()
            dbg = (ctypes.c_uint64 * 24)()
            rta.lib.rt_debug_stats(r._h, dbg)
            counted[label].update(sphere_tests=int(dbg[20]), sphere_nodes=int(dbg[21]), box_exact=int(dbg[22]), redo_rays=int(dbg[18]),
                                  node_steps=int(dbg[6]), node_lanes=int(dbg[7]), leaf_steps=int(dbg[8]), leaf_lanes=int(dbg[9]), queries=int(dbg[10]))
        rta.lib.rt_debug_count_own(r._h, 0)
        c, g = counted["reference"], counted["production"]
        paths_per_launch = float(W) * H * spp_step * args.steps / max(1, kernel_launches)
        avg_launch_s = kernel_ms / 1e3 / max(1, kernel_launches)
        f64, f32 = production_flops(g)
        f64, f32 = f64 / g["paths"], f32 / g["paths"]
        # bytes the production traversal asks for: 128 B per filter-tree record, 80 B per triangle tested, per shaded vertex the
        # winner's triangle, leaf index, leaf box and attribute record (80 + 4 + 24 + 128), 24 B accumulate per path
        own_bytes = (g["nodes"] * 128 + g["tris"] * 80 + (g["diffuse"] + g["specular"]) * 236 + g["paths"] * 24) / g["paths"]
        ref_flops, ref_transc = algorithmic_flops(c, info["n_spheres"])
        ref_flops /= c["paths"]
        ref_bytes = algorithmic_bytes(c, node_bytes=128) / c["paths"]      # two FP64 boxes + links, as the reference holds them
        scale = paths_per_launch / avg_launch_s / 1e12
        own_tflops = (f64 + f32) * scale
        t_floor = f64 * scale / PEAK_FP64_VECTOR_TFLOPS + f32 * scale / PEAK_FP32_VECTOR_TFLOPS  # = time at peak / time taken
        blended_peak = own_tflops / t_floor if t_floor > 0 else PEAK_FP64_VECTOR_TFLOPS
        ref_tflops = ref_flops * scale
        own_gbs = own_bytes * paths_per_launch / avg_launch_s / 1e9
        ahead = flags == 0 and spp_step * W * H * 40 <= (4 << 30)  # the launch's camera rays are made ahead of it (rt_capi.hip: kRtCamRayMaxBytes)
        if info["n_spheres"] > 64:
            kname = "rt_render_spheres_kernel<false, %s>" % ("true" if ahead else "false")
        else:
            kname = "rt_render_kernel<false, true, true>" if ahead else ("rt_render_kernel<false, true, false>" if flags == 0 else "rt_render_kernel<false, false, false>")
        roofline = {
            "bound": "valu", "achieved": round(own_tflops, 4), "peak": round(blended_peak, 2), "unit": "TFLOP/s",
            "frac": round(t_floor, 5),
            "flops_fp64": round(f64 * paths_per_launch, 1), "flops_fp32": round(f32 * paths_per_launch, 1),
            "peak_fp64": PEAK_FP64_VECTOR_TFLOPS, "peak_fp32": PEAK_FP32_VECTOR_TFLOPS,
            "frac_fp64_part": round(f64 * scale / PEAK_FP64_VECTOR_TFLOPS, 5), "frac_fp32_part": round(f32 * scale / PEAK_FP32_VECTOR_TFLOPS, 5),
            "traffic": None, "kernel": kname, "avg_launch_ms": round(avg_launch_s * 1e3, 3), "launches": kernel_launches,
            "avg_call_ms": round(call_ms / max(1, kernel_launches), 3),
            "call_note": "avg_launch_ms: the render kernel alone; avg_call_ms: with the kernels around it in the same launch - rt_camera_rays_kernel (the launch's "
                         "camera rays made ahead, all lanes busy; ~60 of the FP64 flops per path, counted in the numerator) and rt_units_add_kernel",
            "paths_per_launch": paths_per_launch,
            "definition": "achieved = (flops_fp64 + flops_fp32) per launch / mean launch duration (HIP events on the render stream over the timed "
                          "region); flops = what the production traversal executes, counted by the counted variant of the same kernel walking "
                          "the same trees the timed kernel walks (filter tree in single precision, reference trees for unconfirmed rays); "
                          "frac = (flops_fp64 / peak_fp64 + flops_fp32 / peak_fp32) / launch duration; peak = achieved / frac, the peak "
                          "blended by this mix. No FMA contraction in the FP64 code (-ffp-contract=off): hardly any counted flop is half of an FMA, "
                          "so an instruction-issue ceiling for this mix is about half of these peaks",
            "production_work": {"flops_fp64_per_path": round(f64, 1), "flops_fp32_per_path": round(f32, 1), "bytes_per_path": round(own_bytes, 1),
                                "records_per_path": round(g["nodes"] / g["paths"], 2), "box_tests_fp32_per_path": round((g["box_tests"] + g["sphere_nodes"]) / g["paths"], 2),
                                "box_tests_fp64_per_path": round(g["box_exact"] / g["paths"], 3),
                                "tris_per_path": round(g["tris"] / g["paths"], 2), "sphere_tests_per_ray": round(g["sphere_tests"] / max(1, g["rays"]), 2),
                                "second_pass_rays_per_path": round(g["redo_rays"] / g["paths"], 5),
                                "node_steps_per_query": round(g["node_steps"] / max(1, g["queries"]), 2), "node_step_fill": round(g["node_lanes"] / max(1, g["node_steps"]) / 64, 3),
                                "triangle_steps_per_query": round(g["leaf_steps"] / max(1, g["queries"]), 2),
                                "queries_per_path": round(g["queries"] * 64 / g["paths"], 3),
                                "requested_GBs": round(own_gbs, 2),
                                "note": "bytes the traversal asks for (records, triangles, attributes), served mostly by L1/L2: compare `traffic` (HBM side) where a counter pass exists; "
                                        "queries_per_path = wave queries x 64 lanes / paths"},
            "reference_semantics": {"flops_per_path": round(ref_flops, 1), "transcendentals_per_path": round(ref_transc / c["paths"], 2),
                                    "bytes_per_path": round(ref_bytes, 1), "nodes_per_path": round(c["nodes"] / c["paths"], 2),
                                    "tris_per_path": round(c["tris"] / c["paths"], 2), "TFLOPs_equivalent": round(ref_tflops, 4),
                                    "frac_equivalent": round(ref_tflops / PEAK_FP64_VECTOR_TFLOPS, 5),
                                    "note": "the unpruned reference algorithm's flops (all FP64 there) divided by this kernel's time: a speed-up-inclusive figure, not a utilisation"},
            "rays_per_path": round(c["rays"] / c["paths"], 3), "rays_traced_per_path": round(g["rays"] / g["paths"], 3),
            "rays_note": "rays_per_path = rays the reference traces; rays_traced_per_path = rays this kernel traces: the shadow ray of a vertex whose "
                         "direct term is exactly zero and the ray beyond the depth limit (intersected, then dropped, raytracer.cpp:960-963) are left out, "
                         "the image keeps every bit"}
        # HBM-side bytes per launch: not collectable inside this process (PMC needs rocprofv3 around it). Taken from the
        # committed counter pass of this same command ONLY if that pass was made with this very library build (hash stored
        # in the summary); FETCH_SIZE is in KB of 64 B requests, doubled for gfx950's 128-byte requests. Null otherwise.
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r3_%s_mem2.json" % args.config)))
            if pm.get("lib_sha16") == lib_sha16(rta.LIB_PATH) and pm.get("spp_step") == spp_step and pm.get("size") == W:
                kk = [k for k in pm["kernels"] if kname in k][0]
                raw = pm["kernels"][kk]["FETCH_SIZE"] * 1024.0 / pm["kernels"][kk]["dispatches"]
                roofline["traffic"] = round(raw * FETCH_GATHER_FACTOR, 1)
                roofline["traffic_range"] = [round(raw, 1), round(raw * FETCH_GATHER_FACTOR, 1)]
                roofline["traffic_note"] = ("fabric-side fetch bytes per launch from profiles/r3_%s_mem2.json (same library build). FETCH_SIZE tallies every request at "
                                            "64 bytes; calibrated on known byte counts in this kernel's access shapes (profiles/r3_fetch_size_calibration.json): exact for "
                                            "64-byte and smaller gathers, x1.39 for 128-byte record gathers (part of them go out as one 128-byte request), x2.0 only for "
                                            "coalesced streaming. `traffic` uses x1.39, the lower end of `traffic_range` x1.0" % args.config)
        except (OSError, KeyError, IndexError, ValueError, ZeroDivisionError):
            pass
        sz = (ctypes.c_uint64 * 4)()
        rta.lib.rt_debug_scene_sizes(r._h, sz)  # device bytes with every derived array; record counts
        sizes = [int(x) for x in sz]
        occ_b, occ_l = ctypes.c_int(), ctypes.c_int()
        rta.lib.rt_debug_occupancy(r._h, ctypes.byref(occ_b), ctypes.byref(occ_l))  # resident waves per CU with this scene's LDS use
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import bindings as ob
            t1 = time.time()
            orc = ob.OracleScene(scenes.export_objects(scene), scene.I, scene.L, math="libm")  # as the reference: the host's libm
            threads = min(ob.orc.orc_max_threads(), int(os.environ.get("OMP_NUM_THREADS", "0")) or 1 << 30)
            p = rta.make_params(W, H, cpu_spp, flags=flags)
            t2 = time.time()
            orc.render(cam, p, n_threads=threads)
            dt = time.time() - t2
            cpu = {"value": round(W * H * cpu_spp / dt / 1e6, 4), "unit": "Mpaths/s", "cores": threads, "kind": "port",
                   "sample": "same scene and frame, %d of %d spp (%.1f s of CPU work on %d OpenMP threads, schedule(dynamic,1) rows); "
                             "oracle/ref_cpu.cpp with the counter-based RNG (the reference's one shared engine makes its threads wait for each other)" % (cpu_spp, spp_image, dt, threads),
                   "setup_s": round(t2 - t1, 2)}
            # the reference itself, as committed: its one global random engine shared by all OpenMP threads and rand()
            # (raytracer.cpp:26-27, :1006, :1456) - built from the reference's own source where that exists
            # (oracle/_ref/libref_plain.so travels with the tree). Only the committed estimator exists in it (flags == 0).
            if flags == 0:
                ref_spp = max(1, cpu_spp // 6)
                dt_ref = ob.time_plain_reference(scenes.export_objects(scene), scene.I, scene.L, cam, rta.make_params(W, H, ref_spp, flags=0), threads)
                if dt_ref:
                    cpu["reference_shared_rng"] = {"value": round(W * H * ref_spp / dt_ref / 1e6, 4), "unit": "Mpaths/s", "cores": threads, "kind": "reference",
                                                   "sample": "%d spp, %.1f s: /root/reference/raytracer.cpp's own classes, OpenMP rows, shared engine(10) + rand() as committed" % (ref_spp, dt_ref)}
        out = {"metric": "Mpaths/s (whole node) final scene 1024x1024" if args.config == "cfg4" else "Mpaths/s (whole node) %s %dx%d" % (args.config, W, H),
               "value": round(value, 3), "unit": "Mpaths/s",
               "n_gpus": world, "world_size": dist.get_world_size() if world > 1 else 1, "backend": ("RCCL" if backend == "nccl" else backend) if world > 1 else None,
               "devices_on_node": n_dev, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak",
               "scaling_note": "per-rank work per step is fixed (value); time_to_image is the fixed-size image split over the ranks. No 1/2/4/8-GPU curve of "
                               "this build has been measured on hardware yet: only one-GPU boxes were available to it",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s, %d triangles, %d BVH records, %d spheres, %dx%d, %d spp per rank per step" % (what, info["n_tris"], info["n_nodes"], info["n_spheres"], W, H, spp_step),
                          "name": args.config, "spp_per_step": spp_step,
                          "parallelism": "spp-sharded x%d + %s reduce(sum) of the f64 accumulator" % (world, "RCCL" if backend == "nccl" else backend),
                          "scene_bytes_hbm": info["bytes"], "scene_bytes_device": sizes[0], "traversal_records": sizes[1],
                          "filter_tree_records": sizes[2], "waves_per_cu": min(occ_b.value, 8), "lds_bytes_per_wave": occ_l.value,
                          "setup": setup},
               "rays_per_s": round(value * 1e6 * g["rays"] / g["paths"], 1),
               "time_to_image": tti, "fast_mode": fast,
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
