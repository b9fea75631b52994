#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s on the final battle scene (raytracer.cpp:1222-1435), 1024x1024.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = every rank renders --spp-step samples of every pixel of the 1024x1024 frame (1 path =
1 camera sample = one Scene::getColor(r,0,false) call, raytracer.cpp:1485). Ranks own disjoint global
sample indices (spp sharding; weak scaling: per-rank work per step is fixed). Inside the timed
region, after the K steps, the FP64 radiance accumulators are summed onto rank 0 (RCCL reduce over
xGMI) and rank 0 resolves the 8-bit image, so `value` includes everything between "scene resident
in HBM" and "image in host memory". Scene upload and BVH build are outside the timed region.

The OBJ/JPG assets of the reference are not redistributable and are absent on the GPU box:
procedural stand-ins of the same triangle counts, object-space extents and transform chains are
used (ray-tracing_amd/scenes.py: battle()).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_VECTOR_TFLOPS = 78.6   # MI355X vector FP64 = half the 157.3 TFLOP/s FP32 vector peak
PEAK_HBM_GBS = 8000.0
PROFILE_PREFIX = "r1_v9_eight_waves_"  # rocprofv3 summaries of this round's kernel under profiles/


def algorithmic_flops(c, n_spheres):
    """Reference-algorithm flops of a counted run (add/sub/mul/div/sqrt/compare = 1; cos/sin/log/pow are
    counted separately as transcendentals): per ray n_spheres*20 + 25 for the sphere that is hit,
    24 per box test, 54 per triangle test; per path 60 for the camera sample; 150 per diffuse vertex,
    40 per specular vertex."""
    flops = (c["rays"] * (n_spheres * 20 + 25) + c["box_tests"] * 24 + c["tris"] * 54 + c["paths"] * 60 +
             c["diffuse"] * 150 + c["specular"] * 40)
    transc = c["paths"] * 8 + c["diffuse"] * 4
    return flops, transc


def algorithmic_bytes(c, node_bytes=128):
    """node_bytes per BVH record visited (128 B = two FP64 boxes + child links, what the algorithm as the
    reference states it reads; the kernel's own records are 64 B with float boxes), 80 B per triangle
    tested, 128 B of attributes per mesh hit (upper bound: one per mesh call), 24 B accumulate per path."""
    return c["nodes"] * node_bytes + c["tris"] * 80 + c["mesh_calls"] * 128 + c["paths"] * 24


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-step", type=int, default=64, help="samples per pixel per rank per step (one kernel launch up to 64)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=12, help="samples per pixel of the CPU baseline's bounded sample (12 spp = about 15 s on 16 cores)")
    ap.add_argument("--out-png", default="")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world), file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist

    # RT_BENCH_BACKEND=gloo is for rehearsing the multi-rank path on a box with fewer GPUs than ranks (ranks then share
    # devices, which RCCL refuses); the measured configuration is one GPU per rank over RCCL.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    rta = importlib.import_module("ray-tracing_amd")
    scenes = importlib.import_module("ray-tracing_amd.scenes")

    W = H = args.size
    t0 = time.time()
    scene, cam, flags, meta = scenes.battle(W=W)
    r = rta.Renderer(local_rank)
    r.upload(scene)
    info = scene.flat_info()
    setup_s = time.time() - t0

    accum = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    acc_ptr = accum.data_ptr()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(index):
        # global sample indices of this rank in this step: disjoint over (step, rank)
        off = (index * world + rank) * args.spp_step
        r.render(cam, rta.make_params(W, H, args.spp_step, spp_offset=off, flags=flags), acc_ptr)

    for w in range(args.warmup):
        step(w)
    r.sync()
    if world > 1 and args.warmup > 0:
        # the timed region ends with one sum-reduction of the accumulator: let RCCL set up its connections for
        # that collective and size outside it, as the warm-up steps do for the kernel
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    accum.zero_()
    barrier()

    kernel_ms, kernel_launches = 0.0, 0
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
        if rank == 0:  # HIP-event time of this step's kernels on the render stream (waits for them)
            ms, nl = r.last_render_ms()
            kernel_ms += ms
            kernel_launches += nl
    r.sync()
    if world > 1:
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()  # the resolve below runs on the renderer's own stream: the sum must have landed
    image = None
    if rank == 0:
        image = r.resolve(W, H, args.spp_step * args.steps * world, acc_ptr)
    barrier()
    elapsed = time.perf_counter() - t_start

    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    paths_total = float(W) * H * args.spp_step * args.steps * world
    value = paths_total / elapsed / 1e6

    if rank == 0:
        # counted runs (separate, untimed). "Algorithmic" work is counted under the REFERENCE's traversal
        # semantics (every node whose box the ray's line crosses, RT_F_NO_PRUNE) so that the figure does not
        # depend on this implementation; the work the default (pruned, any-hit shadow) traversal really does
        # is reported beside it.
        cnt_spp = 2
        scratch = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
        counted = {}
        for label, extra in (("reference", rta.RT_F_NO_PRUNE), ("gpu", 0)):
            r.reset_counters()
            r.render(cam, rta.make_params(W, H, cnt_spp, spp_offset=0, flags=flags | rta.RT_F_COUNTERS | extra), scratch.data_ptr())
            r.sync()
            counted[label] = r.counters()
        c = counted["reference"]
        g = counted["gpu"]
        flops, transc = algorithmic_flops(c, info["n_spheres"])
        flops_per_path = flops / c["paths"]
        bytes_per_path = algorithmic_bytes(c) / c["paths"]
        gpu_flops_per_path = algorithmic_flops(g, info["n_spheres"])[0] / g["paths"]
        gpu_bytes_per_path = algorithmic_bytes(g, node_bytes=64) / g["paths"]
        paths_per_launch = float(W) * H * args.spp_step * args.steps / max(1, kernel_launches)
        avg_launch_s = kernel_ms / 1e3 / max(1, kernel_launches)
        achieved_tflops = flops_per_path * paths_per_launch / avg_launch_s / 1e12
        roofline = {"bound": "valu_fp64", "achieved": round(achieved_tflops, 4), "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved_tflops / PEAK_FP64_VECTOR_TFLOPS, 5), "traffic": None,
                    "kernel": "rt_render_kernel<false, true>", "avg_launch_ms": round(avg_launch_s * 1e3, 3), "launches": kernel_launches,
                    "algorithmic_flops_per_path": round(flops_per_path, 1), "transcendentals_per_path": round(transc / c["paths"], 2),
                    "algorithmic_bytes_per_path": round(bytes_per_path, 1),
                    "hbm_side": {"achieved_GBs": round(bytes_per_path * paths_per_launch / avg_launch_s / 1e9, 2), "peak_GBs": PEAK_HBM_GBS,
                                 "frac": round(bytes_per_path * paths_per_launch / avg_launch_s / 1e9 / PEAK_HBM_GBS, 4)},
                    "rays_per_path": round(c["rays"] / c["paths"], 3), "nodes_per_path": round(c["nodes"] / c["paths"], 2),
                    "tris_per_path": round(c["tris"] / c["paths"], 2),
                    "gpu_own_work": {"nodes_per_path": round(g["nodes"] / g["paths"], 2), "tris_per_path": round(g["tris"] / g["paths"], 2),
                                     "flops_per_path": round(gpu_flops_per_path, 1), "bytes_per_path": round(gpu_bytes_per_path, 1),
                                     "TFLOPs": round(gpu_flops_per_path * paths_per_launch / avg_launch_s / 1e12, 4),
                                     "GBs": round(gpu_bytes_per_path * paths_per_launch / avg_launch_s / 1e9, 2)}}
        # HBM-side bytes per launch: not collectable inside this process (PMC needs rocprofv3 around it); taken from the
        # committed counter pass of this same command, FETCH_SIZE (KB, 64 B per L2 fabric read request) doubled as the
        # gfx950 correction for 128-byte requests prescribes. Null when the summary is absent.
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", PROFILE_PREFIX + "mem2.json")))
            kk = [k for k in pm if "rt_render_kernel<false, true>" in k][0]
            roofline["traffic"] = round(pm[kk]["FETCH_SIZE"] * 1024.0 * 2.0 / pm[kk]["dispatches"], 1)
            roofline["traffic_note"] = ("bytes per launch of 1024x1024x64 paths from profiles/%smem2.json (rocprofv3 --pmc FETCH_SIZE of "
                                        "`bench.py --steps 2 --warmup 1 --no-cpu-baseline`, 64 spp per launch), x2 gfx950 correction; "
                                        "algorithmic bytes per launch = %.4g" % (PROFILE_PREFIX, bytes_per_path * paths_per_launch))
            if args.spp_step != 64 or W != 1024:
                roofline["traffic"] = None  # the committed counter pass is of the default launch size only
                roofline["traffic_note"] = "no counter pass for this launch size"
        except (OSError, KeyError, IndexError, ValueError, ZeroDivisionError):
            pass
        import ctypes
        sz = (ctypes.c_uint64 * 4)()
        rta.lib.rt_debug_scene_sizes(r._h, sz)  # device bytes with every derived array; record counts
        sizes = [int(x) for x in sz]
        occ_b, occ_l = ctypes.c_int(), ctypes.c_int()
        rta.lib.rt_debug_occupancy(r._h, ctypes.byref(occ_b), ctypes.byref(occ_l))  # resident waves per CU with this scene's LDS use
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import bindings as ob
            t1 = time.time()
            orc = ob.OracleScene(scenes.export_objects(scene), scene.I, scene.L)
            threads = min(ob.orc.orc_max_threads(), int(os.environ.get("OMP_NUM_THREADS", "0")) or 1 << 30)
            p = rta.make_params(W, H, args.cpu_spp, flags=flags)
            t2 = time.time()
            orc.render(cam, p, n_threads=threads)
            dt = time.time() - t2
            cpu = {"value": round(W * H * args.cpu_spp / dt / 1e6, 4), "unit": "Mpaths/s", "cores": threads, "kind": "port",
                   "sample": "same scene and frame, %d spp (%.1f s of CPU work on %d OpenMP threads, schedule(dynamic,1) rows); "
                             "oracle/ref_cpu.cpp with the counter-based RNG" % (args.cpu_spp, dt, threads),
                   "setup_s": round(t2 - t1, 2)}
        if args.out_png and image is not None:
            rta.write_png(args.out_png, image)
        out = {"metric": "Mpaths/s (whole node) final scene 1024x1024", "value": round(value, 3), "unit": "Mpaths/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "cfg4 battle scene (6 spheres + 12 procedural stand-in meshes, %d triangles, %d BVH records, "
                                      "textures, DoF sigma 0.01), %dx%d, %d spp per rank per step; time-to-image at 1000 spp "
                                      "extrapolates to %.2f s on this many GPUs" % (info["n_tris"], info["n_nodes"], W, H, args.spp_step,
                                                                                      1000.0 * W * H / (value * 1e6)),
                          "spp_per_step": args.spp_step, "parallelism": "spp-sharded x%d + %s reduce(sum) of the f64 accumulator" % (world, "RCCL" if backend == "nccl" else backend),
                          "scene_bytes_hbm": info["bytes"], "scene_bytes_device": sizes[0], "traversal_records": sizes[1],
                          "filter_tree_records": sizes[2], "waves_per_cu": min(occ_b.value, 8), "lds_bytes_per_wave": occ_l.value,
                          "setup_s": round(setup_s, 2)},
               "rays_per_s": round(value * 1e6 * c["rays"] / c["paths"], 1),
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
