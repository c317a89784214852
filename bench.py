#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the radiance loop on the dragonChrome scene at 1920x1080.

    python bench.py --gpus N --steps K --warmup W

One step = one full pass of the hot path over the frame: every rank renders `--spp` samples per pixel
(default 256, the single-GPU configuration of BASELINE.json) of the 871 200-triangle dragon stand-in in
the Cornell box into its own float sum buffer in HBM, the buffers are summed onto rank 0 with one RCCL
reduce, and rank 0 scales the sum to mean radiance.  Per-GPU work is fixed as N grows (weak scaling: the
N-GPU image has N*spp samples per pixel, each rank a disjoint range of global sample indices).  The scene
and its BVH are built and uploaded before the timed region; inputs are resident in HBM when timing starts.

Rank 0 prints one JSON line.  `value` is whole-job Mpaths/s (a path = one pixel sample = one top-level
radiance() call).  `roofline` prices the closest-hit kernel: algorithmic bytes per ray
(32 B ray + 8 B hit + 80 B x node visits + 48 B x triangle tests, the visit counts measured by the counting
build of the same kernel on the same rays) times rays per launch, over the kernel's mean launch time
measured with HIP events on the library's stream.  `cpu_baseline` times the CPU checker (oracle/) on a
bounded sample of the same workload on this box's host cores; it is a reported baseline, nothing else.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per GPU and step")
    ap.add_argument("--res", type=str, default="1920x1080")
    ap.add_argument("--scene", type=str, default="dragonChrome", choices=["dragonChrome", "dragonGlass", "dragonMilk", "cornell", "fog"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=32)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # WAPT_BENCH_REHEARSAL=1: the multi-rank control flow on a box with ONE GPU - every rank renders on cuda:0 and the
    # collectives go through host copies over gloo (the numbers mean nothing; the sequence of collectives is the same)
    rehearsal = world > 1 and os.environ.get("WAPT_BENCH_REHEARSAL", "0") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    def reduce_sum_to_rank0(x):
        if rehearsal:
            h = x.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            x.copy_(h)
        else:
            dist.reduce(x, dst=0, op=dist.ReduceOp.SUM)

    from pathtracer_amd import renderer, scenes
    W, H = (int(x) for x in args.res.lower().split("x"))
    if args.scene == "dragonChrome":
        sc = scenes.dragon("chrome", res=(W, H))
    elif args.scene == "dragonGlass":
        sc = scenes.dragon("glass", res=(W, H))
    elif args.scene == "dragonMilk":
        sc = scenes.dragon("milk", res=(W, H))
    elif args.scene == "fog":
        sc = scenes.fog(res=(W, H))
    else:
        sc = scenes.cornell(res=(W, H))
    t0 = time.time()
    content = renderer.wa_content(sc, device=dev.index)
    build_s = time.time() - t0
    info = content.bvh_info()
    # The work queues are sized by the free device memory at the first frame.  Memory of a process that has just ended
    # on this GPU (a test run before the benchmark, say) comes back over a few seconds: wait until it has, so that the
    # warm-up frame allocates what the timed frames will use.
    free_prev, stable = -1, 0
    t_wait = time.time()
    while time.time() - t_wait < 20.0:
        free_b, total_b = torch.cuda.mem_get_info(dev)
        if free_b >= 0.9 * total_b:
            break
        stable = stable + 1 if free_b == free_prev else 0
        if stable >= 4:
            break
        free_prev = free_b
        time.sleep(0.5)
    spp, total_spp = args.spp, args.spp * world
    r = renderer.udp_renderer(content, psamples=total_spp, hsamples=1, lsamples=1, seed=2026)
    acc = torch.zeros(W * H * 3, dtype=torch.float32, device=dev)
    image = torch.empty_like(acc)

    def step(profile=0):
        acc.zero_()
        torch.cuda.synchronize()
        st = r.render_accumulate(acc.data_ptr(), rank * spp, spp, profile=profile)
        if world > 1 and profile == 0:  # (the profiled frames below run on rank 0 alone: no collective there)
            reduce_sum_to_rank0(acc)
        if rank == 0:
            torch.mul(acc, 1.0 / total_spp, out=image)
        return st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays = 0
    for _ in range(args.steps):
        st = step()
        rays += st.rays
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(rays)], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, rays = float(tmax[0]), float(t[1])
    paths = float(W) * H * spp * world * args.steps
    out = {
        "metric": "Mpaths/s (whole job) " + args.scene + " 1080p; Mrays/s alongside",
        "value": paths / dt / 1e6,
        "unit": "Mpaths/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (procedural 871200-triangle dragon stand-in in the bundled Cornell box)",
        "mrays_per_s": rays / dt / 1e6,
        "rays_per_path": rays / paths,
        "config": {"workload": f"{args.scene} {W}x{H}, {spp} spp per GPU ({total_spp} spp image), hsamples 1, lsamples 1, td {sc.td}",
                   "triangles": int(info.n_tris), "bvh_nodes": int(info.n_nodes), "bvh_bytes": int(info.node_bytes + info.tri_bytes),
                   "parallelism": f"spp-sharded x{world}, one RCCL reduce per frame", "scene_build_s": round(build_s, 2)},
    }

    if rank == 0 and not args.no_roofline:
        # one profiled frame with HIP events around every launch, one with the counting closest-hit kernel
        st = step(profile=1)
        ms_trace, n_launch, n_rays = st.ms_trace, st.n_trace_launches, st.n_trace_rays
        fam = {k: getattr(st, "ms_" + k) for k in ("raygen", "trace", "head", "shade", "volume", "shadow", "probe")}
        shadow_rays = st.n_shadow_rays
        st = step(profile=2)
        nodes_per_ray = st.trace_node_visits / max(st.n_trace_rays, 1)
        tris_per_ray = st.trace_tri_tests / max(st.n_trace_rays, 1)
        util = {"wave_steps_per_ray_x64": 64.0 * st.trace_wave_steps / max(st.n_trace_rays, 1),
                "node_phase_lane_util": st.trace_node_visits / max(64.0 * st.trace_node_steps, 1),
                "tri_phase_lane_util": st.trace_tri_tests / max(64.0 * st.trace_tri_trips, 1),
                "node_phases": int(st.trace_node_steps), "tri_trips": int(st.trace_tri_trips),
                "shadow": {"node_visits_per_ray": st.shadow_node_visits / max(st.n_shadow_rays, 1),
                           "tri_tests_per_ray": st.shadow_tri_tests / max(st.n_shadow_rays, 1),
                           "wave_steps_per_ray_x64": 64.0 * st.shadow_wave_steps / max(st.n_shadow_rays, 1),
                           "node_phase_lane_util": st.shadow_node_visits / max(64.0 * st.shadow_node_steps, 1),
                           "tri_phase_lane_util": st.shadow_tri_tests / max(64.0 * st.shadow_tri_trips, 1)}}
        if st.n_probe_rays:
            util["probe"] = {"node_visits_per_ray": st.probe_node_visits / st.n_probe_rays,
                             "tri_tests_per_ray": st.probe_tri_tests / st.n_probe_rays,
                             "wave_steps_per_ray_x64": 64.0 * st.probe_wave_steps / st.n_probe_rays,
                             "node_phase_lane_util": st.probe_node_visits / max(64.0 * st.probe_node_steps, 1),
                             "tri_phase_lane_util": st.probe_tri_tests / max(64.0 * st.probe_tri_trips, 1)}
        bytes_per_ray = 32 + 8 + info.node_size * nodes_per_ray + info.tri_size * tris_per_ray
        rays_per_launch = n_rays / max(n_launch, 1)
        avg_ms = ms_trace / max(n_launch, 1)
        achieved = bytes_per_ray * rays_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if tj.get("workload") == out["config"]["workload"]:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": "k_trace (closest hit, one ray per lane)", "achieved": achieved,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "bytes_per_ray": bytes_per_ray, "node_visits_per_ray": nodes_per_ray,
                           "tri_tests_per_ray": tris_per_ray, "rays_per_launch": rays_per_launch,
                           "avg_launch_ms": avg_ms, "launches_per_step": int(n_launch),
                           "grays_per_s_in_kernel": rays_per_launch / (avg_ms * 1e-3) / 1e9,
                           "kernel_ms_per_step": fam, "lane_utilisation": util, "shadow_rays_per_step": int(shadow_rays), "probe_rays_per_step": int(st.n_probe_rays)}
        if traffic:
            # memory-side rate of the same kernel (PMC passes of the round profile): what HBM / Infinity Cache
            # actually carried, next to the algorithmic rate above
            out["roofline"]["traffic_gbs"] = traffic / (avg_ms * 1e-3) / 1e9
            out["roofline"]["traffic_frac"] = out["roofline"]["traffic_gbs"] / HBM_PEAK_GBS

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api
        cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        o = oracle_api.OracleScene(sc)
        t0 = time.perf_counter()
        o.render(args.cpu_spp, 1, 1, seed=2026, threads=cores)
        cdt = time.perf_counter() - t0
        cpaths = float(W) * H * args.cpu_spp
        out["cpu_baseline"] = {"value": cpaths / cdt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
                               "mrays_per_s": o.rays / cdt / 1e6,
                               "sample": f"same scene and resolution at {args.cpu_spp} spp ({cpaths / 1e6:.1f} M paths, {cdt:.1f} s); "
                                         "CPU restatement of the reference estimator with its own double-precision BVH "
                                         "(the reference's Embree path cannot be built here)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    content.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
