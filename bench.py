#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the radiance loop on the dragonChrome scene at 1920x1080.

    python bench.py --gpus N --steps K --warmup W

One step = one full pass of the hot path over the frame: the ranks together render the frame's samples per pixel of
the 871 200-triangle dragon stand-in in the Cornell box, each rank into its own float sum buffer in HBM, the buffers
are summed onto rank 0 with one RCCL reduce, and rank 0 scales the sum to mean radiance.  The scene and its BVH are
built and uploaded before the timed region; inputs are resident in HBM when timing starts.

Work per step
  N = 1 (default)       256 spp: the single-GPU configuration of BASELINE.json (configs[1]).  The queues hold 256 spp
                        of a 1080p frame, so a 1024-spp image on one GPU is four such steps at the same rate.
  N > 1 (default)       the target's own sharding: ONE image of --total-spp samples (default 1024), rank r renders
                        the contiguous block sample_range(1024, r, N) of global sample indices ("scaling": "strong").
  --spp S               S samples per pixel PER GPU instead (the image has N*S samples; "scaling": "weak").
  --total-spp T         one image of T samples whatever N is.

Rank 0 prints one JSON line.  `value` is whole-job Mpaths/s (a path = one pixel sample = one top-level radiance()
call).  `roofline` describes the dominant kernel, the closest-hit traversal k_trace, with the three figures that
bound it side by side:
  * `frac` = `achieved` / `peak`: the kernel's memory-side traffic (FETCH_SIZE + WRITE_SIZE counters, bytes per launch
    in `traffic`) over its average launch time (HIP events on the library's stream) against the 8 TB/s of HBM;
  * `algorithmic`: 32 B ray + 8 B hit + 80 B x node visits + 48 B x triangle tests per ray (the visit counts measured by
    the counting build of the same kernel on the same rays) x rays per second.  For a tree that fits the caches
    (dragonChrome: 52 MB) this rate EXCEEDS the HBM peak - the bytes are served by L1 / L2 / Infinity Cache - and is
    flagged `cache_served`: it is not a fraction of anything.  For the stress scene (3.3 GB tree) it is the figure
    that the memory-side traffic is to be compared with;
  * `valu_busy`: VALU issue cycles of the kernel's instructions over the SIMDs' cycles, once with every instruction
    at the hardware's floor of 2 cycles and once priced by kind (2 / 4 / 8 cycles: tools/probes/valu_rate.hip,
    profiles/valu_issue_cost.txt; classes from the SQ_INSTS_VALU_* counters, the rest at the mean cost of such
    instructions in the kernel's ISA, profiles/valu_issue_mix.json): the truth lies between the two;
  * `waits`: the share of wave-cycles spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) and waves per SIMD.
`bound` says what the evidence says: the kernel runs at the hardware's maximum of 8 waves per SIMD with neither HBM nor
the VALU port saturated - it is bound by the latency of the dependent round trips of a traversal step.
The counters are collected by THIS run: before the parent process touches the GPU it runs `rocprofv3 --pmc` four times
(one counter group each, no tracing) over a child that renders one frame of the same workload; if that fails, or with
more than one rank, the per-ray figures of the committed profile of the same scene are replayed, tagged with their source.
`cpu_baseline` times the CPU checker (oracle/, compiled -O3 -march=native for this host) on bounded samples of the
same workload on this box's host cores; it is a reported baseline, nothing else.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
L2_PEAK_GBS = 34500.0          # aggregate L2 bandwidth of the 8 XCDs
SIMDS = 256 * 4
CLOCK_GHZ = 2.4
VALU_PEAK_GINST = SIMDS * CLOCK_GHZ / 2.0     # G wave64 VALU instructions per second if every one took 2 cycles
VALU_PEAK_GCYCLES = SIMDS * CLOCK_GHZ         # G VALU issue cycles per second (all SIMDs)
ISSUE_COST = {"SQ_INSTS_VALU_FMA_F32": 2.0, "SQ_INSTS_VALU_MUL_F32": 2.0, "SQ_INSTS_VALU_ADD_F32": 2.0,
              "SQ_INSTS_VALU_CVT": 4.0, "SQ_INSTS_VALU_TRANS_F32": 8.0}

PMC_PASSES = [
    ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAIT_ANY",
            "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]),   # 8 SQ counters: the most one pass can hold
    ("mix", ["SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_CVT",
             "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_SALU"]),
    ("fetch", ["FETCH_SIZE", "TCC_HIT_sum"]),       # FETCH_SIZE takes 3 of the 4 TCC slots
    ("write", ["WRITE_SIZE", "TCC_MISS_sum"]),      # WRITE_SIZE takes 2
]


def family(kernel_name):
    k = kernel_name.replace("void ", "").split("(")[0].split("<")[0].strip()
    return k


def collect_pmc(child_args, log):
    """Runs the counter passes over a child of this script; returns {family: {counter: sum, "launches": n}}."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        raise RuntimeError("rocprofv3 not found")
    out = {}
    tmp = tempfile.mkdtemp(prefix="wapt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for tag, counters in PMC_PASSES:
            d = os.path.join(tmp, tag)
            cmd = ["timeout", "-s", "KILL", "150", rocprof, "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "pmc",
                   "--", sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", *child_args]
            t0 = time.time()
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
            log.append(f"pmc pass {tag}: rc {r.returncode}, {time.time() - t0:.1f} s")
            if r.returncode != 0:
                raise RuntimeError(f"rocprofv3 pass {tag} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}")
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                raise RuntimeError(f"rocprofv3 pass {tag} left no counter file")
            launches = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    fam = family(row["Kernel_Name"])
                    c = row["Counter_Name"]
                    o = out.setdefault(fam, {})
                    o[c] = o.get(c, 0.0) + float(row["Counter_Value"])
                    launches[(fam, c)] = launches.get((fam, c), 0) + 1
            for (fam, c), n in launches.items():
                out[fam]["launches"] = max(out[fam].get("launches", 0), n)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def static_mix(fam):
    """Mean issue cost of the instructions of a kernel family that no class counter breaks out, and of all its
    instructions, from the ISA of the shipped kernels (non-counting instances); (4.0, 4.0) if the file is missing."""
    try:
        ks = json.load(open(os.path.join(ROOT, "profiles", "valu_issue_mix.json")))["kernels"]
        sel = [v for k, v in ks.items() if k.split("<")[0] == fam and (("<" not in k) or k.split("<")[1].startswith("false"))]
        if sel:
            return (sum(v["rest_mean_cost"] for v in sel) / len(sel), sum(v["mean_cost"] for v in sel) / len(sel))
    except Exception:  # noqa: BLE001
        pass
    return (4.0, 4.0)


def issue_cycles(c, fam):
    """VALU issue cycles of a kernel family's launches from its counters (see the module docstring); None without them."""
    n = c.get("SQ_INSTS_VALU")
    if not n:
        return None, None
    rest_cost, mean_cost = static_mix(fam)
    if all(k in c for k in ISSUE_COST):
        counted = sum(c[k] for k in ISSUE_COST)
        cycles = sum(c[k] * w for k, w in ISSUE_COST.items()) + max(n - counted, 0.0) * rest_cost
        how = {"classes_per_inst": {k.replace("SQ_INSTS_VALU_", "").lower(): c[k] / n for k in ISSUE_COST},
               "rest_per_inst": max(n - counted, 0.0) / n, "rest_mean_cost": rest_cost}
    else:
        cycles = n * mean_cost
        how = {"mean_cost_static": mean_cost}
    return cycles, how


def git_sha():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        return None


def build_native_oracle(log):
    """The CPU checker compiled for THIS host (-O3 -march=native) into a scratch directory; None if that fails."""
    src = os.path.join(ROOT, "oracle", "oracle.cpp")
    d = tempfile.mkdtemp(prefix="wapt_oracle_", dir="/tmp")
    lib = os.path.join(d, "liboracle_native.so")
    cmd = ["g++", "-std=c++17", "-O3", "-march=native", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", lib, "-lpthread"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
        if r.returncode == 0:
            return lib
        log.append("native checker build failed: " + r.stderr[-200:])
    except Exception as e:  # noqa: BLE001
        log.append(f"native checker build failed: {e}")
    return None


def make_scene(scenes, name, W, H, materials="mixed"):
    if name == "dragonChrome":
        return scenes.dragon("chrome", res=(W, H))
    if name == "dragonGlass":
        return scenes.dragon("glass", res=(W, H))
    if name == "dragonMilk":
        return scenes.dragon("milk", res=(W, H))
    if name == "fog":
        return scenes.fog(res=(W, H))
    if name == "stress":
        return scenes.stress(res=(W, H), materials=materials)
    return scenes.cornell(res=(W, H))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel PER GPU and step (weak scaling)")
    ap.add_argument("--total-spp", type=int, default=None, help="samples per pixel of the whole image, sharded over the GPUs (strong scaling)")
    ap.add_argument("--res", type=str, default=None, help="default 1920x1080 (stress: 3840x2160)")
    ap.add_argument("--scene", type=str, default="dragonChrome", choices=["dragonChrome", "dragonGlass", "dragonMilk", "cornell", "fog", "stress"])
    ap.add_argument("--materials", type=str, default="mixed", choices=["mixed", "opaque"],
                    help="stress scene: diffuse / mirror / glass / milk cycling, or diffuse / mirror only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 counter passes (replay the committed profile)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dump-pmc", type=str, default=None, help="write the counters collected by this run to FILE (the committed profile)")
    ap.add_argument("--cpu-spp", type=int, default=16)
    ap.add_argument("--dump-image", type=str, default=None, help="rank 0 writes the last timed frame (mean radiance, H x W x 3 float32) to FILE.npy")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.res is None:
        args.res = "3840x2160" if args.scene == "stress" else "1920x1080"
    W, H = (int(x) for x in args.res.lower().split("x"))

    # work per rank
    from pathtracer_amd import distributed
    if args.spp is not None and args.total_spp is not None:
        ap.error("--spp and --total-spp exclude each other")
    if args.spp is None and args.total_spp is None:
        # one GPU: the single-GPU configuration (256 spp; a 1024-spp image is four such steps at the same rate);
        # more: the target's 1024-spp image, spp / N per GPU
        args.total_spp = (8 if args.scene == "stress" else 256) if world == 1 else 1024
    if args.total_spp is not None:
        total_spp = args.total_spp
        begin, count = distributed.sample_range(total_spp, rank, world)
        scaling = "strong"
        work = f"{total_spp} spp image sharded over {world} GPU(s) ({total_spp // world}{'+' if total_spp % world else ''} spp each)"
    else:
        total_spp = args.spp * world
        begin, count = rank * args.spp, args.spp
        scaling = "weak"
        work = f"{args.spp} spp per GPU ({total_spp} spp image)"

    log = []
    pmc, pmc_source = None, None
    want_roofline = rank == 0 and not args.no_roofline and not args.pmc_child
    if want_roofline and world == 1 and not args.no_pmc:
        # Counter passes first: the children need the GPU's memory, and this process has not touched the GPU yet
        # (it only starts other programs before it does).
        try:
            child = ["--res", args.res, "--scene", args.scene, "--materials", args.materials, "--spp", str(count)]
            pmc = collect_pmc(child, log)
            pmc_source = "rocprofv3 --pmc passes of this run (one frame of the same workload each)"
        except Exception as e:  # noqa: BLE001
            log.append(f"live counter passes failed: {e}")
            pmc = None

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    # WAPT_BENCH_REHEARSAL=1: the multi-rank control flow on a box with ONE GPU - every rank renders on cuda:0 and the
    # collectives go through host copies over gloo (the numbers mean nothing; the sequence of collectives is the same)
    rehearsal = world > 1 and os.environ.get("WAPT_BENCH_REHEARSAL", "0") == "1"
    # WAPT_BENCH_FORCE_DIST=1: also with one rank the frame goes through RCCL (communicator of one; launched by
    # torch.distributed.run): the collectives of the N > 1 flow next to this library's own streams, on one GPU
    collectives = world > 1 or os.environ.get("WAPT_BENCH_FORCE_DIST", "0") == "1"
    if collectives:
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
    dev = torch.device("cuda", local_rank if collectives else 0)

    def reduce_sum_to_rank0(x):
        if rehearsal:
            h = x.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            x.copy_(h)
        else:
            dist.reduce(x, dst=0, op=dist.ReduceOp.SUM)

    from pathtracer_amd import renderer, scenes
    sc = make_scene(scenes, args.scene, W, H, args.materials)
    t0 = time.time()
    content = renderer.wa_content(sc, device=dev.index)
    build_s = time.time() - t0
    info = content.bvh_info()
    # The work queues are sized by the free device memory at the first frame.  Memory of a process that has just ended
    # on this GPU (a test run or the counter passes before the benchmark) comes back over a few seconds: wait until it
    # has, so that the warm-up frame allocates what the timed frames will use.
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
    r = renderer.udp_renderer(content, psamples=total_spp, hsamples=1, lsamples=1, seed=2026)
    acc = torch.zeros(W * H * 3, dtype=torch.float32, device=dev)
    image = torch.empty_like(acc)

    def step(profile=0):
        acc.zero_()
        torch.cuda.synchronize()
        st = r.render_accumulate(acc.data_ptr(), begin, count, profile=profile)
        if collectives and profile == 0:  # (the profiled frames below run on rank 0 alone: no collective there)
            reduce_sum_to_rank0(acc)
        if rank == 0:
            torch.mul(acc, 1.0 / total_spp, out=image)
        return st

    if args.pmc_child:
        step()
        torch.cuda.synchronize()
        content.close()
        return

    def fence():
        torch.cuda.synchronize()
        if collectives:
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
    if collectives:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, rays = float(tmax[0]), float(t[1])
    if rank == 0 and args.dump_image:
        np.save(args.dump_image, image.reshape(H, W, 3).cpu().numpy())
    paths = float(W) * H * total_spp * args.steps
    out = {
        "metric": "Mpaths/s (whole job) " + args.scene + (" 1080p" if (W, H) == (1920, 1080) else f" {W}x{H}") + "; Mrays/s alongside",
        "value": paths / dt / 1e6,
        "unit": "Mpaths/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (procedural 871200-triangle dragon stand-in in the bundled Cornell box"
                + ("; 64 transformed copies of it flattened into one soup)" if args.scene == "stress" else ")"),
        "mrays_per_s": rays / dt / 1e6,
        "rays_per_path": rays / paths,
        "config": {"workload": f"{args.scene} {W}x{H}, {work}, hsamples 1, lsamples 1, td {sc.td}",
                   "triangles": int(info.n_tris), "bvh_nodes": int(info.n_nodes), "bvh_bytes": int(info.node_bytes + info.tri_bytes),
                   "parallelism": f"spp-sharded x{world}, one RCCL reduce per frame", "scene_build_s": round(build_s, 2),
                   "queue_bytes": int(st.queue_bytes), "queue_tasks_per_depth": int(st.queue_tasks),
                   "queue_budget_gb": os.environ.get("WAPT_QUEUE_BUDGET_GB")},
    }

    if want_roofline:
        # one profiled frame with HIP events around every launch, one with the counting closest-hit kernel
        st = step(profile=1)
        ms_trace, n_launch, n_rays = st.ms_trace, st.n_trace_launches, st.n_trace_rays
        fam_ms = {k: getattr(st, "ms_" + k) for k in ("raygen", "trace", "head", "shade", "volume", "shadow", "probe")}
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
        grays_kernel = rays_per_launch / (avg_ms * 1e-3) / 1e9
        algorithmic_gbs = bytes_per_ray * grays_kernel

        pmc_rays = None
        scene_tag = f"{args.scene}{'/' + args.materials if args.scene == 'stress' else ''} {W}x{H}"
        if pmc is None:
            # replay of the committed counter summary of the same scene, tagged with where it comes from (a run with
            # more than one rank collects no counters: the per-ray figures of the kernel do not depend on the rank count)
            pf = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pf):
                try:
                    pj = json.load(open(pf))
                    if pj.get("scene") == scene_tag:
                        pmc = pj["kernels"]
                        pmc_rays = pj.get("trace_rays")
                        pmc_source = f"replayed from profiles/pmc_latest.json @{pj.get('commit')} ({pj.get('workload')}); counters were not collected in this run"
                except Exception as e:  # noqa: BLE001
                    log.append(f"committed counter summary unreadable: {e}")
        else:
            pmc_rays = n_rays   # the counter passes rendered this very frame

        alg = {"bytes_per_ray": bytes_per_ray, "gbs": algorithmic_gbs, "ratio_to_hbm_peak": algorithmic_gbs / HBM_PEAK_GBS,
               "ratio_to_l2_peak": algorithmic_gbs / L2_PEAK_GBS, "node_visits_per_ray": nodes_per_ray, "tri_tests_per_ray": tris_per_ray,
               "cache_served": bool(algorithmic_gbs > HBM_PEAK_GBS),
               "note": "32 B ray + 8 B hit + node and triangle records visited; above the HBM peak the bytes come from L1 / L2 / Infinity "
                       "Cache and the ratio is not a fraction of anything"}
        roof = {"bound": "latency of the dependent round trips of a traversal step at the hardware's maximum of 8 waves per SIMD "
                         "(neither HBM nor VALU issue saturated)",
                "kernel": "k_trace (closest hit, one ray per lane)",
                "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "frac_definition": "memory-side bytes of the kernel (FETCH_SIZE + WRITE_SIZE counters) / its launch time (HIP events) / 8 TB/s",
                "counter_source": pmc_source,
                "rays_per_launch": rays_per_launch, "avg_launch_ms": avg_ms, "launches_per_step": int(n_launch),
                "grays_per_s_in_kernel": grays_kernel,
                "algorithmic": alg,
                "kernel_ms_per_step": fam_ms, "lane_utilisation": util,
                "shadow_rays_per_step": int(shadow_rays), "probe_rays_per_step": int(st.n_probe_rays)}
        if pmc:
            kt = pmc.get("k_trace", {})
            rays_c = float(pmc_rays or 0)   # closest-hit rays of the frame(s) the counters cover
            if "FETCH_SIZE" in kt and "WRITE_SIZE" in kt and rays_c:
                f_raw, w = kt["FETCH_SIZE"] * 1024.0, kt["WRITE_SIZE"] * 1024.0   # the counters are in KiB; sums over the frame's launches
                bpr, bpr_x2 = (f_raw + w) / rays_c, (2.0 * f_raw + w) / rays_c
                # bytes per ray from the counters x this run's rays per launch and launch time
                roof["traffic_bytes_per_ray"] = bpr
                roof["traffic"] = bpr * rays_per_launch
                roof["achieved"] = bpr * grays_kernel
                roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
                # raw FETCH_SIZE: the factor 2 that gfx950 needs for wide coalesced streaming reads is not calibrated for
                # this kernel's 16-byte gathers; the doubled figure is kept beside it as the upper bound
                roof["traffic_fetch_x2"] = bpr_x2 * rays_per_launch
                roof["frac_fetch_x2"] = bpr_x2 * grays_kernel / HBM_PEAK_GBS
                roof["traffic_over_algorithmic"] = bpr / bytes_per_ray
                if kt.get("TCC_HIT_sum") is not None and kt.get("TCC_MISS_sum") is not None:
                    roof["l2_hit_rate"] = kt["TCC_HIT_sum"] / max(kt["TCC_HIT_sum"] + kt["TCC_MISS_sum"], 1.0)
            if kt.get("SQ_INSTS_VALU") and rays_c:
                ipr = kt["SQ_INSTS_VALU"] / rays_c   # wave-instructions of the frame's launches / its rays
                cyc, how = issue_cycles(kt, "k_trace")
                vb = {"valu_insts_per_ray": ipr, "issue_cycles_per_ray_priced": cyc / rays_c, "issue_mix": how,
                      "at_2_cycle_floor": ipr * grays_kernel / VALU_PEAK_GINST,
                      "priced_by_kind": cyc / rays_c * grays_kernel / VALU_PEAK_GCYCLES,
                      "peak_definition": "1024 SIMDs x 2.4 GHz; a wave64 instruction takes 2, 4 or 8 issue cycles by kind "
                                         "(tools/probes/valu_rate.hip, profiles/valu_issue_cost.txt)"}
                if pmc_rays == n_rays and kt.get("GRBM_GUI_ACTIVE") and ms_trace:
                    # the counter sums the 8 XCDs; over the kernel's time it gives the clock the launch really ran at
                    vb["clock_ghz_measured"] = kt["GRBM_GUI_ACTIVE"] / 8.0 / (ms_trace * 1e-3) / 1e9
                roof["valu_busy"] = vb
            if kt.get("SQ_WAVE_CYCLES") and kt.get("SQ_WAIT_ANY") is not None:
                roof["waits"] = {"wait_any_frac": kt["SQ_WAIT_ANY"] / kt["SQ_WAVE_CYCLES"],
                                 "wait_inst_frac": (kt.get("SQ_WAIT_INST_ANY") or 0.0) / kt["SQ_WAVE_CYCLES"],
                                 "active_inst_frac": (kt.get("SQ_ACTIVE_INST_ANY") or 0.0) / kt["SQ_WAVE_CYCLES"],
                                 "waves_per_simd": 8, "vgprs": 64,
                                 "note": "share of the waves' resident cycles spent waiting on anything / waiting to issue / issuing"}
            # the whole frame: memory-side bytes of every kernel of the step over the step's time
            per_kernel, raw_sum, x2_sum = {}, 0.0, 0.0
            scale = (n_rays / rays_c) if rays_c else 1.0   # (a replayed profile of another spp count, scaled to this frame)
            for fam, c in sorted(pmc.items()):
                if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or not fam.startswith("k_"):
                    continue
                f_raw, w = c["FETCH_SIZE"] * 1024.0 * scale, c["WRITE_SIZE"] * 1024.0 * scale
                per_kernel[fam] = {"fetch_raw_gb": f_raw / 1e9, "write_gb": w / 1e9, "launches": c.get("launches"),
                                   "valu_insts": c.get("SQ_INSTS_VALU")}
                fam_key = {"k_trace": "trace", "k_surface": "shade", "k_shadow": "shadow", "k_probe": "probe", "k_volume": "volume"}.get(fam)
                cyc_f, _ = issue_cycles(c, fam)
                if fam_key and cyc_f and fam_ms.get(fam_key):
                    per_kernel[fam]["valu_issue_frac"] = cyc_f * scale / (fam_ms[fam_key] * 1e-3) / 1e9 / VALU_PEAK_GCYCLES
                    per_kernel[fam]["hbm_gbs_raw"] = (f_raw + w) / (fam_ms[fam_key] * 1e-3) / 1e9
                if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_ANY") is not None:
                    per_kernel[fam]["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
                raw_sum += f_raw + w
                x2_sum += 2.0 * f_raw + w
            step_s = sum(v for v in fam_ms.values() if v) * 1e-3   # kernel time of a frame on one GPU
            roof["frame"] = {"hbm_gb_raw": raw_sum / 1e9, "hbm_gb_fetch_x2": x2_sum / 1e9,
                             "frame_hbm_gbs_raw": raw_sum / step_s / 1e9, "frame_hbm_gbs": x2_sum / step_s / 1e9,
                             "frame_hbm_frac_raw": raw_sum / step_s / 1e9 / HBM_PEAK_GBS,
                             "frame_hbm_frac": x2_sum / step_s / 1e9 / HBM_PEAK_GBS, "kernels": per_kernel,
                             "note": "queue reads and writes are 16-byte-per-lane streams, for which FETCH_SIZE reads half the bytes on "
                                     "gfx950: frame_hbm_gbs doubles FETCH_SIZE (upper bound where the reads are BVH gathers), _raw does not"}
        out["roofline"] = roof
        if args.dump_pmc and pmc and pmc_source and pmc_source.startswith("rocprofv3"):
            json.dump({"scene": scene_tag, "workload": out["config"]["workload"], "commit": git_sha(), "trace_rays": int(n_rays), "kernels": pmc,
                       "units": "sums over the launches of one frame; FETCH_SIZE and WRITE_SIZE in KiB"},
                      open(args.dump_pmc, "w"), indent=1)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api
        native = build_native_oracle(log)
        if native:
            oracle_api.LIB = native
            oracle_api._lib = None
            oracle_api.build = lambda: native
        flags = "-O3 -march=native (compiled on this host)" if native else "-O2 (prebuilt; the native build failed)"
        share = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        rows = []

        def cpu_row(o, name, spp, threads):
            t0 = time.perf_counter()
            o.render(spp, 1, 1, seed=2026, threads=threads)
            cdt = time.perf_counter() - t0
            w_, h_ = o.scene.resolution
            cpaths = float(w_) * h_ * spp
            rows.append({"workload": f"{name} {w_}x{h_}, {spp} spp", "threads": threads, "mpaths_per_s": cpaths / cdt / 1e6,
                         "mrays_per_s": o.rays / cdt / 1e6, "seconds": cdt, "paths": cpaths})
            return rows[-1]

        # the benchmark's own workload at reduced spp: all cores of this job's share, and the reference's 4 threads
        # (src/unidirectionalPathTracer.cpp:59)
        o = oracle_api.OracleScene(sc)
        head = cpu_row(o, args.scene, args.cpu_spp, share)
        cpu_row(o, args.scene, max(args.cpu_spp // 4, 1), 4)
        o.close()
        # BASELINE config 0 in full: Cornell box with the two spheres, 512x512, 64 spp (the reference's CPU-runnable case)
        o = oracle_api.OracleScene(scenes.cornell(res=(512, 512)))
        cpu_row(o, "cornell+spheres (BASELINE config 0, in full)", 64, share)
        cpu_row(o, "cornell+spheres (BASELINE config 0, in full)", 64, 4)
        o.close()
        out["cpu_baseline"] = {"value": head["mpaths_per_s"], "unit": "Mpaths/s", "cores": share, "kind": "port",
                               "mrays_per_s": head["mrays_per_s"], "build": flags,
                               "host_cpus_visible": os.cpu_count(),
                               "sample": f"same scene and resolution at {args.cpu_spp} spp ({head['paths'] / 1e6:.1f} M paths, "
                                         f"{head['seconds']:.1f} s) on the {share} host threads of this job's share; CPU restatement of the "
                                         "reference estimator with its own double-precision BVH (the reference's Embree path cannot be "
                                         "built here); rows: the same with the reference's 4 threads, and BASELINE config 0",
                               "rows": rows}
    if rank == 0:
        if log:
            out["log"] = log
        print(json.dumps(out), flush=True)
    content.close()
    if collectives:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
