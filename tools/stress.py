#!/usr/bin/env python3
"""Stress configuration: `--copies` transformed dragon stand-ins flattened into one triangle soup with
mixed materials (diffuse / mirror / glass / milk), rendered at `--res`.  Prints one JSON line with the scene
build time, BVH facts and the render rate."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=64)
    ap.add_argument("--res", default="3840x2160")
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--builder", default="auto", choices=["auto", "host", "device"])
    ap.add_argument("--materials", default="mixed", choices=["mixed", "opaque"])
    args = ap.parse_args()
    import torch
    from pathtracer_amd import renderer, scenes
    W, H = (int(x) for x in args.res.split("x"))
    t0 = time.time()
    sc = scenes.stress(res=(W, H), copies=args.copies, materials=args.materials)
    t_mesh = time.time() - t0
    t0 = time.time()
    content = renderer.wa_content(sc, device=0, builder=args.builder)
    t_build = time.time() - t0
    info = content.bvh_info()
    r = renderer.udp_renderer(content, psamples=args.spp, hsamples=1, lsamples=1, seed=7)
    acc = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
    r.render_accumulate(acc.data_ptr(), 0, min(args.spp, 1))  # warm-up: allocates the queues
    acc.zero_()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        st = r.render_accumulate(acc.data_ptr(), 0, args.spp)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.steps
    img = (acc / (args.spp * args.steps)).cpu()
    out = {"copies": args.copies, "materials": args.materials, "triangles": int(info.n_tris), "bvh_nodes": int(info.n_nodes), "bvh_depth": int(info.max_depth), "builder": int(info.builder), "build_s": round(float(info.build_seconds), 2), "sah": float(info.sah_cost),
           "bvh_gb": (info.node_bytes + info.tri_bytes) / 1e9, "mesh_s": round(t_mesh, 1), "scene_create_s": round(t_build, 1),
           "res": args.res, "spp": args.spp, "seconds": dt, "mpaths_per_s": W * H * args.spp / dt / 1e6,
           "mrays_per_s": st.rays / dt / 1e6, "rays_per_path": st.rays / (W * H * args.spp), "stack_overflows": int(st.stack_overflows),
           "mean_radiance": float(img.mean()), "finite": bool(torch.isfinite(img).all())}
    print(json.dumps(out), flush=True)
    content.close()


if __name__ == "__main__":
    main()
