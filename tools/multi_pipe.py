#!/usr/bin/env python3
"""Experiment: aggregate throughput of K independent render pipelines (one scene handle and host thread each) on one
GPU, against a single pipeline: tells how much overlapping the bandwidth-bound surface kernel of one pipeline with the
VALU-bound traversal kernels of another can gain."""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pipes", type=int, default=2)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from pathtracer_amd import renderer, scenes
    W, H = 1920, 1080
    sc = scenes.dragon("chrome", res=(W, H))
    cs = [renderer.wa_content(sc, device=0) for _ in range(args.pipes)]
    rs = [renderer.udp_renderer(c, args.spp * args.pipes, 1, 1, 1, seed=2026) for c in cs]
    accs = [torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0") for _ in cs]
    per = args.spp

    def work(k, steps):
        for _ in range(steps):
            rs[k].render_accumulate(accs[k].data_ptr(), k * per, per)

    def run(steps):
        th = [threading.Thread(target=work, args=(k, steps)) for k in range(args.pipes)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run(1)
    dt = run(args.steps)
    paths = W * H * per * args.pipes * args.steps
    print(f"pipes {args.pipes} spp/pipe {per} blocks/CU {os.environ.get('WAPT_BLOCKS_PER_CU', '6')}: {paths / dt / 1e6:.1f} Mpaths/s", flush=True)


if __name__ == "__main__":
    main()
