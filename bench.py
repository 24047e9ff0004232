#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X render path (BASELINE.json: "Mrays/s (whole node) at 4K,
1/2/4/8 MI355X; % FP32 VALU roofline").

A step = one pass of the hot path over one frame: rustracer's standalone scene
(standalone/src/main.rs:66-89) at 3840x2160, max_depth 5, `--spp` samples per pixel (config C2 of
SURVEY.md §8d), rendered by every rank for its interleaved row bands and gathered to rank 0 over
RCCL. Strong scaling: the frame is fixed, N GPUs split its rows. A ray = one closest-hit query
(primary or bounce segment), counted identically by the device kernel and the CPU oracle.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. `roofline` is the FP32 VALU roofline of the render kernel (there is no
dense contraction => no MFMA, and the scene is a few hundred bytes => not HBM-bound): algorithmic
flops (rthip_algorithmic_flops over the counters of an instrumented run of the same seeds) / the
kernel's average duration from HIP events recorded around each launch on its stream inside the
timed region. `cpu_baseline` times the CPU oracle (oracle/, a C++ restatement of raytracer_core:
kind "port" — the Rust reference cannot be built in this image) on a bounded sample of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)" (v_pk_fma_f32 rate, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz)

WIDTH, HEIGHT, MAX_DEPTH = 3840, 2160, 5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per step")
    ap.add_argument("--band-rows", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a one-GPU box: every rank renders on cuda:0 and the collectives run over gloo (checks the script, not the speed)")
    args = ap.parse_args()

    if os.environ.get("BENCH_DUMP_AFTER"):  # debugging aid: print every thread's stack if the run is still going after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_DUMP_AFTER"]), exit=True)
    import torch
    import torch.distributed as dist

    from rustracer_amd import Raytracer, abi, flatten, presets
    from rustracer_amd.bands import BandRenderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    device = 0 if args.rehearse else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))

    flat = flatten(presets.standalone_scene())
    rt = Raytracer.new(WIDTH, HEIGHT, seed=1, device=device)
    rt.upload(flat)
    br = BandRenderer(rt, rank, world, args.band_rows, gather_on_host=args.rehearse)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(seed):
        return br.step(args.spp, MAX_DEPTH, 0, seed)  # render on this stream + gather on a side stream (overlaps the next render)

    for w in range(args.warmup):
        step(1000 + w)
    barrier()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev0[k].record()  # HIP events on the stream the kernel is launched on (torch's current stream)
        img = br.step(args.spp, MAX_DEPTH, 0, k)
        ev1[k].record()
    br.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / max(1, args.steps)

    # instrumented (untimed) runs of the same seeds: ray counts and the algorithmic-flop counters
    rays_local, flops_local = 0, 0.0
    lib = abi.lib()
    import ctypes as C
    for k in range(args.steps):
        st = abi.Stats()
        p = br.params(args.spp, MAX_DEPTH, abi.F_COUNT_STATS, k)
        rc = lib.rthip_render_device(rt._ctx, C.byref(p), br.rgba.data_ptr(), None, torch.cuda.current_stream().cuda_stream, C.byref(st))
        if rc != 0:
            raise RuntimeError(lib.rthip_last_error(rt._ctx))
        rays_local += st.rays
        flops_local += lib.rthip_algorithmic_flops(C.byref(st))
    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
    cnt = torch.tensor([float(rays_local), flops_local], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed, kernel_ms_max = float(t[0]), float(t[1])
    rays_total, flops_total = float(cnt[0]), float(cnt[1])

    if rank == 0:
        mrays = rays_total / elapsed / 1e6
        # roofline of the dominant kernel (render_kernel), this rank's launches: flops per launch / avg launch duration
        flops_per_launch = flops_local / max(1, args.steps)
        achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("spp") == args.spp:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (whole node) at 4K",
            "value": round(mrays, 1),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"C2: rustracer standalone scene (4 spheres: dielectric, lambertian ground r=100, metal, texture) "
                            f"{WIDTH}x{HEIGHT}, max_depth {MAX_DEPTH}, {args.spp} spp per step, RGBA8 frame gathered to rank 0",
                "width": WIDTH, "height": HEIGHT, "spp": args.spp, "max_depth": MAX_DEPTH,
                "parallelism": f"row bands of {args.band_rows} interleaved over {world} GPU(s)" if world > 1 else "single GPU, whole frame",
                "rays_per_step": rays_total / max(1, args.steps),
            },
            "roofline": {
                "bound": "valu_fp32",
                "kernel": "render_kernel<STATS=false, MODE_LDS, LIGHTS=false>",
                "achieved": round(achieved, 3),
                "peak": FP32_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4),
                "traffic": traffic,
                "kernel_ms_avg": round(kernel_ms, 4),
                "flops_per_launch": flops_per_launch,
                "flops_per_ray": flops_total / max(1.0, rays_total),
            },
        }
        if args.rehearse:
            out["rehearsal"] = "all ranks on cuda:0, gloo collectives: not a measurement"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(flat)
        if world == 1 and not args.no_parity:
            out["parity"] = parity_check(rt, flat)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(flat):
    """the CPU oracle on a bounded sample of the same frame (same scene, 3840x2160, depth 5)"""
    import oracle
    from rustracer_amd import make_params
    cam = oracle.camera_lookat(-1.8, 1.0, 2.0)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, oracle.hw_threads())
    spp1, sppn = 16, 16 * threads  # ~10 s each on the GPU box's host cores
    p1 = make_params(WIDTH, HEIGHT, spp1, MAX_DEPTH, 1)
    _, _, st1, s1 = oracle.render(flat, cam, p1, precision=64, threads=1)
    pn = make_params(WIDTH, HEIGHT, sppn, MAX_DEPTH, 1)
    _, _, stn, sn = oracle.render(flat, cam, pn, precision=64, threads=threads)
    return {
        "value": round(stn["rays"] / sn / 1e6, 2),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"full {WIDTH}x{HEIGHT} frame, max_depth {MAX_DEPTH}, f64 oracle: {sppn} spp on {threads} threads ({sn:.1f} s); "
                  f"{spp1} spp on 1 thread ({s1:.1f} s) — the reference itself renders on one thread",
        "value_1thread": round(st1["rays"] / s1 / 1e6, 2),
    }


def parity_check(rt, flat):
    """small GPU-vs-oracle (f64) parity sample with the mismatch histogram (the full matrix is in tests/)"""
    import oracle
    from parity import mismatch_histogram
    from rustracer_amd import Raytracer, make_params
    w, h, spp = 480, 270, 8
    small = Raytracer.new(w, h, seed=1, device=0)
    res = small.render(flat, spp, max_depth=MAX_DEPTH, want_accum=False)
    cam = oracle.camera_lookat(-1.8, 1.0, 2.0)
    ref, _, _, _ = oracle.render(flat, cam, make_params(w, h, spp, MAX_DEPTH, 1), precision=64)
    hist, ok = mismatch_histogram(res.rgba8, ref)
    small.close()
    return {"vs": "oracle f64", "size": f"{w}x{h}@{spp}spp", "tolerance": "|delta|<=1/255", "fraction_within": round(ok, 6),
            "histogram": {str(i): int(v) for i, v in enumerate(hist) if v}}


if __name__ == "__main__":
    main()
