#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X render path (BASELINE.json: "Mrays/s (whole node) at 4K,
1/2/4/8 MI355X; % FP32 VALU roofline").

A step = one pass of the hot path over one frame: rustracer's standalone scene
(standalone/src/main.rs:66-89) at 3840x2160, max_depth 5, `--spp` samples per pixel (config C2 of
SURVEY.md §8d), rendered by every rank for its interleaved row bands and gathered to rank 0 over
RCCL. Strong scaling: the frame is fixed, N GPUs split its rows. A ray = one closest-hit query
(primary or bounce segment), counted identically by the device kernel and the CPU oracle.

    python bench.py --gpus N --steps K --warmup W
        N > 1 without a launcher: bench.py starts `python -m torch.distributed.run --nproc-per-node N` itself (a CHILD
        process, before this one touches the GPU) and relays rank 0's JSON line and the exit code.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          (what the driver runs)

Rank 0 prints ONE JSON line. `roofline` is the FP32 VALU roofline of the render kernel (there is no
dense contraction => no MFMA, and the scene is a few hundred bytes => not HBM-bound): algorithmic
flops (rthip_algorithmic_flops over the counters of an instrumented run of the same seeds) / the
kernel's average duration from HIP events recorded around each launch on its stream inside the
timed region. `cpu_baseline` times the CPU oracle (oracle/, a C++ restatement of raytracer_core:
kind "port" — the Rust reference cannot be built in this image) on a bounded sample of the same frame.
`configs` (N = 1 only; extra key, the headline is unchanged) times the other configurations of SURVEY §8d at their full
sizes: C2 + point light (the literal "reflections + shadows" of BASELINE config 2), C3, C4, C5.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)" (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0

WIDTH, HEIGHT, MAX_DEPTH = 3840, 2160, 5


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per step")
    ap.add_argument("--band-rows", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config timings (C2+light, C3, C4, C5)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a one-GPU box: every rank renders on cuda:0 and the collectives run over gloo (checks the script, not the speed)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU at all: the ranks exchange synthetic rows over gloo through the same gather (checks the launch, the band split and the JSON relay)")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N > 1 and no launcher around us: start the N ranks as a child torch.distributed.run, BEFORE anything in this
    process touches the GPU (a process that has initialised HIP must never be replaced; a child is fine), relay its output."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return proc.returncode if proc.returncode != 0 or line is not None else 1


def dry_run(args, world, rank):
    """no GPU: every rank fills its owned rows with f(y) on the CPU, the rows go through the gather of bands.py over gloo,
    rank 0 checks the de-interleaved frame and prints the JSON line"""
    import torch
    import torch.distributed as dist

    from rustracer_amd.bands import RowGatherer, owned_row_indices
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    h, w = 270, 16
    rows = owned_row_indices(h, args.band_rows if world > 1 else 0, rank, world)
    local = (torch.from_numpy(rows).to(torch.uint8)[:, None, None] * torch.ones((1, w, 4), dtype=torch.uint8)).contiguous()
    t0 = time.perf_counter()
    full = local
    for _ in range(args.warmup + args.steps):
        if world > 1:
            full = RowGatherer(h, args.band_rows, rank, world, (w, 4), torch.uint8, torch.device("cpu"))(local)
    elapsed = time.perf_counter() - t0
    if rank == 0:
        want = (torch.arange(h).to(torch.uint8)[:, None, None] * torch.ones((1, w, 4), dtype=torch.uint8))
        assert torch.equal(full, want), "the gathered frame is not the band-interleaved frame"
        print(json.dumps({"metric": "Mrays/s (whole node) at 4K", "value": 0.0, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 4), "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "dry run: synthetic rows through the band gather over gloo, no GPU"}, "dry_run": True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0 and args.gpus > 1:
        sys.exit(self_launch(args))
    world = max(world, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.band_rows < 1 or args.steps < 1 or args.spp < 1:
        raise SystemExit("--band-rows, --steps and --spp must be >= 1")
    if args.dry_run:
        return dry_run(args, world, rank)

    if os.environ.get("BENCH_DUMP_AFTER"):  # debugging aid: print every thread's stack if the run is still going after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_DUMP_AFTER"]), exit=True)
    import torch
    import torch.distributed as dist

    from rustracer_amd import Raytracer, abi, flatten, presets
    from rustracer_amd.bands import BandRenderer

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

    for w in range(args.warmup):
        br.step(args.spp, MAX_DEPTH, 0, 1000 + w)  # render on this stream + gather on a side stream (overlaps the next render)
    br.wait()
    barrier()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        # HIP events on the stream the kernel is launched on (torch's current stream), recorded by step() right around the
        # launch: after its wait for the gather that last read the frame buffer, before the hand-over to the gather stream
        br.step(args.spp, MAX_DEPTH, 0, k, events=(ev0[k], ev1[k]))
    br.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / max(1, args.steps)

    # instrumented (untimed) runs of the same seeds: ray counts and the algorithmic-flop counters
    # ... twice: the reference's algorithm (every ray against every shape, ray.rs:47-67: SURVEY §8d's flops), and with
    # RTHIP_F_COUNT_EXECUTED the tests the production loops really make (primary rays only against the spheres their tile can
    # reach, none in sky-only tiles): `frac` and `frac_executed`
    rays_local, flops_local, flops_exec_local = 0, 0.0, 0.0
    lib = abi.lib()
    import ctypes as C
    for k in range(args.steps):
        for extra in (0, abi.F_COUNT_EXECUTED):
            st = abi.Stats()
            p = br.params(args.spp, MAX_DEPTH, abi.F_COUNT_STATS | extra, k)
            rc = lib.rthip_render_device(rt._ctx, C.byref(p), br.rgba.data_ptr(), None, torch.cuda.current_stream().cuda_stream, C.byref(st))
            if rc != 0:
                raise RuntimeError(lib.rthip_last_error(rt._ctx))
            if extra:
                flops_exec_local += lib.rthip_algorithmic_flops(C.byref(st))
            else:
                rays_local += st.rays
                flops_local += lib.rthip_algorithmic_flops(C.byref(st))
    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
    cnt = torch.tensor([float(rays_local), flops_local], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
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
        achieved_exec = flops_exec_local / max(1, args.steps) / (kernel_ms * 1e-3) / 1e12
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
                "kernel": "render_kernel<STATS=false, MODE_LDS, LIGHTS=false, SPH_SCENE=true>",
                "achieved": round(achieved, 3),
                "peak": FP32_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4),
                "frac_executed": round(achieved_exec / FP32_VALU_PEAK_TFLOPS, 4),
                "frac_note": "frac: flops of the reference's algorithm (every ray tested against every shape, SURVEY 8d) / kernel time / peak; "
                             "frac_executed: the same constants over the shape tests the kernel actually executes (RTHIP_F_COUNT_EXECUTED: "
                             "tests culled per 8x8 tile are not counted)",
                "issue_note": "the kernel is bound by VALU instruction issue: 89 600 wave instructions per wave at 2.48 cycles each = the launch time "
                              "(profiles/r03c_summary.md); a loop of independent v_fmac_f32 issues one per 2.5-2.6 cycles on this GPU (tools/ubench/exec_rates.hip), "
                              "not the 2.0 the 157.3 TFLOP/s peak assumes, and the time does not change between 4 and 7 resident waves per SIMD",
                "traffic": traffic,
                "kernel_ms_avg": round(kernel_ms, 4),
                "kernel_ms_slowest_rank": round(kernel_ms_max, 4),
                "flops_per_launch": flops_per_launch,
                "flops_per_ray": flops_total / max(1.0, rays_total),
            },
        }
        if args.rehearse:
            out["rehearsal"] = "all ranks on cuda:0, gloo collectives: not a measurement"
        cpu_frame = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], cpu_frame = cpu_baseline(flat)
        if world == 1 and not args.no_parity:
            out["parity"] = parity_check(rt, flat, cpu_frame)
        if world == 1 and not args.no_configs:
            rt.close()
            out["configs"] = config_lines(device, with_cpu=not args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_threads():
    """worker threads of the CPU legs: the box's CPU share (OMP_NUM_THREADS, which the GPU box sets to its per-GPU share of
    the host) when it is set, else every hardware thread"""
    import oracle
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return env if env > 0 else oracle.hw_threads()


def cpu_baseline(flat):
    """the CPU oracle on a bounded sample of the same frame (same scene, 3840x2160, depth 5)"""
    import oracle
    from rustracer_amd import make_params
    cam = oracle.camera_lookat(-1.8, 1.0, 2.0)
    threads = cpu_threads()
    spp1, sppn = 16, 16 * threads  # ~10 s each on the GPU box's host cores
    p1 = make_params(WIDTH, HEIGHT, spp1, MAX_DEPTH, 1)
    _, _, st1, s1 = oracle.render(flat, cam, p1, precision=64, threads=1)
    pn = make_params(WIDTH, HEIGHT, sppn, MAX_DEPTH, 1)
    ref, _, stn, sn = oracle.render(flat, cam, pn, precision=64, threads=threads)
    return {
        "value": round(stn["rays"] / sn / 1e6, 2),
        "unit": "Mrays/s",
        "cores": threads,
        "host_hw_threads": oracle.hw_threads(),
        "kind": "port",
        "sample": f"full {WIDTH}x{HEIGHT} frame, max_depth {MAX_DEPTH}, f64 oracle: {sppn} spp on {threads} threads ({sn:.1f} s); "
                  f"{spp1} spp on 1 thread ({s1:.1f} s) — the reference itself renders on one thread",
        "value_1thread": round(st1["rays"] / s1 / 1e6, 2),
    }, (ref, sppn)


def parity_check(rt, flat, cpu_frame=None):
    """GPU vs oracle (f64) with the mismatch histogram: on the frame the CPU baseline has just rendered — the bench's own
    3840x2160 frame at the baseline's sample count, seed 1 — or, without a baseline, on a small sample (the full matrix is in tests/)"""
    import oracle
    from parity import mismatch_histogram
    from rustracer_amd import Raytracer, make_params
    if cpu_frame is not None:
        ref, spp = cpu_frame
        w, h = WIDTH, HEIGHT
    else:
        w, h, spp = 480, 270, 8
        cam = oracle.camera_lookat(-1.8, 1.0, 2.0)
        ref, _, _, _ = oracle.render(flat, cam, make_params(w, h, spp, MAX_DEPTH, 1), precision=64)
    small = Raytracer.new(w, h, seed=1, device=0)
    res = small.render(flat, spp, max_depth=MAX_DEPTH, want_accum=False)
    hist, ok = mismatch_histogram(res.rgba8, ref)
    small.close()
    return {"vs": "oracle f64", "size": f"{w}x{h}@{spp}spp", "tolerance": "|delta|<=1/255", "fraction_within": round(ok, 6),
            "histogram": {str(i): int(v) for i, v in enumerate(hist) if v}}


def config_lines(device, with_cpu=True):
    """Every other configuration of SURVEY §8d at its full size on this GPU: kernel time (HIP events of rthip_render_device,
    best of the timed frames), Mrays/s, algorithmic flop per ray, share of the FP32 VALU peak — and for the BVH configs the
    bound they are on: node + record bytes fetched per second against HBM peak, with the measured counters of profiles/."""
    import ctypes as C

    import torch

    from rustracer_amd import Raytracer, abi, flatten, make_params, presets
    lib = abi.lib()
    configs = [
        ("C2+light", "standalone scene + one point light (5,5,5): reflections depth 5 + shadow rays (BASELINE config 2 as worded)",
         presets.standalone_scene, presets.standalone_lights, abi.F_POINT_LIGHTS, 3840, 2160, 256, 5, 3),
        ("C3", "textured scene: 9 textured spheres + textured quad + ground", presets.textured_scene, None, 0, 3840, 2160, 64, 50, 3),
        ("C4", "1,003,520-triangle mesh scene, linear BVH", presets.mesh_scene, None, 0, 3840, 2160, 16, 50, 4),
        ("C5", "10,001 spheres + one 100,000-triangle mesh, BVH, depth 8", lambda: presets.big_scene(100, 7, "uv100k"), None, 0, 7680, 4320, 64, 8, 2),
    ]
    pmc = {}
    ppath = os.path.join(ROOT, "profiles", "bvh_pmc.json")
    if os.path.exists(ppath):
        try:
            with open(ppath) as f:
                pmc = json.load(f)
        except Exception:
            pmc = {}
    lines = []
    for name, what, scene_f, lights_f, flags, w, h, spp, depth, frames in configs:
        t0 = time.perf_counter()
        flat = flatten(scene_f(), lights_f() if lights_f else ())
        rt = Raytracer.new(w, h, seed=1, device=device)
        rt.upload(flat)
        setup_s = time.perf_counter() - t0
        rt._sync_scene_camera(None)
        rgba = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream

        def run(fl):
            st = abi.Stats()
            p = make_params(w, h, spp, depth, 1, fl)
            rc = lib.rthip_render_device(rt._ctx, C.byref(p), rgba.data_ptr(), None, stream, C.byref(st))
            if rc != 0:
                raise RuntimeError(lib.rthip_last_error(rt._ctx))
            return st
        counted = run(flags | abi.F_COUNT_STATS)
        flops = lib.rthip_algorithmic_flops(C.byref(counted))
        counted_exec = run(flags | abi.F_COUNT_STATS | abi.F_COUNT_EXECUTED)
        flops_exec = lib.rthip_algorithmic_flops(C.byref(counted_exec))
        ms = min(run(flags).kernel_ms for _ in range(frames))
        tflops = flops / (ms * 1e-3) / 1e12
        line = {"config": name, "workload": f"{what}; {w}x{h}, {spp} spp, max_depth {depth}", "spheres": len(flat.spheres), "triangles": len(flat.triangles),
                "kernel_ms": round(ms, 3), "mrays_per_s": round(counted.rays / ms / 1e3, 1), "rays": counted.rays, "shadow_rays": counted.shadow_rays,
                "flops_per_ray": round(flops / max(1, counted.rays), 1),
                "roofline": {"bound": "valu_fp32", "achieved": round(tflops, 3), "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tflops / FP32_VALU_PEAK_TFLOPS, 4),
                             "frac_executed": round(flops_exec / (ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4)},
                "scene_setup_s": round(setup_s, 2)}
        if counted.node_tests:
            # bytes the traversal ASKS the memory system for: 16 B per box test (a 32-B node holds two boxes), 48 B per primitive
            # record. Nearly all of it is served by L1/L2 (profiles/: L2 hit rate 85-94 %); the HBM side is in `counters`.
            req = 16.0 * counted.node_tests + 48.0 * (counted.tri_tests + counted.sphere_tests)
            line["bvh"] = {"node_tests_per_ray": round(counted.node_tests / counted.rays, 2),
                           "prim_tests_per_ray": round((counted.tri_tests + counted.sphere_tests) / counted.rays, 2),
                           "requested_gb_per_s": round(req / (ms * 1e-3) / 1e9, 1),
                           "bound": "per-ray work at a VALU lane utilisation of 0.3-0.5: latency of dependent divergent node fetches, the CU's address "
                                    "unit and VALU issue share it; NOT HBM bandwidth (counters.hbm_fetch_gb_per_s vs hbm_peak_gb_per_s)",
                           "hbm_peak_gb_per_s": HBM_PEAK_GBS,
                           "counters": pmc.get(name)}
            if with_cpu:
                import oracle
                cam = oracle.camera_lookat(-1.8, 1.0, 2.0)
                # the CPU beside C4/C5 is a cropped window. Two numbers: the f64 oracle answering from its own
                # bounding-volume tree (oracle accel: the scan's answers bit for bit, tests/test_oracle_accel.py) on a 1024x1024
                # window, and the scan of ray.rs:47-67 itself — what the reference would do — on 32x32
                crop = 1024
                window = ((w - crop) // 2, (h - crop) // 2, crop, crop)
                _, _, ost, secs = oracle.render(flat, cam, make_params(w, h, 1, depth, 1, flags), precision=64, threads=cpu_threads(), crop=window, accel=True)
                small = 32
                swin = ((w - small) // 2, (h - small) // 2, small, small)
                _, _, sst, ssecs = oracle.render(flat, cam, make_params(w, h, 1, depth, 1, flags), precision=64, threads=cpu_threads(), crop=swin)
                line["cpu_baseline"] = {"value": round(ost["rays"] / max(secs, 1e-9) / 1e6, 3), "unit": "Mrays/s", "cores": cpu_threads(), "kind": "port",
                                        "sample": f"{crop}x{crop} window in the middle of the frame @ 1 spp, f64 oracle with its own bounding-volume tree "
                                                  f"({(ost['sphere_tests'] + ost['tri_tests']) / max(1, ost['rays']):.1f} shape tests per ray), {secs:.2f} s",
                                        "value_linear_scan": round(sst["rays"] / max(ssecs, 1e-9) / 1e6, 5),
                                        "sample_linear_scan": f"{small}x{small} window @ 1 spp, the scan of ray.rs:47-67 over all "
                                                              f"{len(flat.spheres) + len(flat.triangles)} shapes per ray (what the reference does), {ssecs:.1f} s"}
        lines.append(line)
        rt.close()
        del rgba
    return lines


if __name__ == "__main__":
    main()
