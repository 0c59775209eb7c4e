#!/usr/bin/env python3
"""Headline benchmark: Mray/s + frame ms on Cornell 1920x1080, 2 indirect bounces, ReSTIR
(temporal + indirect spatial) on, denoise on, upscale ratio 1.0 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" is one frame of the hot path (prepass rays -> light passes -> ReSTIR -> denoise) over
synthetic input that is resident in HBM before the timed region.  N > 1 is launched by
torch.distributed.run, one rank per GPU; the frame is sharded into N horizontal bands with two
RCCL halo exchanges per frame (bevy-hikari_amd/distributed.py).  Rank 0 prints ONE JSON line.

Ray accounting: rays = primary rays (one per pixel) + traverse_top invocations + stand-alone
traverse_bottom invocations.  They are counted by replaying the same frames on a second
context created with HK_CTX_COUNT_RAYS (the path is deterministic, so the replay traces exactly the
rays of the timed run) - the timed region itself carries no counters.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic (compulsory, unique) bytes per pixel of the dominant kernel indirect_lit_ambient
# (SURVEY 8d): reads G-buffer 44 + noise 4 + previous reservoir 64, writes reservoir 64 +
# variance 4 + render 8.
INDIRECT_BYTES_PER_PIXEL = (44 + 4 + 64) + (64 + 4 + 8)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def cgroup_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--passes", action="store_true", help="also report a per-pass time breakdown (extra untimed frames)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # Test hooks (tests/test_bench_ranks.py): HIKARI_BENCH_BACKEND=gloo + HIKARI_BENCH_DEVICE=0 run every rank on ONE GPU with
    # halos staged through host memory, to exercise the multi-rank code path where a node has a single GPU (RCCL refuses
    # two ranks on one device).  The driver never sets them: N ranks = N GPUs over RCCL.
    backend = os.environ.get("HIKARI_BENCH_BACKEND", "nccl")
    if "HIKARI_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["HIKARI_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        torch.cuda.set_stream(torch.cuda.Stream())  # a real stream handle (the default stream's is 0)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from bevy_hikari_amd.distributed import BandRenderer

    W, H = args.width, args.height
    settings = hk.HikariSettings(indirect_bounces=args.bounces, upscale=hk.Upscale.SMAA_TU_1_0)  # others = defaults
    sc = settings.to_c()
    scene = hk.load_cornell()
    camera = hk.cornell_camera(W, H)
    view, pview, lights = camera.view_uniform(), camera.previous_view_uniform(), hk.lights_uniform()

    def make_engine(flags):
        e = hk.Engine(device=local_rank, flags=flags)
        e.upload_noise()
        e.upload_scene(scene)
        e.resize(W, H, 1.0)
        r = None
        if world > 1:
            # run the library on the (non-default) torch stream RCCL orders itself against
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            e.on_host_stream = torch.cuda.current_stream().cuda_stream != 0
            r = BandRenderer(e, rank, world)
        return e, r

    def run_frames(e, r, first, last):
        for n in range(first, last + 1):
            frame = hk.frame_uniform(settings, n)
            if r is None:
                e.frame_render(frame, view, pview, lights, sc)
            else:
                r.render(frame, view, pview, lights, settings, W, H)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ timed run
    eng, rend = make_engine(0)
    run_frames(eng, rend, 1, args.warmup)
    eng.wait()
    eng.reset_stats()
    eng.set_timing_mask(1 << F.PASS_INDIRECT)  # HIP events around the dominant kernel only
    barrier()
    t0 = time.perf_counter()
    run_frames(eng, rend, args.warmup + 1, args.warmup + args.steps)
    eng.wait()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = eng.stats()
    ind_ms = st.pass_ms_total[F.PASS_INDIRECT] / max(1, st.pass_launches[F.PASS_INDIRECT])
    eng.set_timing_mask(0)

    # ------------------------------------------------------------------ ray count by deterministic replay
    ceng, crend = make_engine(F.CTX_COUNT_RAYS)
    run_frames(ceng, crend, 1, args.warmup)
    ceng.wait()
    ceng.reset_stats()
    run_frames(ceng, crend, args.warmup + 1, args.warmup + args.steps)
    cst = ceng.stats()
    # the dominant kernel ALONE on the GPU: same frames on a single-stream context (in the timed run the two
    # direct-light dispatches share the GPU with it from a second stream, which stretches its own duration)
    xeng, xrend = make_engine(F.CTX_SINGLE_STREAM)
    run_frames(xeng, xrend, 1, args.warmup)
    xeng.wait()
    xeng.reset_stats()
    xeng.set_timing_mask(1 << F.PASS_INDIRECT)
    run_frames(xeng, xrend, args.warmup + 1, args.warmup + min(args.steps, 16))
    xst = xeng.stats()
    ind_ms_alone = xst.pass_ms_total[F.PASS_INDIRECT] / max(1, xst.pass_launches[F.PASS_INDIRECT])
    traced = np.array([cst.rays_tlas + cst.rays_blas], dtype=np.float64)
    if dist is not None:
        t = torch.tensor(traced, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        traced = t.cpu().numpy()
    # primary rays: one per pixel per frame (apron rows ray-cast redundantly by neighbouring bands are not counted)
    total_rays = float(traced[0]) + float(W) * H * args.steps
    # the replay must reproduce the timed frames bit for bit
    same = bool((ceng.read(F.BUF_TONE_MAPPED) == eng.read(F.BUF_TONE_MAPPED)).all())

    passes = None
    if args.passes and world == 1:   # per-pass times with every dispatch alone on the GPU
        xeng.reset_stats()
        xeng.set_timing_mask(0xFFFF)
        run_frames(xeng, xrend, args.warmup + min(args.steps, 16) + 1, args.warmup + min(args.steps, 16) + 6)
        ps = xeng.stats()
        passes = {F.PASS_NAMES[i]: round(ps.pass_ms_total[i] / 6.0, 4) for i in range(F.PASS_COUNT) if ps.pass_launches[i]}
        xeng.set_timing_mask(0)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    band_rows = H if rend is None else (rend.band(H)[1] - rend.band(H)[0])
    algo_bytes = INDIRECT_BYTES_PER_PIXEL * W * band_rows
    achieved = algo_bytes / (ind_ms * 1e-3) / 1e9 if ind_ms > 0 else 0.0
    traffic, limiter = None, None
    tpath = os.path.join(ROOT, "profiles", "r01_indirect_hbm_traffic.json")
    if os.path.exists(tpath) and world == 1 and (W, H) == (1920, 1080):
        try:
            prof = json.load(open(tpath))
            traffic, limiter = prof.get("hbm_bytes_per_launch"), prof.get("limiter")  # PMC passes of the same command, see profiles/
        except Exception:
            traffic = None

    out = {
        "metric": "Mray/s (Cornell 1080p 2-bounce, whole job) + frame ms",
        "value": round(total_rays / elapsed / 1e6, 3),
        "unit": "Mray/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (assets/models/cornell.glb geometry, reference blue-noise tiles, zeroed reservoirs)",
        "config": {
            "workload": f"Cornell box {W}x{H}, {args.bounces} indirect bounces, ReSTIR temporal + indirect spatial reuse on, denoise on, upscale ratio 1.0, static camera",
            "frames": f"warmup 1..{args.warmup}, timed {args.warmup + 1}..{args.warmup + args.steps}",
            "parallelism": f"band{world}" if world > 1 else "single",
        },
        "mray_per_s_per_gpu": round(total_rays / elapsed / 1e6 / world, 3),
        "rays_per_frame": round(total_rays / args.steps, 1),
        "replay_bit_identical": same,
        "roofline": {
            "kernel": "k_indirect<MULTIPLE_BOUNCES> (indirect_lit_ambient, light.wgsl:1263-1498)",
            "bound": "hbm",
            "achieved": round(achieved, 3),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6),
            "traffic": traffic,
            "algorithmic_bytes_per_launch": algo_bytes,
            "avg_launch_ms": round(ind_ms, 5),
            "launches": int(st.pass_launches[F.PASS_INDIRECT]),
            # the same kernel with nothing else on the GPU (HK_CTX_SINGLE_STREAM replay of the same frames)
            "alone": {"avg_launch_ms": round(ind_ms_alone, 5), "achieved": round(algo_bytes / (ind_ms_alone * 1e-3) / 1e9, 3) if ind_ms_alone > 0 else 0.0,
                      "frac": round(algo_bytes / (ind_ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if ind_ms_alone > 0 else 0.0},
        },
    }
    if limiter:
        out["roofline"]["limiter"] = limiter
    if passes:
        out["pass_ms"] = passes

    # ------------------------------------------------------------------ CPU baseline (oracle = port of the reference WGSL)
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import oracle_engine, set_threads

        cores = cgroup_cpus()
        set_threads(cores)
        o = oracle_engine()
        o.upload_noise()
        o.upload_scene(scene)
        o.resize(W, H, 1.0)
        c0 = time.perf_counter()
        done = 0
        while done < 2 or (time.perf_counter() - c0 < args.cpu_seconds and done < args.warmup + args.steps):
            done += 1
            o.frame_render(hk.frame_uniform(settings, done), view, pview, lights, sc)
        cdt = time.perf_counter() - c0
        os_ = o.stats()
        crays = os_.rays_primary + os_.rays_tlas + os_.rays_blas
        out["cpu_baseline"] = {
            "value": round(crays / cdt / 1e6, 4),
            "unit": "Mray/s",
            "cores": cores,
            "kind": "port",
            "sample": f"frames 1..{done} of the same {W}x{H} workload ({cdt:.1f} s, OpenMP, {cores} threads); CPU restatement of the reference WGSL - "
                      "the reference itself (wgpu + lavapipe) cannot be built or run here",
            "ms_per_frame": round(cdt / done * 1e3, 1),
        }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
