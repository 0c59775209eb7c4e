#!/usr/bin/env python3
"""Headline benchmark: Mray/s + frame ms on Cornell 1920x1080, 2 indirect bounces, ReSTIR
(temporal + indirect spatial) on, denoise on, upscale ratio 1.0 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}] [--blocks B]

One "step" is one frame of the hot path (prepass rays -> light passes -> ReSTIR -> denoise) over
synthetic input that is resident in HBM before the timed region.  N > 1 runs one rank per GPU under
torch.distributed.run (a plain `python bench.py --gpus N` re-launches itself that way); the frame is sharded into N
horizontal bands with the halo exchanges running INSIDE the library over RCCL (hk_comm_*; include/hikari_hip.h) - if RCCL
does not come up on every rank the run FAILS, it never falls back to staging halos through the host.  Rank 0 prints ONE JSON line.

Timing: W warm-up frames, then B (default 5) timed blocks of EXACTLY K frames each, every block bracketed by
barrier + device synchronisation on both sides, MAX over ranks per block.  `value` / `ms_per_step` are the
MEDIAN block; `blocks_ms_per_step` lists all of them, `min_ms_per_step` the fastest.

Ray accounting: rays = primary rays (one per pixel) + traverse_top invocations + stand-alone
traverse_bottom invocations.  They are counted by replaying the same frames on a second
context created with HK_CTX_COUNT_RAYS (the path is deterministic, so the replay traces exactly the
rays of the timed run) - the timed region itself carries no counters.

Scenes beyond LDS (configs 3, 4; also measured briefly inside the default run: `extra_configs`): `roofline` is the TIMED trace kernel's
(k_wf_trace_wide) - every trace launch between its own HIP events (HK_TIMING_TRACE_STAGES), its walks counted by its own counting twin
on a third context (HK_CTX_COUNT_WALKS: same schedule, same walks, same bytes out - `replay_bit_identical` refers to it; the fused
ray-counting replay, which may resolve an exact tie differently, is `ray_count_replay_bit_identical`) - as algorithmic bytes / HBM
peak, record fetches / the 128-B gather roof measured in the same run, HBM-side counter bytes / HBM peak, and the fraction of the
trace time after a stage's queue first runs dry.

--config selects the other BASELINE.json configs on ONE GPU (3: Sponza-class stand-in 1080p 3 bounces;
4: city-class stand-in 4K 2 bounces; 5: Cornell 4K 8 bounces, both spatial passes, denoise off).  The
default (2) is the configuration the metric is quoted on; the default single-GPU invocation also measures configs 3 and 5
briefly after the headline (`extra_configs`, same protocol, fewer frames) and runs a sustained block of >= 3 s of headline
frames (`sustained`; `value` stays the median of the short blocks).
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


def workload(hk, config, width, height, bounces):
    """(scene, camera, settings, lights, description, default steps) of a BASELINE.json config."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    U = hk.Upscale.SMAA_TU_1_0
    if config == 2:
        W, H = width or 1920, height or 1080
        b = 2 if bounces is None else bounces
        return (hk.load_cornell(), hk.cornell_camera(W, H), hk.HikariSettings(indirect_bounces=b, upscale=U), hk.lights_uniform(),
                f"Cornell box {W}x{H}, {b} indirect bounces, ReSTIR temporal + indirect spatial reuse on, denoise on, upscale ratio 1.0, static camera")
    if config == 3:
        W, H = width or 1920, height or 1080
        b = 3 if bounces is None else bounces
        scene, sun = synthetic_large()
        return (scene, synthetic_camera(W, H, extent=9.0), hk.HikariSettings(indirect_bounces=b, upscale=U), hk.lights_uniform(directional=sun),
                f"config 3 stand-in: seeded Sponza-class scene (256 k unique triangles, 409 instances, 50 materials, 8 emitters, sun 100 000 lux) {W}x{H}, {b} bounces, denoise on")
    if config == 4:
        W, H = width or 3840, height or 2160
        b = 2 if bounces is None else bounces
        scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
        return (scene, synthetic_camera(W, H, extent=30.0), hk.HikariSettings(indirect_bounces=b, upscale=U),
                hk.lights_uniform(directional=dict(sun, illuminance=10000.0)),
                f"config 4 stand-in: seeded city-class scene (1.5 M unique triangles, 2002 instances, 1 emitter, sun 10 000 lux) {W}x{H}, {b} bounces, denoise on")
    if config == 5:
        W, H = width or 3840, height or 2160
        b = 8 if bounces is None else bounces
        return (hk.load_cornell(), hk.cornell_camera(W, H), hk.HikariSettings(indirect_bounces=b, emissive_spatial_reuse=True, denoise=False, upscale=U),
                hk.lights_uniform(), f"config 5: Cornell box {W}x{H}, {b} bounces, emissive + indirect spatial reuse on, denoise off")
    raise SystemExit(f"unknown --config {config}")


def pmc_traffic_in_run(timeout_s=150):
    """HBM-side bytes per launch of every production kernel of a config-2 frame, measured now: two child runs of this script (short,
    no probes) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... WRITE_SIZE` - one counter per pass.  Bytes = 2 x FETCH_SIZE +
    WRITE_SIZE in KB (MI355X_MICROARCH.md, HBM section: gfx950 tallies a 128-B read request as 64 B).  None on any failure."""
    import glob
    import shutil
    import signal
    import sqlite3
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    # this process is itself being profiled (rocprofv3 exports these to its application): no profiler inside a profiler
    if any(k.startswith(("ROCPROF_", "ROCP_TOOL_", "ROCPROFILER_")) for k in os.environ):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--config", "2", "--steps", "6", "--warmup", "4", "--blocks", "1", "--no-cpu-baseline", "--no-hbm-probe",
             "--no-extra-configs", "--sustained-seconds", "0", "--no-pmc"]
    per = {}
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU")):   # (memory counters one per pass; three of the SQ block's)
            with tempfile.TemporaryDirectory(dir="/tmp") as d:
                p = subprocess.Popen([rocprof, "--kernel-trace", "--pmc"] + list(counters) + ["-d", d, "--"] + child, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
                try:
                    p.wait(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    os.killpg(p.pid, signal.SIGKILL)   # (the group this call started: rocprofv3 and its child)
                    p.wait()
                    return None
                dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
                if not dbs:
                    return None
                db = sqlite3.connect(dbs[0])
                rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
                db.close()
                acc = {}
                for name, counter, value in rows:
                    if counter in counters:
                        acc.setdefault((name, counter), []).append(float(value))
                for (name, counter), vals in acc.items():
                    short = name.replace("void hkd::", "").replace("hkd::", "").split("(")[0]
                    per.setdefault(short, {})[counter] = sum(vals) / len(vals)
    except Exception:
        return None
    import re

    out, frame = {}, 0
    for short, v in per.items():
        if not short.startswith("k_") or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        out[short] = int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
        # (the frame's production kernels: the ray-counting replay's instantiations and the probes excluded, as tools/make_traffic_profile.py does)
        if not re.search(r"<(true|false), true, \d>|k_prepass<true|k_stream|k_valu|k_gather|k_copy|k_resolve|k_join|k_count", short):
            frame += out[short]
    if not out:
        return None
    sq = {short: {"valu_wave_instructions": v["SQ_INSTS_VALU"], "lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4)}
          for short, v in per.items() if short.startswith("k_") and all(k in v for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU")) and v["SQ_ACTIVE_INST_VALU"] > 0}
    return {"per_kernel": out, "frame_bytes": frame, "sq": sq,
            "source": "measured in this invocation: rocprofv3 --kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE, then SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU, on "
                      "three child runs of `bench.py --config 2 --steps 6 --warmup 4`; bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: a 128-B read request is tallied as 64 B), "
                      "lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); per-launch averages"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="frames per timed block (default 48; 12 for --config 3/4/5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up frames (default 512 for the headline config: half a second, enough for the clocks to settle "
                    "- with 16 the five timed blocks of a run still drifted 2.5 %% downwards; 16 for --config 3/4/5)")
    ap.add_argument("--blocks", type=int, default=None, help="timed blocks of --steps frames each; the median is reported (default: 5, and as many more as it "
                    "takes to time 240 frames in all - the default run's 5 x 48 - when --steps is smaller: the figure is then the median over the same "
                    "number of frames whatever --steps is)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5))
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--band-split", choices=("auto", "balanced", "equal"), default="auto",
                    help="--gpus N: bands of equal cost (geometry pixels per row, counted on frame 1 by every rank: HK_FRAME_BALANCE_BANDS) or of equal height; "
                         "auto = balanced for scenes beyond LDS (configs 3, 4: sky rows cost nothing, city rows everything - predicted 3.2x instead of 2.75x at 8 GPUs), "
                         "equal for the Cornell configs (every row costs about the same: 1.97x against 2.01x, profiles/r03_band_balance_probe.json)")
    ap.add_argument("--band-rebalance-rounds", type=int, default=6, help="--gpus N: during the warm-up the split follows MEASURED band times - this many times a frame is rendered "
                    "with HK_FRAME_TIME_BAND, the ranks all-gather their band's time (N floats) and move the boundaries with hk_rebalanced_band_bounds, the history rows that "
                    "change owner travel (hk_migrate_bands); the split then stays fixed through the timed blocks.  0 = off")
    ap.add_argument("--no-gather", action="store_true", help="--gpus N: leave every band's rows of the tone-mapped image on the GPU that rendered them (default: rank 0 "
                    "collects them every frame, HK_FRAME_GATHER - SURVEY 8e step 7)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-probe", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect the HBM-side bytes of the headline's kernels in this invocation (two child runs of a short config-2 bench "
                    "under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE after the timed region; default for the plain single-GPU config-2 run); the committed profile "
                    "under profiles/ is used instead")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--ctx-flags", type=int, default=0, help="OR-ed into the flags of the timed contexts (64 = HK_CTX_WAVEFRONT, 128 = HK_CTX_FUSED_INDIRECT, 32 = HK_CTX_EXACT_TRAVERSAL, 256 = HK_CTX_NO_WIDE_WALK)")
    ap.add_argument("--no-wide-walk", action="store_true", help="A/B: scenes beyond LDS keep the threaded skip-link walk for closest-hit rays too (HK_CTX_NO_WIDE_WALK)")
    ap.add_argument("--passes", action="store_true", help="also report a per-pass time breakdown (extra untimed frames)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the short config-3 / config-5 measurements of the default run")
    ap.add_argument("--motion", action="store_true", help="ONLY the frames under motion (tools/motion_bench.py): config 2 with an orbiting camera (two speeds), config 3 with the orbit + "
                    "instances moving through hk_refit_scene_instances - ms per frame with HK_CTX_RACING_SCATTER, in the default (race resolved) and with HK_CTX_DETERMINISTIC_SCATTER, their deviation from each other "
                    "and from the CPU oracle per frame; the default run carries a shorter version of the same in extra_configs['motion']")
    ap.add_argument("--sustained-seconds", type=float, default=3.0, help="length of the sustained block of the default run (0 = none)")
    ap.add_argument("--sustained", dest="sustained_seconds_forced", action="store_true", help="run the sustained block for any config / rank count")
    args = ap.parse_args()
    if args.no_wide_walk:
        args.ctx_flags |= 256  # HK_CTX_NO_WIDE_WALK
    if args.steps is None:
        args.steps = 48 if args.config == 2 else 12
    if args.warmup is None:
        args.warmup = 512 if args.config == 2 else 16
    if args.blocks is None:
        # the default run of the headline config times 5 blocks of 48 frames; a caller that asks for shorter blocks gets as many as time the
        # same 240 frames (each block still EXACTLY --steps frames between a barrier + device synchronisation on both sides): the frames of a
        # young history are dearer (profiles/r06_history_age.txt), and five short blocks would be a median over the first hundred of them
        args.blocks = max(5, -(-240 // max(1, args.steps))) if args.config == 2 else 5

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and rank != 0:
        os.dup2(2, 1)  # the launcher merges the ranks' stdout: only rank 0's JSON line belongs there (libraries print banners at exit)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            # plain `python bench.py --gpus N`: re-launch under torch.distributed.run, one rank per GPU (what the driver's own
            # launcher does); the ranks' exit code and rank 0's JSON line pass straight through
            if "HIKARI_BENCH_DEVICE" not in os.environ and torch.cuda.device_count() < args.gpus:
                sys.exit(f"bench.py --gpus {args.gpus}: this node exposes {torch.cuda.device_count()} GPU(s)")
            import subprocess

            # --standalone: the launcher binds a port of its own on 127.0.0.1 (a port picked here by bind / close can be taken before the
            # launcher binds it)
            cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)
        args.gpus = world
    # Test hooks (tests/test_bench_ranks.py): HIKARI_BENCH_TRANSPORT=host + HIKARI_BENCH_DEVICE=0 run every rank on ONE GPU with
    # halos staged through host memory over gloo, to exercise the multi-rank code path where a node has a single GPU (RCCL
    # refuses two ranks on one device).  The driver never sets them: N ranks = N GPUs, halos over RCCL inside the library, and
    # BandRenderer raises on every rank if RCCL does not come up everywhere (no fallback: VERDICT r02 weak 10).
    transport = os.environ.get("HIKARI_BENCH_TRANSPORT", "rccl")
    if "HIKARI_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["HIKARI_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        # the process group only carries the rendezvous (RCCL unique id), the barriers and the max-over-ranks of the timing;
        # the data path - the halo rows - moves inside libhikari_hip.so over its own RCCL communicator
        dist.init_process_group("gloo")

    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from bevy_hikari_amd.distributed import BandRenderer

    transport_used = [None]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def measure(config, steps, warmup, n_blocks, width=None, height=None, bounces=None, alone=True, passes=False, sustained_s=0.0):
        """The timed protocol on one BASELINE config: warm-up, n_blocks timed blocks of EXACTLY `steps` frames (barrier + device
        synchronisation on both sides, max over ranks), the ray count by a deterministic replay, optionally the dominant
        dispatch alone on the GPU and a sustained block.  Returns a dict; the engines stay alive in it under "_engines"."""
        scene, camera, settings, lights, description = workload(hk, config, width, height, bounces)
        W, H = camera.width, camera.height
        sc = settings.to_c()
        view, pview = camera.view_uniform(), camera.previous_view_uniform()

        def make_engine(flags):
            e = hk.Engine(device=local_rank, flags=flags)
            e.upload_noise()
            e.upload_scene(scene)
            e.resize(W, H, 1.0)
            r = None
            if world > 1:
                r = BandRenderer(e, rank, world, transport=transport)  # (fallback=None: raises on every rank without RCCL)
                transport_used[0] = r.transport
            return e, r

        bounds_log = {}   # frame number -> the split in force FROM that frame on (the timed engine's; the replays follow it)

        def run_frames(e, r, first, last, rebalance_at=(), follow=None):
            for n in range(first, last + 1):
                frame = hk.frame_uniform(settings, n)
                if r is None:
                    e.frame_render(frame, view, pview, lights, sc)
                    continue
                if follow is not None and n in follow:   # a replay: the split the timed engine took before this frame
                    r.migrate(follow[n], n, settings, W, H)
                # frame 1 splits the rows by cost (hk_balance_bands: every rank derives the same boundaries), the rest keep them
                r.render(frame, view, pview, lights, settings, W, H, balance=(n == 1 and (args.band_split == "balanced" or (args.band_split == "auto" and config in (3, 4)))),
                         gather=not args.no_gather,   # rank 0 collects the finished image, every frame, inside the timed region
                         time_band=n in rebalance_at)
                if n in rebalance_at:   # (warm-up only) one all-gather of N floats, the same controller on every rank, the moved rows migrate
                    before = list(r.bounds) if r.bounds is not None else None
                    after = r.rebalance(r.band_time_ms(), n + 1, settings, W, H, damping=0.6 if len(bounds_log) < 3 else 0.35)
                    if after is not None and after != before:
                        bounds_log[n + 1] = list(after)

        eng, rend = make_engine(args.ctx_flags)
        if warmup < 256 and config == args.config:
            # A caller that asks for few warm-up FRAMES still gets settled clocks: ~0.4 s of the VALU issue probe (register-only FMA
            # chains, hk_measure_valu) before the first frame.  No frame is added, removed or cached; with 3 warm-up frames and none of
            # this the headline reads 3 % low (1.017 instead of 0.98 ms per frame).  (What the probe cannot settle is the HISTORY: the
            # frames themselves get cheaper while the reservoirs age - frames 6-25 of the Cornell run cost 0.96 ms, 25-85 0.92, a spike when
            # every reservoir reaches max_reservoir_lifetime = 100 together, 0.90 from frame 200 on; the spatial pass alone 0.299 -> 0.265 ms:
            # old reservoirs skip their history merge - profiles/r06_history_age.txt.  --warmup 5 --steps 20 therefore reads 0.925 ms where
            # --warmup 512 reads 0.900 on the same box: both are the workload, at different ages.)
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < 0.4:
                eng.measure_valu(4096)
        rounds = args.band_rebalance_rounds if (rend is not None and warmup >= 16) else 0
        rebalance_at = tuple(sorted({max(2, (k + 1) * (warmup - 4) // (rounds + 1)) for k in range(rounds)})) if rounds > 0 else ()
        run_frames(eng, rend, 1, warmup, rebalance_at=rebalance_at)
        eng.wait()
        eng.reset_stats()
        # HIP events ON the two long dispatches of the frame (hipExtLaunchKernelGGL: no extra stream operation) + each trace launch of the
        # queue-based schedule; scenes beyond LDS (configs 3 / 4) also bracket their two direct-light dispatches on the side stream
        mask = (1 << F.PASS_INDIRECT) | (1 << F.PASS_INDIRECT_SPATIAL_REUSE) | (1 << F.TIMING_TRACE_STAGES)
        if config in (3, 4):
            mask |= (1 << F.PASS_DIRECT_LIT) | (1 << F.PASS_DIRECT_EMISSIVE)
        blocks = []
        n0 = warmup
        for k in range(max(1, n_blocks)):
            # The events ride on the FIRST timed block only.  An event on a dispatch is a completion signal the host can observe, and the
            # frame pays for it (presumably the wider release such a dispatch ends with) - measured in one run: 0.942 ms per frame with the
            # two long dispatches instrumented against 0.904 without (round 6; rounds 4 / 5 saw 0.945 against 0.928).  The roofline figures
            # come from that block (K frames of the timed region, HIP events on the dispatches' own stream), `value` is the MEDIAN block -
            # with the default five blocks an un-instrumented one; blocks_ms_per_step lists all of them, the instrumented one first.
            eng.set_timing_mask(mask if k == 0 else 0)
            barrier()
            t0 = time.perf_counter()
            run_frames(eng, rend, n0 + 1, n0 + steps)
            eng.wait()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            barrier()
            blocks.append(max_over_ranks(t1 - t0))
            n0 += steps
        last_frame = n0
        elapsed = float(np.median(blocks))
        st = eng.stats()
        schedule = eng.indirect_schedule()
        traversal = eng.traversal_mode() + (eng.wide_walk(),)
        ind_ms = st.pass_ms_total[F.PASS_INDIRECT] / max(1, st.pass_launches[F.PASS_INDIRECT])
        ind_launches = int(st.pass_launches[F.PASS_INDIRECT])
        sp_ms = st.pass_ms_total[F.PASS_INDIRECT_SPATIAL_REUSE] / max(1, st.pass_launches[F.PASS_INDIRECT_SPATIAL_REUSE])
        direct_ms = {F.PASS_NAMES[k]: round(st.pass_ms_total[k] / max(1, st.pass_launches[k]), 5) for k in (F.PASS_DIRECT_LIT, F.PASS_DIRECT_EMISSIVE) if st.pass_launches[k]}
        trace_launches = int(st.pass_launches[F.TIMING_TRACE_STAGES])
        trace_ms = st.pass_ms_total[F.TIMING_TRACE_STAGES] / max(1, trace_launches)   # average TRACE launch of the queue-based indirect pass (0 launches: fused schedule)
        eng.set_timing_mask(0)
        tone = eng.read(F.BUF_TONE_MAPPED)

        # a sustained block of the same frames (>= sustained_s seconds, one barrier-bracketed region): long enough for the
        # driver's GPU-activity sampling and for the clock to settle under load; reported beside `value`, never as `value`
        sustained = None
        if sustained_s > 0:
            n_frames = max(steps, int(1.1 * sustained_s / (elapsed / steps)) + 1)  # (+10 %: frames run a little faster in one long block)
            barrier()
            t0 = time.perf_counter()
            run_frames(eng, rend, last_frame + 1, last_frame + n_frames)
            eng.wait()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            barrier()
            sustained = {"frames": n_frames, "seconds": round(max_over_ranks(t1 - t0), 3)}
            sustained["ms_per_step"] = round(sustained["seconds"] / n_frames * 1e3, 4)

        # frame LATENCY: one frame, wait, repeat (the timed blocks keep three streams full - frame n's a-trous levels run beside frame
        # n + 1's light passes - so ms_per_step is a THROUGHPUT figure; this is what one frame takes from its first dispatch to its image).
        # After everything that is compared with the replay: these frames are not part of any count.
        lat, nl = [], last_frame + (sustained["frames"] if sustained else 0)
        for k in range(min(steps, 16)):
            barrier()
            t0 = time.perf_counter()
            run_frames(eng, rend, nl + 1, nl + 1)
            eng.wait()
            lat.append(max_over_ranks(time.perf_counter() - t0))
            nl += 1
        frame_latency_ms = float(np.median(lat)) * 1e3

        # ray count by deterministic replay (one block's worth of frames: the camera is static and the rays per frame are
        # counted over the LAST timed block)
        ceng, crend = make_engine(F.CTX_COUNT_RAYS | (args.ctx_flags & (F.CTX_EXACT_TRAVERSAL | F.CTX_NO_WIDE_WALK)))  # (the primary rays of the replay walk what the timed ones do)
        run_frames(ceng, crend, 1, last_frame - steps, follow=bounds_log)
        ceng.wait()
        ceng.reset_stats()
        run_frames(ceng, crend, last_frame - steps + 1, last_frame, follow=bounds_log)
        cst = ceng.stats()
        traced = float(cst.rays_tlas + cst.rays_blas)
        if dist is not None:
            t = torch.tensor([traced], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            traced = float(t.item())
        # primary rays: one per pixel per frame (apron rows ray-cast redundantly by neighbouring bands are not counted)
        total_rays = traced + float(W) * H * steps
        same = bool((ceng.read(F.BUF_TONE_MAPPED) == tone).all())  # the replay must reproduce the timed frames bit for bit
        # what the walks of those frames did (HkStats walk_*: node steps, triangle tests, instance entries, closest hits) - per frame,
        # and for the indirect pass alone (its dispatch repeated once on the last frame's inputs: same rays, same walks)
        walk = {"per_frame": {k: getattr(cst, "walk_" + k) / steps for k in ("node_steps", "triangle_tests", "instance_entries", "closest_hits", "top_node_steps")},
                "rays_per_frame": (cst.rays_tlas + cst.rays_blas + cst.rays_primary) / steps}
        if crend is None:
            ceng.reset_stats()
            ceng.pass_run(F.PASS_INDIRECT)
            ist = ceng.stats()
            walk["indirect_pass"] = {k: float(getattr(ist, "walk_" + k)) for k in ("node_steps", "triangle_tests", "instance_entries", "closest_hits", "top_node_steps")}
            walk["indirect_pass"]["rays"] = float(ist.rays_tlas + ist.rays_blas)
        del ceng, crend
        # Scenes beyond LDS, product default: what the TIMED trace kernel's walks do, counted by its own counting twin (HK_CTX_COUNT_WALKS:
        # same schedule, same walks, same bytes out) - the same frames once more, then the indirect pass alone on the last frame's inputs
        if schedule == "wavefront" and traversal[2] and rend is None:
            import ctypes as C

            weng, _ = make_engine(F.CTX_COUNT_WALKS | args.ctx_flags)
            run_frames(weng, None, 1, last_frame)
            weng.wait()
            same_w = bool((weng.read(F.BUF_TONE_MAPPED) == tone).all())
            weng.pass_run(F.PASS_INDIRECT)
            raw = np.zeros(64 * 32, dtype=np.uint64)
            weng.api.call("debug_read_wf_timeline", weng.ctx, raw.ctypes.data_as(C.POINTER(C.c_uint64)), raw.size)
            raw = raw.reshape(64, 32)
            inv = np.uint64(0xFFFFFFFFFFFFFFFF)
            names = ("records", "records_in_the_instance_tree", "triangle_tests", "instance_entries", "rays", "any_hit_rays", "closest_hits_found", "pieces_handed_to_idle_lanes")
            stages = []
            for sidx in range(settings.indirect_bounces + 1):
                r = raw[sidx]
                if r[4] == 0:
                    continue
                t0, tdry, tend = int(inv - r[0]), int(inv - r[1]), int(r[2])
                stages.append({"stage": sidx, "ticks": tend - t0, "ticks_after_the_queue_ran_dry": tend - max(t0, min(tdry, tend)),
                               "mean_wave_residency": round(int(r[3]) / int(r[4]) / max(1, tend - t0), 3), **{k: int(r[8 + j]) for j, k in enumerate(names)}})
            # the counting twin runs the timed kernels themselves: ITS frames are what "the replay reproduces the timed frames" is asked of
            # (the ray-counting replay above walks the fused, direction-threaded kernels - the only form HK_CTX_COUNT_RAYS exists in -
            # whose closest hits may fall differently where two candidates tie exactly: a pixel or two per 4K frame)
            same_rays, same = same, same_w
            walk["ray_count_replay_bit_identical"] = same_rays
            walk["trace_kernel"] = {"stages": stages, "replay_bit_identical": same_w, **{k: sum(st_[k] for st_ in stages) for k in names},
                                    "tail_fraction_of_trace_time": round(sum(st_["ticks_after_the_queue_ran_dry"] for st_ in stages) / max(1, sum(st_["ticks"] for st_ in stages)), 4)}
            del weng

        res = {"config": config, "description": description, "W": W, "H": H, "steps": steps, "warmup": warmup, "blocks": blocks, "elapsed": elapsed,
               "band_bounds": (rend.bounds if rend is not None else None), "band_bounds_history": {str(k): v for k, v in bounds_log.items()},
               "last_frame": last_frame, "schedule": schedule, "traversal": traversal, "ind_ms": ind_ms, "ind_launches": ind_launches, "sp_ms": sp_ms, "direct_ms": direct_ms,
               "frame_latency_ms": frame_latency_ms, "trace_ms": trace_ms, "trace_launches": trace_launches,
               "total_rays": total_rays, "same": same,
               "walk": walk, "sustained": sustained, "scene": scene, "settings": settings, "lights": lights, "view": view, "pview": pview, "sc": sc,
               "band_rows": H if rend is None else (rend.band(H)[1] - rend.band(H)[0])}
        if sustained:
            sustained["value"] = round(total_rays / steps * sustained["frames"] / sustained["seconds"] / 1e6, 3)
            sustained["unit"] = "Mray/s"
        if alone:
            # the dominant kernel ALONE on the GPU: same frames on a single-stream context (in the timed run the two
            # direct-light dispatches share the GPU with it from a second stream, which stretches its own duration)
            xeng, xrend = make_engine(F.CTX_SINGLE_STREAM | args.ctx_flags)
            run_frames(xeng, xrend, 1, warmup, follow=bounds_log)
            xeng.wait()
            xeng.reset_stats()
            xeng.set_timing_mask((1 << F.PASS_INDIRECT) | (1 << F.PASS_INDIRECT_SPATIAL_REUSE))
            run_frames(xeng, xrend, warmup + 1, warmup + min(steps, 16))
            xst = xeng.stats()
            res["ind_ms_alone"] = xst.pass_ms_total[F.PASS_INDIRECT] / max(1, xst.pass_launches[F.PASS_INDIRECT])
            res["spatial_ms_alone"] = xst.pass_ms_total[F.PASS_INDIRECT_SPATIAL_REUSE] / max(1, xst.pass_launches[F.PASS_INDIRECT_SPATIAL_REUSE])
            if passes and world == 1:   # per-pass times with every dispatch alone on the GPU
                xeng.reset_stats()
                xeng.set_timing_mask(0xFFFF)
                run_frames(xeng, xrend, warmup + min(steps, 16) + 1, warmup + min(steps, 16) + 6)
                ps = xeng.stats()
                res["passes"] = {F.PASS_NAMES[i]: round(ps.pass_ms_total[i] / 6.0, 4) for i in range(F.PASS_COUNT) if ps.pass_launches[i]}
                xeng.set_timing_mask(0)
            res["_probe_engine"] = xeng
        res["_engines"] = (eng, rend)
        return res

    def motion_lines(short):
        """frames under motion + the price of determinism: tools/motion_bench.py"""
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import motion_bench
        from oracle_lib import oracle_engine   # (the checker, compared against - never timed)

        res = {"orbit_radians_per_frame": motion_bench.ORBIT_RAD_PER_FRAME, "note": motion_bench.__doc__.split("\n\n")[1].replace("\n", " ")}
        kw = dict(device=local_rank, blocks=3, warmup=16 if short else 24, compare_frames=6 if short else 12, oracle_engine=oracle_engine)
        res["2"] = motion_bench.run(hk, F, 2, frames=24 if short else 48, oracle_frames=2 if short else 6, **kw)
        if not short:
            res["2_fast_orbit"] = motion_bench.run(hk, F, 2, frames=48, oracle_frames=6, orbit=10.0 * motion_bench.ORBIT_RAD_PER_FRAME, **kw)
        res["3"] = motion_bench.run(hk, F, 3, frames=6 if short else 12, oracle_frames=0, **kw)
        return res

    if args.motion:
        if world != 1:
            sys.exit("bench.py --motion is a single-GPU measurement")
        print(json.dumps({"metric": "frame ms under motion: HK_CTX_RACING_SCATTER vs the default (race resolved, light form) vs HK_CTX_DETERMINISTIC_SCATTER", "unit": "ms", "n_gpus": 1, "higher_is_better": False, "motion": motion_lines(False)}), flush=True)
        return

    # ------------------------------------------------------------------ timed run (headline)
    default_run = world == 1 and args.config == 2 and args.width is None and args.height is None and args.bounces is None and not args.ctx_flags
    m = measure(args.config, args.steps, args.warmup, args.blocks, args.width, args.height, args.bounces, alone=True, passes=args.passes,
                sustained_s=args.sustained_seconds if default_run or args.sustained_seconds_forced else 0.0)
    W, H, blocks, elapsed, last_frame, schedule = m["W"], m["H"], m["blocks"], m["elapsed"], m["last_frame"], m["schedule"]
    ind_ms, ind_ms_alone, total_rays, same, description = m["ind_ms"], m["ind_ms_alone"], m["total_rays"], m["same"], m["description"]
    passes = m.get("passes")
    xeng = m["_probe_engine"]

    # ------------------------------------------------------------------ the other single-GPU configs, briefly, in the same invocation
    def walk_roofline(x, probe_engine):
        """SURVEY 8d for scenes beyond LDS.  Dominant kernel: k_wf_trace_wide, the trace launches of the queue-based indirect pass
        (bounces + 1 per pass) - timed by HIP events around EVERY trace launch of the timed frames (HK_TIMING_TRACE_STAGES), its walks
        counted by its own counting twin (HK_CTX_COUNT_WALKS: records of 128 B, triangle tests of 48 B, instance entries of 208 B, 96 B
        per closest hit).  Three fractions, none of which can exceed 1: algorithmic bytes against the HBM peak (small by nature: a
        walk moves little), record fetches against the rate at which the chip serves dependent divergent 128-B gathers out of its L2s
        (measured in this run at the kernel's own occupancy: no chain of dependent record fetches runs faster), and the HBM-side
        bytes of the committed counter passes against the HBM peak.  The tail fraction says how much of the trace time passes after
        the stage's queue ran dry (the longest walks finishing): the kernel is latency- and tail-bound, not bandwidth-bound."""
        tk = x["walk"].get("trace_kernel")
        if not tk or x["trace_launches"] == 0 or x["trace_ms"] <= 0:
            return None
        scene = x["scene"]
        per_pass = x["settings"].indirect_bounces + 1
        t = x["trace_ms"] * 1e-3 * per_pass                       # trace time of one pass
        bvh_bytes = tk["records"] * 128 + tk["triangle_tests"] * 48 + tk["instance_entries"] * 208 + tk["closest_hits_found"] * 96
        wide_bytes = (len(scene.asset_nodes) + len(scene.instance_nodes)) * 128 + len(scene.primitives) * 48
        r = {"kernel": "k_wf_trace_wide: the %d trace launches of the queue-based indirect pass (closest-hit + any-hit rays of a bounce in one launch)" % per_pass,
             "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
             "avg_launch_ms": round(x["trace_ms"], 4), "launches_per_pass": per_pass, "trace_ms_per_pass": round(t * 1e3, 4), "launches_timed": x["trace_launches"],
             "pass_ms": round(x["ind_ms"], 4),
             "counted_by": "HK_CTX_COUNT_WALKS: the counting twin of the timed kernel on the same frames (its frames bit-identical to the timed ones: %s)" % tk["replay_bit_identical"],
             "walk_counts_per_pass": {k: tk[k] for k in ("records", "records_in_the_instance_tree", "triangle_tests", "instance_entries", "rays", "any_hit_rays", "closest_hits_found",
                                                         "pieces_handed_to_idle_lanes")},
             "per_ray": {"records": round(tk["records"] / max(1, tk["rays"]), 2), "triangle_tests": round(tk["triangle_tests"] / max(1, tk["rays"]), 2),
                         "instance_entries": round(tk["instance_entries"] / max(1, tk["rays"]), 2), "bvh_bytes": round(bvh_bytes / max(1, tk["rays"]), 1)},
             # (cache-served: the walk's algorithmic bytes mostly never reach HBM - so their ratio to the HBM peak is NOT an HBM roofline fraction;
             # `frac` below is the HBM-side counter bytes over the peak, when the committed counter passes exist)
             "algorithmic_bytes_per_pass": bvh_bytes, "achieved": round(bvh_bytes / t / 1e9, 2), "algorithmic_bytes_over_hbm_peak": round(bvh_bytes / t / 1e9 / HBM_PEAK_GBS, 5), "frac": None,
             "formula": "records fetched x 128 + triangle tests x 48 + instance entries x 208 + 96 per closest hit found (SURVEY 8d's terms, in the units of THIS walk)",
             "tail": {"fraction_of_trace_time_after_the_queue_ran_dry": tk["tail_fraction_of_trace_time"],
                      "per_stage": [round(st_["ticks_after_the_queue_ran_dry"] / max(1, st_["ticks"]), 3) for st_ in tk["stages"]],
                      "mean_wave_residency_per_stage": [st_["mean_wave_residency"] for st_ in tk["stages"]],
                      "note": "from the counting twin's stamps (first wave in, queue first seen dry, last wave out)"},
             "wide_tree_bytes": wide_bytes, "traffic": None}
        # HBM-side bytes of the pass's trace launches by the PMC counters (tools/pmc_fetch.sh: separate FETCH_SIZE / WRITE_SIZE passes of
        # `bench.py --config N`; not measured in this run): lower = x 1 per 64-B request (right for gathers), upper = x 2
        try:
            tfile = next(q for q in (os.path.join(ROOT, "profiles", f"r0{k}_walk_hbm_traffic.json") for k in (6, 5, 4)) if os.path.exists(q))
            with open(tfile) as f:
                tw = json.load(f).get(f"config{x.get('config', 0)}_trace_stages")
            if tw:
                r["traffic"] = tw["hbm_bytes_per_pass_lower"]
                r["frac"] = round(tw["hbm_bytes_per_pass_lower"] / t / 1e9 / HBM_PEAK_GBS, 5)
                r["frac_is"] = "HBM-side bytes of the pass's trace launches (FETCH_SIZE + WRITE_SIZE, committed counter passes) / trace time / HBM peak"
                r["hbm_side"] = {"bytes_per_pass": tw["hbm_bytes_per_pass_lower"], "bytes_per_pass_if_every_request_were_128_B": tw["hbm_bytes_per_pass_upper"],
                                 "frac_of_peak": round(tw["hbm_bytes_per_pass_lower"] / t / 1e9 / HBM_PEAK_GBS, 5),
                                 "frac_of_peak_upper": round(tw["hbm_bytes_per_pass_upper"] / t / 1e9 / HBM_PEAK_GBS, 5),
                                 "ratio_to_algorithmic": round(tw["hbm_bytes_per_pass_lower"] / bvh_bytes, 3),
                                 "source": os.path.relpath(tfile, ROOT) + " (FETCH_SIZE + WRITE_SIZE of the pass's k_wf_trace_wide launches; separate PMC passes of the same command, not this run)"}
        except (StopIteration, OSError, ValueError, KeyError, TypeError, IndexError):
            pass
        if probe_engine is not None:
            # the roofs of dependent divergent 128-B record fetches, measured now, at the trace kernel's occupancy (HK_WF_WIDE_WAVES = 5):
            # table in L1 / in the L2s / as large as the scene's wide trees (random over all of it: a walk WITHOUT locality)
            roofs = {}
            for name, fp in (("l1_resident_16KiB", 16 << 10), ("l2_resident_1MiB", 1 << 20), ("random_over_the_wide_trees", max(wide_bytes, 1 << 20))):
                gl, gb = probe_engine.measure_gather(fp, 128, 5, 256)
                roofs[name] = {"g_lane_records_s": round(gl / 8 * 64, 2), "gbytes_s": round(gb, 1), "footprint_bytes": fp}
            rec_s = tk["records"] / t / 1e9
            r["record_fetches"] = {"achieved_g_records_s": round(rec_s, 3), "roofs_measured": roofs,
                                   "frac_of_l2_resident_roof": round(rec_s / roofs["l2_resident_1MiB"]["g_lane_records_s"], 4) if roofs["l2_resident_1MiB"]["g_lane_records_s"] > 0 else None,
                                   "frac_of_random_gather_over_the_trees": round(rec_s / roofs["random_over_the_wide_trees"]["g_lane_records_s"], 4) if roofs["random_over_the_wide_trees"]["g_lane_records_s"] > 0 else None,
                                   "note": "hk_measure_gather, 128 B per dependent step, 5 waves per SIMD, every lane its own chain: the L2-resident rate is the roof of ANY chain "
                                           "of dependent record fetches short of L1 hits (a walk's upper levels do hit the L2s, which is why the random-over-the-trees figure is not a roof)"}
        return r

    extra = None
    if default_run and not args.no_extra_configs:
        extra = {}
        # The headline's timed engine is done: free it before the other configs' engines are created.  A HIP process has few hardware queues
        # per stream priority; with the headline's engines still alive the extra configs' high-priority chains shared one and config 3 read
        # 6.4 ms here against 6.15 ms by itself (DESIGN 4 "Two streams").  The probe engine stays (one stream: HK_CTX_SINGLE_STREAM).
        import gc

        m.pop("_engines", None)
        gc.collect()
        for cfg, steps_x in ((3, 8), (4, 6), (5, 8)):
            x = measure(cfg, steps_x, 6, 3, alone=False)
            extra[str(cfg)] = {"workload": x["description"], "value": round(x["total_rays"] / x["elapsed"] / 1e6, 3), "unit": "Mray/s",
                               "ms_per_step": round(x["elapsed"] / steps_x * 1e3, 4), "steps": steps_x, "warmup": 6,
                               "blocks_ms_per_step": [round(b / steps_x * 1e3, 4) for b in x["blocks"]], "rays_per_frame": round(x["total_rays"] / steps_x, 1),
                               "indirect_schedule": x["schedule"], "traversal": x["traversal"][0], "wide_walk": x["traversal"][2], "indirect_avg_launch_ms": round(x["ind_ms"], 5),
                               "trace_avg_launch_ms": round(x["trace_ms"], 5), "replay_bit_identical": x["same"]}
            if "ray_count_replay_bit_identical" in x["walk"]:   # (scenes beyond LDS: `replay_bit_identical` is the counting twin of the timed kernels; this is the fused ray-counting replay)
                extra[str(cfg)]["ray_count_replay_bit_identical"] = x["walk"]["ray_count_replay_bit_identical"]
            if cfg in (3, 4) and rank == 0 and not args.no_hbm_probe:
                extra[str(cfg)]["roofline"] = walk_roofline(x, xeng)
            if cfg in (3, 4) and x["direct_ms"]:
                # the two direct-light dispatches (side stream, beside the trace stages; HIP events around them in the timed frames).  Config 4's
                # sun pass is as heavy as all trace launches of the frame together; its lanes: profiles/*_lanes_config4.txt
                lanes = {}
                try:
                    lf = next(q for q in (os.path.join(ROOT, "profiles", f"r0{k}_final_lanes_config{cfg}.txt") for k in (6, 5)) if os.path.exists(q))
                    for line in open(lf):   # "<kernel signature, cut>  util 0.500 valu  593.7 M  waves 129600" (tools/pmc_lanes.sh)
                        if "k_direct_lit<" in line and ", false, 0>" in line and " util " in line:
                            lanes[line.split("hkd::")[1].split("(")[0]] = float(line.split(" util ")[1].split()[0])
                    lanes["source"] = os.path.relpath(lf, ROOT) + " (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), committed counter pass)"
                except (StopIteration, OSError, ValueError, IndexError):
                    pass
                extra[str(cfg)]["direct_passes"] = {"avg_launch_ms_in_run": x["direct_ms"], "lane_utilisation": lanes or None,
                                                    "note": "k_direct_lit<false, false, 0> (sun) / <true, false, 0> (emissive): any-hit shadow rays that keep their occluder - the "
                                                            "reference's own visit order, skip-link walk per lane"}
            del x

    # ------------------------------------------------------------------ empirical HBM ceiling, same run
    hbm = None
    valu = None
    if rank == 0 and not args.no_hbm_probe:
        copy_gbs, triad_gbs = xeng.measure_hbm(1 << 30, 8)
        hbm = {"copy_gbs": round(copy_gbs, 1), "triad_gbs": round(triad_gbs, 1), "bytes_per_array": 1 << 30, "guide_copy_gbs": 6290.0,
               "note": "float4 copy (2 x 1 GiB per pass) / triad (3 x 1 GiB per pass), best of three access shapes (grid-stride loop; one-shot with 4 accesses per "
                       "lane; one-shot with ONE access per lane - the shape that reaches the ceiling, tools/ubench.hip / profiles/r03_ubench.json), 8 passes each, "
                       "HIP events; guide_copy_gbs = MI355X_MICROARCH.md's measured float4 copy"}
        # the other roof: wave64 VALU instructions per second the chip issues, measured in this run (register-only v_fma_f32 chains)
        valu = {str(k) + "_waves_per_simd": round(v, 1) for k, v in xeng.measure_valu(2048).items()}

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    band_rows = m["band_rows"]
    algo_bytes = INDIRECT_BYTES_PER_PIXEL * W * band_rows
    achieved = algo_bytes / (ind_ms * 1e-3) / 1e9 if ind_ms > 0 else 0.0
    ms_blocks = [round(b / args.steps * 1e3, 4) for b in blocks]

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
        "data": "synthetic (assets/models/cornell.glb geometry or seeded generator, reference blue-noise tiles, zeroed reservoirs)",
        "config": {
            "workload": description,
            "baseline_config": args.config,
            "frames": f"warmup 1..{args.warmup}, then {len(blocks)} timed blocks of {args.steps} frames ({args.warmup + 1}..{last_frame}); value = median block; "
                      "the roofline's HIP events ride on the first block",
            "parallelism": f"band{world}" if world > 1 else "single",
            **({"band_split": ("measured" if m["band_bounds_history"] else ("balanced" if m["band_bounds"] else "equal")), "band_bounds": m["band_bounds"],
                "band_rebalancing": {"rounds_during_warmup": args.band_rebalance_rounds, "splits_taken": m["band_bounds_history"],
                                     "how": "HK_FRAME_TIME_BAND frame -> all-gather of N floats -> hk_rebalanced_band_bounds -> hk_migrate_bands; fixed through the timed blocks"},
                "gather": "none" if args.no_gather else "rank 0 collects the tone-mapped image every frame (HK_FRAME_GATHER), inside the timed region"} if world > 1 else {}),
            # hk_traversal_mode: "one-level" = one BVH over all triangles in the instances' shared local space (the Cornell box),
            # "threaded" = two-level walk over 8 direction-ordered flattenings (scenes beyond LDS), "reference" = the reference's order;
            # wide_walk: the closest-hit walks (primary rays, trace stages) take 128-B records of four grandchildren, nearest first
            "traversal": {"mode": m["traversal"][0], "orderings": m["traversal"][1], "wide_walk": m["traversal"][2]},
        },
        "blocks_ms_per_step": ms_blocks,
        "blocks": {"count": len(ms_blocks), "frames_each": args.steps, "first_timed_frame": args.warmup + 1, "last_timed_frame": args.warmup + len(ms_blocks) * args.steps,
                   "note": "every block is EXACTLY --steps frames between a barrier + device synchronisation on both sides; ms_per_step / value = the MEDIAN block.  Without "
                           "--blocks the run times 240 frames in all whatever --steps is (5 x 48 by default): the workload gets cheaper while its reservoirs age "
                           "(frames 6-25 of a fresh context 0.96 ms, 0.90 from frame 200 on; profiles/r06_history_age.txt), so the blocks are listed in frame order"},
        "instrumented_block": {"index": 0, "ms_per_step": ms_blocks[0], "note": "the HIP events of `roofline` (on the two long dispatches; configs 3 / 4: + every trace launch and the "
                               "direct-light dispatches) ride on this timed block only - they cost the frames that carry them 2-4 %; `value` is the median block"},
        "min_ms_per_step": min(ms_blocks),
        "mray_per_s_per_gpu": round(total_rays / elapsed / 1e6 / world, 3),
        "rays_per_frame": round(total_rays / args.steps, 1),
        "replay_bit_identical": same,
        **({"ray_count_replay_bit_identical": m["walk"]["ray_count_replay_bit_identical"]} if "ray_count_replay_bit_identical" in m["walk"] else {}),
        "frame_latency_ms": round(m["frame_latency_ms"], 4),
        "ms_per_step_is": "throughput: frames enqueued back to back, frame n's a-trous levels (third stream) beside frame n + 1's light passes; frame_latency_ms = one frame, wait, repeat",
        "roofline": None,
    }
    # The dominant kernel is the LONGER of the frame's two long dispatches AS TIMED IN THIS RUN (HIP events on both dispatches inside the
    # timed region: in the pipelined frame their durations stretch differently - rocprofv3 of round 5: k_spatial_reuse 521 us against
    # k_indirect 331 us - while alone on the GPU they tie).  `second_kernel` is the other one, same arithmetic.
    sp_ms, sp_ms_alone = m["sp_ms"], m.get("spatial_ms_alone") or 0.0
    sp_bytes = 240 * W * band_rows   # G-buffer 40 + own reservoir 64 + previous spatial 64 read, reservoir 64 + render 8 written (the gathered neighbour records are re-reads)

    def kernel_roofline(name, ms, ms_alone, bytes_per_launch, launches):
        ach = bytes_per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"kernel": name, "bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": round(ms, 5), "launches": launches,
                # the same kernel with nothing else on the GPU (HK_CTX_SINGLE_STREAM replay of the same frames): in the timed run the other
                # streams' dispatches share the GPU with it - the frame gets shorter, this dispatch's own duration longer
                "alone": {"avg_launch_ms": round(ms_alone, 5), "achieved": round(bytes_per_launch / (ms_alone * 1e-3) / 1e9, 3) if ms_alone > 0 else 0.0,
                          "frac": round(bytes_per_launch / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if ms_alone > 0 else 0.0}}

    k_ind = kernel_roofline("k_indirect<true, false, 2> (indirect_lit_ambient, light.wgsl:1263-1498)" if schedule == "fused" else
                            "indirect_lit_ambient (light.wgsl:1263-1498) as k_wf_setup + k_wf_trace / k_wf_shade per bounce + k_wf_final: first dispatch start to last dispatch end",
                            ind_ms, ind_ms_alone, algo_bytes, m["ind_launches"])
    k_ind["schedule"] = schedule
    k_sp = kernel_roofline("k_spatial_reuse<false, %s> (spatial_reuse, light.wgsl:1503-1684)" % ("true" if ((W + 15) // 16) * ((band_rows + 15) // 16) >= 16384 else "false"), sp_ms, sp_ms_alone, sp_bytes, m["ind_launches"]) if sp_ms > 0 else None
    spatial_dominant = bool(k_sp and schedule == "fused" and sp_ms > ind_ms)
    out["roofline"] = dict(k_sp if spatial_dominant else k_ind)
    out["roofline"]["dominant_by"] = "the longer average launch of the two long dispatches, HIP events in the timed region: k_indirect %.4f ms, k_spatial_reuse %.4f ms" % (ind_ms, sp_ms)
    if k_sp:
        out["roofline"]["second_kernel"] = k_ind if spatial_dominant else k_sp
    achieved = out["roofline"]["achieved"]
    tpath = next((q for q in (os.path.join(ROOT, "profiles", f"r0{k}_indirect_hbm_traffic.json") for k in (6, 5, 4, 3, 2)) if os.path.exists(q)), "")
    rf_ind = out["roofline"]["second_kernel"] if spatial_dominant else out["roofline"]              # where k_indirect's / k_spatial_reuse's figures live in the line
    rf_sp = out["roofline"] if spatial_dominant else out["roofline"].get("second_kernel")
    if tpath and rf_sp is not None and world == 1 and args.config == 2:   # counter traffic from the committed PMC passes (replaced below by this invocation's own when it measures them)
        t2 = json.load(open(tpath)).get("second_kernel")
        if t2:
            rf_sp["traffic"] = t2["hbm_bytes_per_launch"]
            rf_sp["traffic_ratio_to_algorithmic"] = t2["ratio_to_algorithmic"]
            if "ratio_to_algorithmic_lower_bound" in t2:   # FETCH_SIZE x 1 (right for 64-B record gathers) .. x 2 (right for coalesced streams): profiles/r04_fetch_calibration.json
                rf_sp["traffic_ratio_bounds"] = [t2["ratio_to_algorithmic_lower_bound"], t2["ratio_to_algorithmic"]]
    if hbm:
        out["roofline"]["hbm_ceiling_measured"] = hbm
        out["roofline"]["frac_of_measured_copy"] = round(achieved / hbm["copy_gbs"], 6) if hbm["copy_gbs"] > 0 else None
        # the whole frame against the same ceiling: SURVEY 8d's 1.70 KB/px of compulsory traffic (config 2's pass list)
        if args.config == 2:
            frame_bytes = 1700.0 * W * H
            out["frame_roofline"] = {"algorithmic_bytes_per_frame": frame_bytes, "achieved_gbs": round(frame_bytes / (elapsed / args.steps) / 1e9, 1),
                                     "frac_of_peak": round(frame_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                     "note": "SURVEY 8d's 1.70 KB/px charged to all pixels; the counter figure below is what the frame really moves"}
            try:  # what the frame's kernels move by the PMC counters (committed passes of this command): the honest figure
                fr_prof = json.load(open(tpath)).get("frame") if tpath else None
                if fr_prof and world == 1:
                    cb = float(fr_prof["hbm_bytes_per_frame"])
                    out["frame_roofline"]["counter"] = {"hbm_bytes_per_frame": cb, "achieved_gbs": round(cb / (elapsed / args.steps) / 1e9, 1),
                                                        "frac_of_peak": round(cb / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                                        "frac_of_measured_copy": round(cb / (elapsed / args.steps) / 1e9 / hbm["copy_gbs"], 4) if hbm["copy_gbs"] > 0 else None,
                                                        "source": os.path.relpath(tpath, ROOT) + " (sum of 2 x FETCH_SIZE + WRITE_SIZE over the frame's kernels; not measured in this run)"}
            except Exception:
                pass
    if valu:
        # VALU issue: the peak is MEASURED in this run (hk_measure_valu).  MI355X_MICROARCH.md: a wave64 VALU instruction issues
        # over 2 cycles on a SIMD-32, i.e. 256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1 229 G wave-instructions/s nominal; the probe
        # reaches ~0.96 T/s with 8 waves per SIMD and ~0.88 T/s with the 4 waves per SIMD k_indirect runs at (128 VGPRs) - one
        # wave alone issues only one instruction per ~6 cycles.  (Round 2 priced this against one instruction per 4 cycles = 614 G/s;
        # that was wrong.)
        vi = {"peak_measured_ginstr_s": valu, "peak_nominal_ginstr_s": round(256 * 4 * 2.4e9 / 2 / 1e9, 1),
              "peak_source": "hk_measure_valu in this run; nominal = MI355X_MICROARCH.md (wave64 VALU: 2 cycles on a SIMD-32, 2.4 GHz)"}
        if tpath and world == 1 and args.config == 2:
            # the kernel's VALU wave-instructions per launch come from the committed SQ PMC pass of this command (counters cannot be
            # read inside this process); its launch time is this run's
            try:
                prof = json.load(open(tpath))
                n_valu = float(prof["limiter"]["valu_wave_instructions"])
                alone_ms = ind_ms_alone or ind_ms
                rate = n_valu / (alone_ms * 1e-3) / 1e9
                vi.update({"wave_instructions_per_launch": n_valu, "achieved_ginstr_s": round(rate, 1),
                           "frac_of_measured_peak_8_waves": round(rate / valu["8_waves_per_simd"], 4),
                           "frac_of_measured_peak_4_waves": round(rate / valu["4_waves_per_simd"], 4),
                           "frac_of_nominal": round(rate / (256 * 4 * 2.4 / 2), 4),
                           "lane_utilisation": prof["limiter"].get("lane_utilisation"),
                           "source": "SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) of " + os.path.relpath(tpath, ROOT) +
                                     " (not measured in this run); launch time of this run, kernel alone"})
            except Exception:
                pass
        rf_ind["valu_issue"] = vi   # (k_indirect's: its VALU wave-instructions against the issue ceiling measured in this run)
    if tpath and world == 1 and args.config == 2:
        out["traffic_profile"] = {"source": os.path.relpath(tpath, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run)"}
        try:  # separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction (MI355X_MICROARCH.md, HBM section), per launch
            rf_ind["traffic"] = json.load(open(tpath)).get("hbm_bytes_per_launch")
            rf_ind["traffic_source"] = os.path.relpath(tpath, ROOT)
        except Exception:
            pass
    if default_run and not args.no_pmc and not args.no_extra_configs and not args.no_hbm_probe:   # (the full default invocation only: every tool that profiles a short run passes one of these)
        # ... and MEASURED IN THIS INVOCATION where rocprofv3 is at hand: after the timed region, two child runs of a
        # short config-2 bench under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (one counter per pass, as the micro-arch
        # guide prescribes; every failure falls back to the committed profile above)
        live = pmc_traffic_in_run()
        if live:
            k1 = next((v for k, v in live["per_kernel"].items() if k.startswith("k_indirect<true, false, 2>")), None)
            k2 = next((v for k, v in live["per_kernel"].items() if k.startswith("k_spatial_reuse<false, false>")), None)
            if k1:
                rf_ind["traffic"] = k1
                rf_ind["traffic_source"] = live["source"]
                rf_ind["traffic_ratio_to_algorithmic"] = round(k1 / algo_bytes, 3)
            if k2 and rf_sp is not None:
                rf_sp["traffic"] = k2
                rf_sp["traffic_ratio_to_algorithmic"] = round(k2 / (240.0 * W * band_rows), 3)
                rf_sp["traffic_source"] = live["source"]
                rf_sp.pop("traffic_ratio_bounds", None)
            if "frame_roofline" in out and live["frame_bytes"] > 0:
                cb = float(live["frame_bytes"])
                out["frame_roofline"]["counter"] = {"hbm_bytes_per_frame": cb, "achieved_gbs": round(cb / (elapsed / args.steps) / 1e9, 1),
                                                    "frac_of_peak": round(cb / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                                    "frac_of_measured_copy": round(cb / (elapsed / args.steps) / 1e9 / hbm["copy_gbs"], 4) if hbm and hbm["copy_gbs"] > 0 else None,
                                                    "per_kernel": live["per_kernel"], "source": live["source"]}
            out["traffic_profile"] = {"source": live["source"]}
            q1 = next((v for k, v in live["sq"].items() if k.startswith("k_indirect<true, false, 2>")), None)
            if q1 and valu and "valu_issue" in rf_ind:   # the kernel's VALU wave-instructions and lane utilisation, of this invocation too
                alone_ms = ind_ms_alone or ind_ms
                rate = q1["valu_wave_instructions"] / (alone_ms * 1e-3) / 1e9
                rf_ind["valu_issue"].update({"wave_instructions_per_launch": q1["valu_wave_instructions"], "achieved_ginstr_s": round(rate, 1),
                                                      "frac_of_measured_peak_8_waves": round(rate / valu["8_waves_per_simd"], 4),
                                                      "frac_of_measured_peak_4_waves": round(rate / valu["4_waves_per_simd"], 4),
                                                      "frac_of_nominal": round(rate / (256 * 4 * 2.4 / 2), 4), "lane_utilisation": q1["lane_utilisation"],
                                                      "source": live["source"] + "; launch time of this run, kernel alone"})
            q2 = next((v for k, v in live["sq"].items() if k.startswith("k_spatial_reuse<false, false>")), None)
            if q2 and valu and rf_sp is not None:
                alone_ms = sp_ms_alone or sp_ms
                rate = q2["valu_wave_instructions"] / (alone_ms * 1e-3) / 1e9
                rf_sp["valu_issue"] = {"wave_instructions_per_launch": q2["valu_wave_instructions"], "achieved_ginstr_s": round(rate, 1),
                                       "frac_of_measured_peak_4_waves": round(rate / valu["4_waves_per_simd"], 4), "lane_utilisation": q2["lane_utilisation"],
                                       "source": live["source"] + "; launch time of this run, kernel alone"}
    if args.config in (3, 4) and world == 1 and not args.no_hbm_probe:
        out["roofline"]["bvh_walk"] = walk_roofline(m, xeng)
    if m.get("direct_ms"):
        out["direct_passes_in_run_ms"] = m["direct_ms"]
    if m["sustained"]:
        out["sustained"] = m["sustained"]
    if extra is not None and rank == 0:
        try:
            extra["motion"] = motion_lines(True)
        except Exception as e:   # (a measurement beside the headline: never the reason the line is missing)
            extra["motion"] = {"error": repr(e)}
    if extra:
        out["extra_configs"] = extra
    if transport_used[0]:
        out["config"]["halo_transport"] = transport_used[0]
    if passes:
        out["pass_ms"] = passes

    # ------------------------------------------------------------------ CPU baseline (oracle = port of the reference WGSL)
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import oracle_engine, set_threads

        cores = cgroup_cpus()
        set_threads(cores)
        scene, settings, lights, view, pview, sc = m["scene"], m["settings"], m["lights"], m["view"], m["pview"], m["sc"]
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
    # the JSON line is the LAST thing on stdout: what C libraries (RCCL's start-up banner ...) left in their stdio buffers goes out
    # first, and whatever they print later goes to stderr
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
