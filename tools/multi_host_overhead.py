#!/usr/bin/env python3
"""Host time of hk_multi_frame_render (one process driving n bands; here all on device 0): the call's return-to-return
time with nothing waited for, serial enqueue (hk_debug_multi_serial, one thread walks the bands) against one enqueue thread per
band.  A band's GPU time at 8 GPUs is ~0.3 ms (profiles/r03_final_band_probe.json): the host must stay below that.

    python tools/multi_host_overhead.py [bands ...]      -> JSON
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(bands):
    import bevy_hikari_amd as hk
    from bevy_hikari_amd.distributed import MultiEngine

    W, H = 1920, 1080
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    sc = s.to_c()
    cam = hk.cornell_camera(W, H)
    view, pview, lights = cam.view_uniform(), cam.previous_view_uniform(), hk.lights_uniform()
    out = {}
    for n in bands:
        m = MultiEngine([0] * n)
        m.upload_noise(); m.upload_scene(hk.load_cornell()); m.resize(W, H, 1.0)
        for k in range(1, 17):
            m.frame_render(hk.frame_uniform(s, k), view, pview, lights, sc)
        m.wait()
        N = 200
        frames = [hk.frame_uniform(s, k) for k in range(17, 17 + N)]
        t0 = time.perf_counter()
        for f in frames:
            m.frame_render(f, view, pview, lights, sc)
        t1 = time.perf_counter()
        m.wait()
        t2 = time.perf_counter()
        out[str(n)] = {"enqueue_ms_per_frame": round(1e3 * (t1 - t0) / N, 4), "total_ms_per_frame": round(1e3 * (t2 - t0) / N, 4)}
        m.close()
    return out


if __name__ == "__main__":
    if os.environ.get("HK_MULTI_PROBE_CHILD"):
        if os.environ.get("HK_MULTI_PROBE_SERIAL"):
            import bevy_hikari_amd as _hk
            _hk.api().call("debug_multi_serial", 1)   # hikari_hip_debug.h
        print(json.dumps(measure([int(a) for a in sys.argv[1:]])))
        sys.exit(0)
    bands = sys.argv[1:] or ["2", "4", "8"]
    res = {}
    for label, env in (("one_thread_per_band", {}), ("serial", {"HK_MULTI_PROBE_SERIAL": "1"})):
        r = subprocess.run([sys.executable, __file__] + bands, env=dict(os.environ, HK_MULTI_PROBE_CHILD="1", **env), capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr[-2000:])
            sys.exit(1)
        res[label] = json.loads(r.stdout.strip().splitlines()[-1])
    res["note"] = "hk_multi_frame_render, Cornell 1920x1080 2 bounces, every band on device 0 (so total_ms is n bands sharing ONE GPU); enqueue = host time per call"
    print(json.dumps(res, indent=1))
