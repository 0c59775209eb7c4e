#!/usr/bin/env python3
"""Does the GPU overlap the tail of one cohort's trace stage with the bulk of another's?  The cheapest way to ask: render the frame
as N bands ON ONE DEVICE (hk_multi with repeated device ids: every band has its own context, streams and persistent launches) and
compare the frame time with the single context's.  Bands pay aprons and exchanges; if the frame still gets faster, staggered
cohorts are worth building inside one context.    python tools/cohort_probe.py [3 4]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bevy_hikari_amd as hk  # noqa: E402
from bevy_hikari_amd import _ffi as F  # noqa: E402
from bevy_hikari_amd.distributed import MultiEngine  # noqa: E402
from bench import workload  # noqa: E402


def main():
    out = {}
    for cfg in [int(a) for a in sys.argv[1:]] or [3, 4]:
        scene, camera, settings, lights, description = workload(hk, cfg, None, None, None)
        W, H = camera.width, camera.height
        view, pview, sc = camera.view_uniform(), camera.previous_view_uniform(), settings.to_c()
        rows = {}
        for bands in (1, 2, 3, 4):
            m = MultiEngine([0] * bands)
            m.upload_noise(); m.upload_scene(scene); m.resize(W, H, 1.0)
            n = 0

            def frames(k, flags=0):
                nonlocal n
                for _ in range(k):
                    n += 1
                    m.frame_render(hk.frame_uniform(settings, n), view, pview, lights, sc, flags)
                m.wait()

            frames(1, F.FRAME_BALANCE_BANDS if bands > 1 else 0)
            frames(6)
            t0 = time.perf_counter()
            frames(8)
            rows[bands] = round((time.perf_counter() - t0) / 8 * 1e3, 3)
            m.close()
        out[str(cfg)] = {"workload": description, "frame_ms_by_bands_on_one_device": rows}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
