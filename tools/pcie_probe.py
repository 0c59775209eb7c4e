#!/usr/bin/env python3
"""Frame time with the output left in HBM vs read back over PCIe after every frame ("What crosses PCIe")."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
p = hk.HikariPlugin(device=0)
p.set_scene(hk.load_cornell())
s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
cam = hk.cornell_camera(1920, 1080)
for n in range(1, 17):
    p.render(cam, s, frame_number=n)
p.engine.wait()
t0 = time.perf_counter()
for n in range(17, 65):
    p.render(cam, s, frame_number=n)
p.engine.wait()
t_res = (time.perf_counter() - t0) / 48
t0 = time.perf_counter()
for n in range(65, 113):
    p.render(cam, s, frame_number=n)
    img = p.engine.read(F.BUF_TONE_MAPPED)
t_rb = (time.perf_counter() - t0) / 48
print(json.dumps({"ms_per_frame_resident": round(t_res * 1e3, 3), "ms_per_frame_with_readback_of_tone_mapped": round(t_rb * 1e3, 3), "readback_bytes": int(img.nbytes)}))
