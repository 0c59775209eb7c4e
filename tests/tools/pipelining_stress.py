"""Stress of the primary-ray pipelining (context.hip stage TEMPORAL): random sequences of frames, refits, whole-scene uploads, anti-aliased
frames and reads on a context that pipelines against one that never does - every buffer that is read must agree.
    python tests/tools/pipelining_stress.py <first seed> <last seed>"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import diff_buffers, snapshot, product_default_traversal
from bevy_hikari_amd.scenes import synthetic_large
scene, sun = synthetic_large(0x5EED0003, 40, 40, 80, 400, 50, 8, 12.0)
lights = hk.lights_uniform(directional=sun)
rest = [np.ctypeslib.as_array(i.model).copy() for i in scene.instances]
setter = scene.builder.api.raw("scene_builder_set_instance_transform")
bad_runs = 0
total_pipelined = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    with product_default_traversal():
        a, b = hk.HikariPlugin(device=0), hk.HikariPlugin(device=0)
    a.engine.set_debug_option(F.DEBUG_OPT_PREPASS_PIPELINE, 1)
    b.engine.set_debug_option(F.DEBUG_OPT_PREPASS_PIPELINE, 0)
    for p in (a, b): p.set_scene(scene)
    size = [(320, 180), (640, 360), (197, 111)][seed % 3]
    first = None
    next_read = int(rng.integers(2, 7))
    for n in range(1, 41):
        s = hk.HikariSettings(indirect_bounces=int(rng.choice([2, 2, 3])) if n % 9 == 0 else 2, upscale=hk.Upscale.SMAA_TU_1_0)
        cam = hk.Camera(hk.look_at_transform((1.6 * 9.0 + 0.03 * n * (seed % 2), 1.1 * 9.0, 2.0 * 9.0), (0.0, 0.6, 0.0)), *size)
        r = rng.random()
        if r < 0.25:      # a refit between two frames
            for k in rng.choice(len(rest), size=11, replace=False):
                m = rest[k].reshape(4, 4).T.copy(); m[0, 3] += 0.02 * n; m[2, 3] -= 0.01 * n
                t = m.T.astype(np.float32).reshape(-1)
                setter(scene.builder.h, int(k), t.ctypes.data_as(C.POINTER(F.f32)))
            for p in (a, b): p.engine.refit_instances(scene.builder)
        elif r < 0.30:    # the whole scene again
            for p in (a, b): p.engine.upload_scene(scene)
        aa = rng.random() < 0.1
        for p in (a, b): p.render(cam, s, lights=lights, frame_number=n, antialias=bool(aa))
        next_read -= 1
        if next_read > 0: continue
        next_read = int(rng.integers(1, 7))
        bad = {k: v[:50] for k, v in diff_buffers(snapshot(a), snapshot(b)).items() if k not in ("reservoir4", "reservoir5")}
        if bad and first is None: first = (n, sorted(bad)[:6])
    total_pipelined += a.engine.prepasses_pipelined()
    if first:
        bad_runs += 1
        print("seed", seed, "MISMATCH", first)
print("runs", int(sys.argv[2]) - int(sys.argv[1]), "mismatching", bad_runs, "frames pipelined", total_pipelined)
