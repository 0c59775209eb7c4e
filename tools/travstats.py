import sys, os
sys.path.insert(0, os.getcwd())
import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
p = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS); p.set_scene(hk.load_cornell())
cam = hk.cornell_camera(1920, 1080)
for n in range(1, 9): p.render(cam, s, frame_number=n)
p.engine.wait(); p.engine.reset_stats()
e = p.engine
e.frame_begin(hk.frame_uniform(s, 9), cam.view_uniform(), cam.previous_view_uniform(), hk.lights_uniform())
for name, pid in (("prepass", F.PASS_PREPASS), ("direct", F.PASS_DIRECT_LIT), ("emissive", F.PASS_DIRECT_EMISSIVE), ("indirect", F.PASS_INDIRECT)):
    e.reset_stats(); e.pass_run(pid); print(name, file=sys.stderr); e.stats()
