import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e:
    print("no cpu.max", e)
os.system("grep -m1 'model name' /proc/cpuinfo; cat /proc/loadavg")
import bevy_hikari_amd as hk
from oracle_lib import oracle_plugin, set_threads
settings = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
scene = hk.load_cornell(); cam = hk.cornell_camera(256, 256)
for nt in (1, 4, 8, 16, 32, 64):
    if nt > os.cpu_count(): break
    set_threads(nt)
    p = oracle_plugin(); p.set_scene(scene)
    p.render(cam, settings, frame_number=1)
    t = time.time(); p.render(cam, settings, frame_number=2); p.render(cam, settings, frame_number=3); dt = (time.time() - t) / 2
    print(f"threads {nt}: {dt*1e3:.0f} ms/frame at 256x256")
