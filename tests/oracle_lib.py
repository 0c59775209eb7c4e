"""Loads the CPU oracle (oracle/_build/libhikari_oracle.so) behind the same driver classes the
product uses.  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; nothing under bevy-hikari_amd/ does."""
import ctypes as C
import os

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libhikari_oracle.so")
_API = None


def oracle_api():
    global _API
    if _API is None:
        _API = F.Api(ORACLE_LIB, "orc_")
        for name, argtypes in {
            "orc_set_threads": [C.c_int],
            "orc_kat_intersects_aabb": [C.POINTER(F.f32)] * 4 + [C.POINTER(F.f32)],
            "orc_kat_intersects_triangle": [C.POINTER(F.f32)] * 4,
            "orc_kat_reservoir_roundtrip": [C.c_void_p, C.c_void_p],
            "orc_kat_normal_basis": [C.POINTER(F.f32), C.POINTER(F.f32)],
            "orc_kat_hash": [F.u32, C.POINTER(F.u32), C.POINTER(F.f32)],
            "orc_kat_trace": [C.c_void_p, C.POINTER(F.f32), C.POINTER(F.f32), F.f32, F.f32, F.u32, C.POINTER(F.u32), C.POINTER(F.u32),
                              C.POINTER(F.f32), C.POINTER(F.f32)],
            "orc_frame_stage_rows": [C.c_void_p, F.u32, C.POINTER(F.HkSettings), F.u32, F.u32, F.u32],
        }.items():
            fn = getattr(_API.dll, name)
            fn.argtypes, fn.restype = argtypes, C.c_int
        _API.dll.orc_set_threads(default_threads())
    return _API


def default_threads():
    """Usable cores: the affinity mask capped by the cgroup CPU quota (the GPU box shows 256 CPUs
    but grants 16; running 256 OpenMP threads there is 50x slower than 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def oracle_engine():
    return hk.Engine(api=oracle_api())


def oracle_plugin():
    return hk.HikariPlugin(api=oracle_api())


def set_threads(n):
    return oracle_api().dll.orc_set_threads(n)
