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


class OracleApi:
    """The oracle's entry points behind the interface of bevy_hikari_amd._ffi.Api (call / raw / _fns / last_error / abi_version):
    the oracle exports the frame-path part of the C ABI (F._SIGNATURES) with the prefix `orc_`, plus orc_debug_math."""

    prefix = "orc_"

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found - build it first (make -C oracle)")
        self.path = path
        self.dll = C.CDLL(path, mode=C.RTLD_GLOBAL)
        self._fns = {}
        table = dict(F._SIGNATURES)
        table["debug_math"] = F._DEBUG["debug_math"]
        for name, argtypes in table.items():
            fn = getattr(self.dll, "orc_" + name)
            fn.argtypes, fn.restype = argtypes, C.c_int
            self._fns[name] = fn
        self.dll.orc_destroy.argtypes, self.dll.orc_destroy.restype = [C.c_void_p], None
        self._fns["destroy"] = self.dll.orc_destroy
        self.dll.orc_last_error.restype = C.c_char_p
        self.dll.orc_abi_version.restype = F.u32

    def abi_version(self):
        return int(self.dll.orc_abi_version())

    def last_error(self):
        msg = self.dll.orc_last_error()
        return msg.decode() if msg else ""

    def raw(self, name):
        return self._fns[name]

    def call(self, name, *args):
        rc = self._fns[name](*args)
        if rc is not None and rc != F.HK_OK:
            raise F.HikariError(rc, "orc_" + name, self.last_error())
        return rc


def oracle_api():
    global _API
    if _API is None:
        _API = OracleApi(ORACLE_LIB)
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
