#!/usr/bin/env python3
"""Builds libhikari_hip.so for gfx950: every source to an object of its own (in parallel, only what changed), then one link - 25 s
instead of the 3 minutes of one hipcc invocation over all sources.  Used by __graft_entry__.build() and tools/build_variant.sh.
    python tools/build_lib.py [-o out.so] [--objdir dir] [-DFLAG ...]      (extra flags reach every compile)"""
import concurrent.futures
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bevy-hikari_amd", "csrc")
# -ffp-contract=off: only the fmaf() calls written in the sources become v_fma_f32 (numeric contract).  gfx950 only.
# -fvisibility=hidden: the library exports the entry points of include/hikari_hip.h / hikari_hip_debug.h (their #pragma GCC visibility) and nothing else.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
SOURCES = ["kernels.hip", "kernels_denoise.hip", "kernels_aa.hip", "kernels_wavefront.hip", "kernels_scene.hip", "context.hip", "scene_layout.hip", "scene_refit.hip",
           "probes.hip", "host_logic.cpp", "scene_builder.cpp", "comm.cpp"]


def source_hash():
    """sha256 over every source the library is built from (csrc/*, include/*.h), names included: what hk_build_info() reports and
    __graft_entry__ compares with the tree, so that a stale binary is visible (and rebuilt) wherever the tree travels."""
    import hashlib

    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h", ".cpp"))]
    files += [os.path.join(ROOT, "include", f) for f in ("hikari_hip.h", "hikari_hip_debug.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_library(out, objdir=None, extra=(), force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # one object directory per OUTPUT (a variant built with other flags never evicts the shipped library's objects)
    objdir = objdir or os.path.join(ROOT, "build", "obj_" + os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))] + [os.path.join(ROOT, "include", "hikari_hip.h"),
                                                                                                   os.path.join(ROOT, "include", "hikari_hip_debug.h")]
    # what an object depends on besides its sources: the compiler (path + version) and EVERY flag - the numeric contract lives in
    # -ffp-contract=off, an object compiled without it must never be linked silently
    stamp = os.path.join(objdir, "flags.txt")
    try:
        version = subprocess.run([hipcc, "--version"], capture_output=True, text=True, check=True).stdout.strip().replace("\n", " | ")
    except (OSError, subprocess.CalledProcessError):
        version = "unknown"
    flag_line = " ".join([hipcc, version] + FLAGS + list(extra))
    old = None
    if os.path.exists(stamp):
        with open(stamp) as f:
            old = f.read()
    if old != flag_line:
        force = True
    jobs = []
    for src in SOURCES:
        base = os.path.splitext(src)[0]
        jobs.append((src, os.path.join(objdir, base + ".o"), []))

    def stale(obj, src):
        if force or not os.path.exists(obj):
            return True
        t = os.path.getmtime(obj)
        return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, src)] + headers)

    def compile_one(job):
        src, obj, more = job
        cmd = [hipcc] + FLAGS + list(extra) + more + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    todo = [j for j in jobs if stale(j[1], j[0])]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as pool:
        list(pool.map(compile_one, todo))
    with open(stamp, "w") as f:
        f.write(flag_line)
    objs = [j[1] for j in jobs]
    # the build stamp (hk_build_info, include/hikari_hip.h): sources' hash, compiler, every flag, when - compiled into the library
    import time

    src_hash = source_hash()
    stamp_text = "sources %s | %s | built %s" % (src_hash, flag_line, time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()))
    stamp_cpp, stamp_obj = os.path.join(objdir, "build_stamp.cpp"), os.path.join(objdir, "build_stamp.o")
    old_stamp = open(stamp_cpp).read() if os.path.exists(stamp_cpp) else ""
    if todo or force or ("sources %s |" % src_hash) not in old_stamp or not os.path.exists(stamp_obj):
        with open(stamp_cpp, "w") as f:
            f.write('extern "C" __attribute__((visibility("default"))) const char* hk_build_info(void) { return "%s"; }\n' % stamp_text.replace("\\", "/").replace('"', "'"))
        subprocess.run(["g++", "-O1", "-fPIC", "-c", stamp_cpp, "-o", stamp_obj], check=True)
        todo = todo or [None]
    objs.append(stamp_obj)
    if todo or not os.path.exists(out) or any(os.path.getmtime(o) > os.path.getmtime(out) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(out + ".stamp", "w") as f:   # (sidecar: what __graft_entry__.build() compares with the tree BEFORE loading the library)
            f.write(src_hash + "\n")
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    out = os.path.join(ROOT, "bevy-hikari_amd", "libhikari_hip.so")
    objdir = None
    extra = []
    while args:
        a = args.pop(0)
        if a == "-o":
            out = os.path.abspath(args.pop(0))
        elif a == "--objdir":
            objdir = os.path.abspath(args.pop(0))
        else:
            extra.append(a)
    build_library(out, objdir, extra)
