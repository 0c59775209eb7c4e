#!/usr/bin/env python3
"""Register / scratch / LDS use of every gfx950 kernel in the built libhikari_hip.so, read from the code objects' metadata (no
GPU, no recompile): the clang offload bundles in the library are unpacked and handed to llvm-readelf --notes.
Usage: python tools/kernel_resources.py [path/to/libhikari_hip.so]   (prints one line per kernel; importable: resources(path))"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
          "max_flat_workgroup_size")


def code_objects(path, arch="gfx950"):
    data = open(path, "rb").read()
    for m in re.finditer(MAGIC, data):
        p = m.start()
        (n,) = struct.unpack_from("<Q", data, p + len(MAGIC))
        q = p + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            q += 24
            triple = data[q:q + tl].decode()
            q += tl
            if arch in triple and size:
                yield data[p + off:p + off + size]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def resources(path):
    """{demangled kernel name: {field: int}} for every kernel of every gfx950 code object in `path`."""
    out = {}
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(blob)
            f.flush()
            text = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        # one YAML-ish block per kernel between "- .agpr_count" / "- .args" list items; fields are unordered, .name is among them
        for block in re.split(r"\n\s+- \.", text):
            name = re.search(r"\.?name:\s+(_Z\S+|\w+)\s*$", block, re.M)
            if not name or "vgpr_count" not in block:
                continue
            rec = {}
            for k in FIELDS:
                m = re.search(rf"\.?{k}:\s+(\d+)", block)
                if m:
                    rec[k] = int(m.group(1))
            out[name.group(1)] = rec
    names = demangle(list(out))
    return {names[k]: v for k, v in out.items()}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bevy-hikari_amd", "libhikari_hip.so")
    for name, r in sorted(resources(lib).items()):
        short = re.sub(r"\(.*", "", name.replace("void hkd::", "").replace("hkd::", ""))
        print(f"{short[:60]:<60} vgpr={r.get('vgpr_count'):>3} spill={r.get('vgpr_spill_count'):>3} scratch={r.get('private_segment_fixed_size'):>5} "
              f"lds={r.get('group_segment_fixed_size'):>6} sgpr={r.get('sgpr_count'):>3}")
