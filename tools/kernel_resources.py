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


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def instruction_counts(path):
    """{demangled kernel name: {"valu": n, "salu": n, "vmem": n, "lds": n, "total": n}} - STATIC counts from the disassembly of the
    kernels' code (every instruction once, whatever the control flow): what a VALU-bound kernel costs to first order."""
    out = {}
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(blob)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in text.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
            if m:
                cur = out.setdefault(m.group(1), {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "total": 0})
                continue
            t = line.strip().split(" ", 1)[0] if cur is not None else ""
            if not t or not re.match(r"^[a-z]+_", t):
                continue
            cur["total"] += 1
            if t.startswith("v_"):
                cur["valu"] += 1
            elif t.startswith("s_"):
                cur["salu"] += 1
            elif t.startswith(("global_", "buffer_", "flat_", "scratch_")):
                cur["vmem"] += 1
            elif t.startswith("ds_"):
                cur["lds"] += 1
    names = demangle(list(out))
    return {names[k]: v for k, v in out.items() if v["total"]}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bevy-hikari_amd", "libhikari_hip.so")
    isa = instruction_counts(lib) if "--isa" in sys.argv else {}
    for name, r in sorted(resources(lib).items()):
        short = re.sub(r"\(.*", "", name.replace("void hkd::", "").replace("hkd::", ""))
        if name in isa:
            i = isa[name]
            print(f"{short[:60]:<60} valu={i['valu']:>5} salu={i['salu']:>5} vmem={i['vmem']:>4} lds={i['lds']:>4} total={i['total']:>5}  ", end="")
        print(f"{short[:60]:<60} vgpr={r.get('vgpr_count'):>3} spill={r.get('vgpr_spill_count'):>3} scratch={r.get('private_segment_fixed_size'):>5} "
              f"lds={r.get('group_segment_fixed_size'):>6} sgpr={r.get('sgpr_count'):>3}")
