#!/usr/bin/env python3
"""Generate the Rust FFI crate of the drop-in boundary from include/hikari_hip.h.

    python tools/gen_rust_ffi.py            # writes rust/hikari-hip-sys/src/lib.rs
    python tools/gen_rust_ffi.py --check    # exit 1 if the committed file is stale

The header is the single source: every `#define` constant, every `Hk*` struct (`#[repr(C)]`, field for field, with a
compile-time size assertion computed here by the C layout rules), every enum (as `u32` constants - the ABI passes them as
uint32_t) and every `hk_*` function becomes Rust.  `tests/test_rust_ffi.py` parses BOTH files independently and compares
names, arity, parameter types, field order and sizes (also against the ctypes mirror in bevy-hikari_amd/_ffi.py).
There is no rustc in this image; the file is what `bevy-hikari`'s `src/lib.rs:95-370` / `light.rs:581-702` would link.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hikari_hip.h")
OUT = os.path.join(ROOT, "rust", "hikari-hip-sys", "src", "lib.rs")

SCALARS = {"uint8_t": ("u8", 1), "uint16_t": ("u16", 2), "uint32_t": ("u32", 4), "uint64_t": ("u64", 8), "int32_t": ("i32", 4), "int": ("i32", 4),
           "float": ("f32", 4), "double": ("f64", 8), "size_t": ("usize", 8), "char": ("c_char", 1)}
OPAQUE = {"hk_ctx": "HkCtx", "hk_scene_builder": "HkSceneBuilder", "hk_multi": "HkMulti"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def parse_header(path=HEADER):
    """-> dict(defines=[(name, value)], enums=[(name, [(member, value)])], structs=[(name, [(field, ctype, dims)])],
    opaque=[name], functions=[(name, ret, [(pname, ctype, is_const, ptr_depth, is_array)])])"""
    raw = open(path).read()
    text = strip_comments(raw)
    defines = []
    for m in re.finditer(r"^#define\s+(HK_\w+)\s+(.+?)\s*$", text, flags=re.M):
        val = m.group(2).strip()
        if re.fullmatch(r"\(?-?(0x[0-9a-fA-F]+|\d+)u?\)?", val):
            defines.append((m.group(1), val.strip("()").rstrip("u")))
        else:
            defines.append((m.group(1), val))  # an expression over other constants (kept verbatim, see emit)
    body = re.sub(r"^#.*$", " ", text, flags=re.M)
    body = body.replace('extern "C" {', " ")
    enums, structs, opaque, functions = [], [], [], []
    for m in re.finditer(r"typedef\s+enum\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", body, flags=re.S):
        members, nxt = [], 0
        for item in m.group(2).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = (x.strip() for x in item.split("="))
                nxt = int(v, 0)
            else:
                k = item
            members.append((k, nxt))
            nxt += 1
        enums.append((m.group(3), members))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", body, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            head = re.match(r"(const\s+)?(\w+)\s*", decl)
            assert head, f"cannot parse field {decl!r} of {m.group(3)}"
            for part in decl[head.end():].split(","):   # "uint32_t width, height": several declarators of one type
                fm = re.fullmatch(r"\s*(\**)\s*(\w+)((?:\[\w+\])*)\s*", part)
                assert fm, f"cannot parse field {decl!r} of {m.group(3)}"
                dims = [d for d in re.findall(r"\[(\w+)\]", fm.group(3))]
                ctype = head.group(2) if not fm.group(1) else ("const " if head.group(1) else "") + head.group(2) + fm.group(1)
                fields.append((fm.group(2), ctype, dims))
        structs.append((m.group(3), fields))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", body):
        opaque.append(m.group(2))
    no_types = re.sub(r"typedef\s+(?:enum|struct)\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", body, flags=re.S)
    for stmt in no_types.replace("}", ";").split(";"):
        m = re.fullmatch(r"\s*((?:const\s+)?\w+\s*\**)\s*\b(hk_\w+)\s*\(([^)]*)\)\s*", stmt, flags=re.S)
        if not m:
            continue
        ret = " ".join(m.group(1).split())
        params = []
        plist = " ".join(m.group(3).split())
        if plist and plist != "void":
            for p in plist.split(","):
                p = p.strip()
                pm = re.fullmatch(r"(const\s+)?(\w+)\s*(\**)\s*(const\s+)?(\w+)?\s*((?:\[\w*\])*)", p)
                assert pm, f"cannot parse parameter {p!r} of {m.group(2)}"
                params.append((pm.group(5) or f"arg{len(params)}", pm.group(2), bool(pm.group(1)), len(pm.group(3)), bool(pm.group(6))))
        functions.append((m.group(2), ret, params))
    return {"defines": defines, "enums": enums, "structs": structs, "opaque": opaque, "functions": functions}


def const_values(api):
    vals = {}
    for k, v in api["defines"]:
        try:
            vals[k] = int(v, 0)
        except ValueError:
            expr = re.sub(r"\b(\d+)u\b", r"\1", v)
            try:
                vals[k] = int(eval(expr, {}, vals))  # noqa: S307 - an arithmetic expression over the header's own constants
            except Exception:
                pass
    for _n, members in api["enums"]:
        vals.update(dict(members))
    return vals


def layout(api):
    """C layout (natural alignment) of every struct: {name: (size, align, [(field, offset, size)])}."""
    vals = const_values(api)
    done = {}

    def dim(d):
        return int(d) if d.isdigit() else vals[d]

    def type_info(ctype):
        if ctype.endswith("*"):
            return 8, 8
        if ctype in SCALARS:
            return SCALARS[ctype][1], SCALARS[ctype][1]
        return done[ctype][0], done[ctype][1]

    for name, fields in api["structs"]:
        off, align, out = 0, 1, []
        for fname, ctype, dims in fields:
            size, al = type_info(ctype)
            count = 1
            for d in dims:
                count *= dim(d)
            off = (off + al - 1) // al * al
            out.append((fname, off, size * count))
            off += size * count
            align = max(align, al)
        done[name] = ((off + align - 1) // align * align, align, out)
    return done


def rust_type(ctype, is_const=False, ptr=0, dims=(), vals=None):
    if ctype.endswith("*"):  # a pointer-typed struct field: "const uint8_t*"
        is_const = ctype.startswith("const ")
        ptr += ctype.count("*")
        ctype = ctype.replace("const ", "").rstrip("*")
    if ctype == "void":
        base = "c_void"
    elif ctype in SCALARS:
        base = SCALARS[ctype][0]
    elif ctype in OPAQUE:
        base = OPAQUE[ctype]
    else:
        base = ctype
    for d in reversed(list(dims)):
        base = f"[{base}; {d if d.isdigit() else d + ' as usize'}]"
    for level in range(ptr):  # `const T** p`: the const binds to T, the outer pointer is mutable
        base = ("*const " if is_const and level == 0 else "*mut ") + base
    return base


def emit(api):
    vals = const_values(api)
    lay = layout(api)
    L = []
    w = L.append
    w("// @generated by tools/gen_rust_ffi.py from include/hikari_hip.h - do not edit; `python tools/gen_rust_ffi.py` regenerates it,")
    w("// tests/test_rust_ffi.py checks it against the header (names, arity, types, field order, sizes).")
    w("//")
    w("// hikari-hip-sys: the raw binding bevy-hikari's render-graph nodes call instead of recording wgpu compute passes")
    w("// (reference src/lib.rs:252-367, src/light.rs:581-702, src/prepass.rs:769-852, src/post_process.rs:1190-1311).")
    w("#![allow(non_camel_case_types, non_snake_case, non_upper_case_globals, clippy::too_many_arguments)]")
    w("")
    w("use core::ffi::{c_char, c_void};")
    w("")
    w("// ---------------------------------------------------------------- constants")
    for k, v in api["defines"]:
        if k not in vals:
            continue
        ty = "i32" if vals[k] < 0 or k == "HK_OK" or k.startswith("HK_E_") else "u32"
        w(f"pub const {k}: {ty} = {vals[k]};")
    w("")
    w("// ---------------------------------------------------------------- enums (passed as uint32_t across the ABI)")
    for name, members in api["enums"]:
        w(f"pub type {name} = u32;")
        for k, v in members:
            w(f"pub const {k}: {name} = {v};")
        w("")
    w("// ---------------------------------------------------------------- opaque handles")
    for c in api["opaque"]:
        w("#[repr(C)]")
        w(f"pub struct {OPAQUE[c]} {{")
        w("    _private: [u8; 0],")
        w("}")
    w("")
    w("// ---------------------------------------------------------------- plain-data structs (std430 / std140 layouts of the reference)")
    for name, fields in api["structs"]:
        w("#[repr(C)]")
        w("#[derive(Clone, Copy, Debug)]")
        w(f"pub struct {name} {{")
        for fname, ctype, dims in fields:
            w(f"    pub {fname}: {rust_type(ctype, dims=dims)},")
        w("}")
        w(f"const _: () = assert!(core::mem::size_of::<{name}>() == {lay[name][0]});")
        w("")
    w("// ---------------------------------------------------------------- entry points")
    w('#[link(name = "hikari_hip")]')
    w('extern "C" {')
    for fname, ret, params in api["functions"]:
        ps = []
        for pname, ctype, is_const, ptr, is_array in params:
            ps.append(f"{pname}: {rust_type(ctype, is_const, ptr + (1 if is_array else 0))}")
        if ret == "void":
            r = ""
        elif ret.replace(" ", "") == "constchar*":
            r = " -> *const c_char"
        else:
            r = f" -> {rust_type(ret)}"
        w(f"    pub fn {fname}({', '.join(ps)}){r};")
    w("}")
    return "\n".join(L) + "\n"


def main():
    api = parse_header()
    text = emit(api)
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            sys.exit("rust/hikari-hip-sys/src/lib.rs is stale: run python tools/gen_rust_ffi.py")
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"{os.path.relpath(OUT, ROOT)}: {len(api['functions'])} functions, {len(api['structs'])} structs, {len(api['enums'])} enums, "
          f"{len(const_values(api))} constants")


if __name__ == "__main__":
    main()
