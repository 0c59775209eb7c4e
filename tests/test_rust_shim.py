"""rust/bevy-hikari-hip/hip.rs - the node-level shim a bevy-hikari maintainer drops into the reference crate (VERDICT r05
missing 3) - cannot be compiled here (no Rust toolchain).  What CAN be checked without one: every `hk::` item it names exists in the
generated -sys crate (which tests/test_rust_ffi.py holds to include/hikari_hip.h), every hk_* call passes as many arguments as the
function takes, every struct literal names each field of the struct exactly once, and every field it assigns exists."""
import os
import re

from conftest import ROOT
from test_rust_ffi import parse_rust

SHIM = os.path.join(ROOT, "rust", "bevy-hikari-hip", "hip.rs")


def strip_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def call_arguments(text, start):
    """the top-level comma-separated arguments of the call whose '(' is at `start`"""
    depth, args, cur = 0, [], ""
    for ch in text[start:]:
        if ch in "([{":
            depth += 1
            if depth == 1:
                continue
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                break
        if ch == "," and depth == 1:
            args.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur)
    return [a.strip() for a in args]


def test_every_item_the_shim_names_exists_and_every_call_has_the_right_arity():
    consts, structs, _, fns = parse_rust()
    text = strip_comments(open(SHIM).read())
    named = set(re.findall(r"\bhk::(\w+)", text))
    assert len(named) > 25
    for name in named:
        assert name in consts or name in structs or name in fns or name in ("HkCtx",), name
    calls = 0
    for m in re.finditer(r"\bhk::(hk_\w+)\s*\(", text):
        args = call_arguments(text, m.end() - 1)
        assert len(args) == len(fns[m.group(1)][0]), (m.group(1), args, fns[m.group(1)][0])
        calls += 1
    assert calls >= 12
    # the frame's calls, in the order include/hikari.hpp and plugin.py issue them
    order = [text.index(s) for s in ("hk::hk_resize(", "hk::hk_frame_render(")]
    assert order == sorted(order)


def test_struct_literals_and_field_assignments_match_the_sys_crate():
    _, structs, _, _ = parse_rust()
    text = strip_comments(open(SHIM).read())
    literals = 0
    for m in re.finditer(r"(?<!-> )\bhk::(Hk\w+)\s*\{", text):   # (not a function's return type in front of its body)
        fields = [re.match(r"(\w+)", a).group(1) for a in call_arguments(text, m.end() - 1)]
        want = [f for f, _ in structs[m.group(1)]]
        assert sorted(fields) == sorted(want), (m.group(1), sorted(set(want) ^ set(fields)))
        literals += 1
    assert literals >= 3
    # `let mut out: hk::HkFrame = ..; out.field = ..` / `lights.field = ..`
    for var, ty in re.findall(r"let mut (\w+): hk::(Hk\w+)", text):
        want = {f for f, _ in structs[ty]}
        assigned = set(re.findall(r"\b%s\.(\w+)(?:\[\w+\])?\s*=" % var, text))
        assert assigned and assigned <= want, (ty, assigned - want)
        if ty == "HkFrame":   # everything but the padding is written
            assert want - assigned <= {"_pad"}, want - assigned
