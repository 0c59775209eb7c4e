"""Preprocess (bevy 0.9 `#import` / `#define_import_path` / `#ifdef`), translate and execute WGSL compute entry
points one invocation at a time.  TEST INFRASTRUCTURE ONLY (see runtime.py)."""
import glob
import os
import re

import numpy as np

from . import runtime as R
from . import translate
from . import types as T

HERE = os.path.dirname(os.path.abspath(__file__))


def _split_modules(path):
    """a file may hold several `#define_import_path X` sections (the bevy stub does)"""
    text = open(path).read()
    parts = re.split(r"^#define_import_path\s+(\S+)\s*$", text, flags=re.M)
    if len(parts) == 1:
        return {}
    return {parts[i]: parts[i + 1] for i in range(1, len(parts), 2)}


def library(shader_dir):
    mods = {}
    for path in sorted(glob.glob(os.path.join(shader_dir, "*.wgsl"))) + [os.path.join(HERE, "bevy_0_9_1.wgsl")]:
        mods.update(_split_modules(path))
    return mods


def preprocess(text, defs, mods, done=None):
    done = set() if done is None else done
    out, stack = [], []
    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("#ifdef") or s.startswith("#ifndef"):
            name = s.split()[1]
            stack.append((name in defs) == s.startswith("#ifdef"))
            continue
        if s.startswith("#else"):
            stack[-1] = not stack[-1]
            continue
        if s.startswith("#endif"):
            stack.pop()
            continue
        if not all(stack):
            continue
        if s.startswith("#import"):
            name = s.split()[1]
            if name in done:
                continue
            done.add(name)
            if name not in mods:
                raise KeyError(f"#import {name}: no module with that import path")
            out.append(preprocess(mods[name], defs, mods, done))
            continue
        if s.startswith("#define_import_path"):
            continue
        out.append(line)
    return "\n".join(out)


class Module:
    def __init__(self, shader_dir, filename, defs=()):
        mods = library(shader_dir)
        self.source = preprocess(open(os.path.join(shader_dir, filename)).read(), set(defs), mods)
        self.python = translate.translate(self.source)
        self.ns = {"_R": R, "_T": T, "RESOURCES": {}, "WORKGROUP_VARS": {}, "ENTRY_POINTS": {}, "_ONCE": (0,)}
        exec(compile(self.python, f"<wgsl:{filename}>", "exec"), self.ns)

    def bind(self, **resources):
        """name -> numpy uint8 buffer (decoded through the declared type), Texture, Sampler, list of them, or a ready object"""
        for name, obj in resources.items():
            group, binding, ty, space = self.ns["RESOURCES"][name]
            if isinstance(obj, np.ndarray) and obj.dtype == np.uint8 and obj.ndim == 1:
                obj = ty.decode(obj, 0)
            self.ns[translate.pyname(name)] = obj

    def unbound(self, used_only_by=None):
        return [n for n in self.ns["RESOURCES"] if translate.pyname(n) not in self.ns]

    def dispatch(self, entry, groups_x, groups_y, only=None, limit=None):
        """run `entry` for every invocation of a groups_x x groups_y grid of workgroups (z = 1).  `only(gx, gy)` filters
        workgroups.  Entry points that contain workgroupBarrier() are generators and run in lockstep per workgroup."""
        fn, wg, builtins = self.ns["ENTRY_POINTS"][entry]
        wx, wy = wg[0], wg[1] if len(wg) > 1 else 1
        if not self.ns["WORKGROUP_VARS"] or "yield" not in self.python[self.python.index(f"def {translate.pyname(entry)}("):].split("\ndef ", 1)[0]:
            # no barrier: invocations are independent; run them in LINEAR order (y, then x) so that stores which race in the
            # reference (a thread writing another thread's slot) resolve as "highest invocation index wins", the oracle's rule
            for y in range(groups_y * wy):
                for x in range(groups_x * wx):
                    if limit is not None and (x >= limit[0] or y >= limit[1]):
                        continue
                    vals = {"global_invocation_id": T.vec3u32(x, y, 0), "local_invocation_id": T.vec3u32(x % wx, y % wy, 0),
                            "workgroup_id": T.vec3u32(x // wx, y // wy, 0), "num_workgroups": T.vec3u32(groups_x, groups_y, 1),
                            "local_invocation_index": R.u32((y % wy) * wx + x % wx)}
                    fn(*[vals[b] for b in builtins.values()])
            return
        for gy in range(groups_y):
            for gx in range(groups_x):
                if only is not None and not only(gx, gy):
                    continue
                for name, ty in self.ns["WORKGROUP_VARS"].items():
                    self.ns[name] = ty.zero()
                pending = []
                for ly in range(wy):
                    for lx in range(wx):
                        if limit is not None and (gx * wx + lx >= limit[0] or gy * wy + ly >= limit[1]):
                            continue
                        vals = {"global_invocation_id": T.vec3u32(gx * wx + lx, gy * wy + ly, 0), "local_invocation_id": T.vec3u32(lx, ly, 0),
                                "workgroup_id": T.vec3u32(gx, gy, 0), "num_workgroups": T.vec3u32(groups_x, groups_y, 1),
                                "local_invocation_index": R.u32(ly * wx + lx)}
                        r = fn(*[vals[b] for b in builtins.values()])
                        if r is not None and hasattr(r, "__next__"):
                            pending.append(r)
                while pending:       # advance every invocation to its next barrier
                    alive = []
                    for g in pending:
                        try:
                            next(g)
                            alive.append(g)
                        except StopIteration:
                            pass
                    pending = alive
