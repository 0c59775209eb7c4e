"""GLSL (the subset of FidelityFX FSR 1.0's ffx_a.h / ffx_fsr1.h / FSR_Pass.glsl after `cpp`) -> Python, on the runtime of
tests/tools/wgsl.  TEST INFRASTRUCTURE ONLY: it executes the GLSL the reference ships in src/shaders/fsr/source.zip (the source of
its fsr_pass_{easu,rcas}.spv) so that the oracle's FSR1 restatement can be pinned against it.

C-like recursive-descent parser producing Python text directly: functions, declarations with several declarators, if / else / for /
return, the ternary operator, compound assignment, ++/--; `out` / `inout` parameters travel in _R.Box cells that the caller writes
back after the statement; block scoping by renaming, as in wgsl/translate.py.  Top-level functions that use anything outside the
subset are skipped (reported in Module.skipped) - nothing the two passes reach is."""
import re
import subprocess

import numpy as np

from wgsl import runtime as R
from wgsl import types as T

TYPES = {"void", "float", "int", "uint", "bool", "vec2", "vec3", "vec4", "ivec2", "ivec3", "ivec4", "uvec2", "uvec3", "uvec4", "bvec2", "bvec3", "bvec4",
         "mat2", "mat3", "mat4", "texture2D", "sampler", "image2D", "sampler2D"}
TOKEN = re.compile(r"\s+|//[^\n]*|/\*.*?\*/|(?P<num>0[xX][0-9a-fA-F]+[uU]?|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[uUfF]?)|(?P<id>[A-Za-z_]\w*)"
                   r"|(?P<op><<=|>>=|\+\+|--|&&|\|\||==|!=|<=|>=|<<|>>|\+=|-=|\*=|/=|%=|&=|\|=|\^=|[-+*/%&|^~!<>=?:.,;(){}\[\]])", re.S)
SCALAR = {"float": "f32", "int": "i32", "uint": "u32", "bool": "bool"}
VEC = {"vec": "f32", "ivec": "i32", "uvec": "u32", "bvec": "bool"}
BUILTINS = {"min": "_R.w_min", "max": "_R.w_max", "abs": "_R.w_abs", "clamp": "_R.w_clamp", "mix": "_R.w_mix", "fract": "_R.w_fract", "floor": "_R.w_floor",
            "trunc": "_R.w_trunc", "sqrt": "_R.w_sqrt", "pow": "_R.w_pow", "exp2": "_R.w_exp2", "log2": "_R.w_log2", "sin": "_R.w_sin", "cos": "_R.w_cos",
            "sign": "_R.w_sign", "step": "_R.w_step", "dot": "_R.dot", "inversesqrt": "_R.w_inverse_sqrt", "packHalf2x16": "_R.pack2x16float",
            "unpackHalf2x16": "_R.unpack2x16float", "floatBitsToUint": "_G.float_bits_to_uint", "uintBitsToFloat": "_G.uint_bits_to_float",
            "bitfieldExtract": "_G.bitfield_extract", "bitfieldInsert": "_G.bitfield_insert", "textureSize": "_G.texture_size", "texture": "_G.texture",
            "textureLod": "_G.texture", "texelFetch": "_G.texel_fetch", "imageStore": "_G.image_store", "sampler2D": "_G.sampler2d"}
PYKW = {"lambda", "from", "in", "is", "not", "and", "or", "def", "class", "pass", "global", "del", "with", "as", "import", "yield", "None", "True", "False", "try",
        "except", "raise", "while", "print", "len", "id", "min", "max", "abs", "type", "hash", "input", "iter", "next", "filter", "map", "round", "pow"}


class Skip(Exception):
    pass


class G:
    """GLSL builtins that are not in the WGSL runtime"""

    @staticmethod
    def float_bits_to_uint(x):
        return R.bitcast("u32", R.vconvert("f32", x))

    @staticmethod
    def uint_bits_to_float(x):
        return R.bitcast("f32", R.vconvert("u32", x))

    @staticmethod
    def bitfield_extract(v, off, bits):
        off, bits = int(off), int(bits)
        return R.u32((int(v) >> off) & ((1 << bits) - 1)) if bits else R.u32(0)

    @staticmethod
    def bitfield_insert(base, ins, off, bits):
        off, bits = int(off), int(bits)
        mask = ((1 << bits) - 1) << off
        return R.u32((int(base) & ~mask & 0xFFFFFFFF) | ((int(ins) << off) & mask))

    @staticmethod
    def sampler2d(tex, smp):
        return (tex, smp)

    @staticmethod
    def texture_size(ts, lod):
        return T.vec2i32(ts[0].w, ts[0].h)

    @staticmethod
    def texture(ts, uv, lod=None):
        """bilinear with the weights snapped to 1/256, the sub-texel precision of the texture units the blob runs on: a sample at a
        texel centre (+- an ulp of address arithmetic) returns that texel - what the oracle's contract takes fakeTextureGather to be"""
        tex, smp = ts
        w, h = R.f32(tex.w), R.f32(tex.h)
        px, py = uv[0] * w - R.f32(0.5), uv[1] * h - R.f32(0.5)
        fx0, fy0 = np.floor(px), np.floor(py)
        fx, fy = R.f32(np.rint((px - fx0) * R.f32(256.0))) / R.f32(256.0), R.f32(np.rint((py - fy0) * R.f32(256.0))) / R.f32(256.0)
        ix, iy = int(fx0), int(fy0)
        cl = lambda i, n: min(max(i, 0), n - 1)
        t00, t10 = tex.texel(cl(ix, tex.w), cl(iy, tex.h)), tex.texel(cl(ix + 1, tex.w), cl(iy, tex.h))
        t01, t11 = tex.texel(cl(ix, tex.w), cl(iy + 1, tex.h)), tex.texel(cl(ix + 1, tex.w), cl(iy + 1, tex.h))
        return R.w_mix(R.w_mix(t00, t10, fx), R.w_mix(t01, t11, fx), fy)

    @staticmethod
    def texel_fetch(ts, coords, lod):
        return T.texture_load(ts[0], coords, lod)

    @staticmethod
    def image_store(img, coords, value):
        T.texture_store(img, coords, value)


def pyname(n):
    return n + "_" if n in PYKW else n


class Translator:
    def __init__(self, src):
        self.t = [(m.lastgroup, m.group(m.lastgroup)) for m in TOKEN.finditer(src) if m.lastgroup]
        self.t.append(("eof", ""))
        self.i = 0
        self.out, self.skipped, self.consts = [], [], []
        self.sig = {}            # function -> list of bool: parameter is out / inout
        self.tmp = 0

    # ------------------------------------------------------------ tokens
    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        self.i += 1
        return self.t[self.i - 1]

    def at(self, text):
        return self.t[self.i][1] == text and self.t[self.i][0] != "num"

    def accept(self, text):
        if self.at(text):
            self.i += 1
            return True
        return False

    def expect(self, text):
        if not self.accept(text):
            raise Skip(f"expected {text!r}, got {self.peek()[1]!r}")

    def hoist(self, expr):
        if expr not in self.consts:
            self.consts.append(expr)
        return f"_k{self.consts.index(expr)}"

    # ------------------------------------------------------------ module
    def module(self):
        while self.peek()[0] != "eof":
            start = self.i
            try:
                self.top_level()
            except Skip as e:
                self.i = start
                name = self.skip_top_level()
                self.skipped.append((name, str(e)))
        head = ["# generated by tests/tools/glsl.py - do not edit"] + [f"_k{i} = {c}" for i, c in enumerate(self.consts)]
        return "\n".join(head + self.out) + "\n"

    def skip_top_level(self):
        name, depth = None, 0
        while self.peek()[0] != "eof":
            kind, text = self.next()
            if kind == "id" and name is None and text not in TYPES and text not in ("layout", "uniform", "in", "const"):
                name = text
            if text == "{":
                depth += 1
            elif text == "}":
                depth -= 1
                if depth == 0:
                    self.accept(";")
                    return name
            elif text == ";" and depth == 0:
                return name
        return name

    def top_level(self):
        if self.accept(";"):
            return
        if self.at("layout"):
            self.next()
            self.expect("(")
            while not self.accept(")"):
                self.next()
            if self.accept("in"):                 # layout(local_size_x=64) in;
                self.expect(";")
                return
            self.expect("uniform")
            if self.peek()[0] == "id" and self.peek()[1] not in TYPES:      # uniform block: members become module globals
                self.next()
                self.expect("{")
                while not self.accept("}"):
                    self.next()
                self.accept(";")
                return
            self.next()                           # texture2D / image2D / sampler
            self.next()
            self.expect(";")
            return
        ty = self.next()[1]
        if ty not in TYPES:
            raise Skip(f"unexpected {ty!r} at top level")
        name = self.next()[1]
        if not self.at("("):
            raise Skip("global variable")
        self.next()
        params, boxed = [], []
        while not self.accept(")"):
            qual = "in"
            while self.peek()[1] in ("in", "out", "inout", "const"):
                q = self.next()[1]
                qual = q if q != "const" else qual
            pty = self.next()[1]
            if pty not in TYPES:
                raise Skip(f"parameter type {pty!r}")
            params.append(self.next()[1])
            boxed.append(qual in ("out", "inout"))
            if self.at("["):
                raise Skip("array parameter")
            self.accept(",")
        self.sig[name] = boxed
        self.scopes, self.used = [{}], set()
        self.boxes = set()
        pnames = []
        for p, b in zip(params, boxed):
            pnames.append(self.declare(p))
            if b:
                self.boxes.add(p)
        lines = [f"def {pyname(name)}({', '.join(pnames)}):"]
        body = []
        self.block(body, 1, [])
        if not body:
            body.append("    pass")
        self.out.extend(lines + body)

    # ------------------------------------------------------------ scoping
    def declare(self, name):
        py = pyname(name)
        if any(name in sc for sc in self.scopes[:-1]) or (py in self.used and name not in self.scopes[-1]):
            k = 1
            while f"{py}__{k}" in self.used:
                k += 1
            py = f"{py}__{k}"
        self.scopes[-1][name] = py
        self.used.add(py)
        return py

    def lookup(self, name):
        for sc in reversed(self.scopes):
            if name in sc:
                return sc[name] + (".v" if name in self.boxes and sc is self.scopes[0] else "")
        return pyname(name)

    # ------------------------------------------------------------ statements
    def block(self, out, depth, loops):
        self.expect("{")
        self.scopes.append({})
        while not self.accept("}"):
            self.statement(out, depth, loops)
        self.scopes.pop()

    def body(self, out, depth, loops):
        n0 = len(out)
        if self.at("{"):
            self.block(out, depth, loops)
        else:
            self.scopes.append({})
            self.statement(out, depth, loops)
            self.scopes.pop()
        if len(out) == n0:
            out.append("    " * depth + "pass")

    def flush(self, out, depth, line_fn):
        """emit one statement built from expressions that may have produced boxed-argument prologues / epilogues"""
        self.pre, self.post = [], []
        lines = line_fn()
        for l in self.pre:
            out.append("    " * depth + l)
        for l in lines:
            out.append("    " * depth + l)
        for target, src in self.post:
            for l in self.assign_lines(target, src):
                out.append("    " * depth + l)
        self.pre, self.post = [], []

    def statement(self, out, depth, loops):
        ind = "    " * depth
        if self.accept(";"):
            return
        if self.at("{"):
            self.block(out, depth, loops)
            return
        if self.accept("if"):
            self.expect("(")
            self.pre, self.post = [], []
            cond = self.expr()
            if self.pre:
                raise Skip("boxed call in a condition")
            self.expect(")")
            out.append(f"{ind}if {cond}:")
            self.body(out, depth + 1, loops)
            if self.accept("else"):
                out.append(f"{ind}else:")
                self.body(out, depth + 1, loops)
            return
        if self.accept("for"):
            self.expect("(")
            self.scopes.append({})
            if not self.at(";"):
                self.simple(out, depth)
            self.expect(";")
            self.pre, self.post = [], []
            cond = "True" if self.at(";") else self.expr()
            self.expect(";")
            upd = []
            if not self.at(")"):
                self.simple(upd, depth + 1)
            self.expect(")")
            flag = f"_brk{len(loops)}"
            out.append(f"{ind}while {cond}:")
            out.append(f"{ind}    {flag} = False")
            out.append(f"{ind}    for _once in _ONCE:")
            self.body(out, depth + 2, loops + [flag])
            out.append(f"{ind}    if {flag}: break")
            out.extend(upd)
            self.scopes.pop()
            return
        if self.accept("return"):
            if self.accept(";"):
                out.append(f"{ind}return")
                return
            self.flush(out, depth, lambda: [f"return {self.expr()}"])
            self.expect(";")
            return
        if self.accept("break"):
            self.expect(";")
            out.extend([f"{ind}{loops[-1]} = True", f"{ind}break"])
            return
        if self.accept("continue"):
            self.expect(";")
            out.append(f"{ind}continue")
            return
        self.simple(out, depth)
        self.expect(";")

    def simple(self, out, depth):
        """declaration (several declarators) or expression / assignment statement"""
        while self.peek()[1] in ("const", "highp", "mediump", "lowp"):
            self.next()
        if self.peek()[1] in TYPES and self.peek(1)[0] == "id":
            ty = self.next()[1]
            while True:
                name = self.next()[1]
                if self.at("["):
                    raise Skip("array variable")
                if self.accept("="):
                    def line():
                        value = self.convert(ty, self.expr())
                        return [f"{self.declare(name)} = {value}"]
                    self.flush(out, depth, line)
                else:
                    out.append("    " * depth + f"{self.declare(name)} = {self.zero(ty)}")
                if not self.accept(","):
                    return
        def line():
            lhs = self.unary_ast()
            kind, text = self.peek()
            if text in ("=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>="):
                self.next()
                targets = [lhs]
                while text == "=":                      # a = b = value;
                    mark = self.i
                    node = self.unary_ast() if (self.peek()[0] == "id" and self.peek()[1] not in TYPES) else None
                    if node is not None and node[0] != "text" and self.at("="):
                        self.next()
                        targets.append(node)
                    else:
                        self.i = mark
                        break
                rhs = self.expr()
                if len(targets) > 1:
                    self.tmp += 1
                    lines = [f"_v{self.tmp} = {rhs}"]
                    for tgt in reversed(targets):
                        lines += self.assign_lines(tgt, f"_v{self.tmp}")
                    return lines
                value = rhs if text == "=" else self.binop(text[:-1], self.emit_ast(lhs), f"({rhs})")
                return self.assign_lines(lhs, value)
            if text in ("++", "--"):
                self.next()
                return self.assign_lines(lhs, self.binop(text[0], self.emit_ast(lhs), "1"))
            return [self.emit_ast(lhs)]
        self.flush(out, depth, line)

    def zero(self, ty):
        if ty in SCALAR:
            return {"f32": "_R.F0", "i32": "_R.i32(0)", "u32": "_R.u32(0)", "bool": "False"}[SCALAR[ty]]
        m = re.fullmatch(r"([iub]?vec)([234])", ty)
        if m:
            return f"_T.vec{m.group(2)}{VEC[m.group(1)]}()"
        raise Skip(f"zero value of {ty}")

    def convert(self, ty, src):
        return src

    # ------------------------------------------------------------ expressions: AST for lvalues, text otherwise
    def assign_lines(self, target, value):
        kind = target[0]
        if kind == "id":
            return [f"{self.lookup(target[1])} = {value}"]
        if kind == "member":      # swizzle of a vector variable
            base = self.emit_ast(target[1])
            return self.assign_lines(target[1], f"{base}.with_({target[2]!r}, {value})")
        if kind == "index":
            base = self.emit_ast(target[1])
            return self.assign_lines(target[1], f"{base}.with_index({target[2]}, {value})")
        raise Skip(f"assignment to {kind}")

    def emit_ast(self, a):
        if a[0] == "id":
            return self.lookup(a[1])
        if a[0] == "member":
            return f"{self.emit_ast(a[1])}.{a[2]}"
        if a[0] == "index":
            return f"{self.emit_ast(a[1])}[{a[2]}]"
        return a[1]

    def unary_ast(self):
        """postfix expression kept as a small AST when it is an lvalue path"""
        kind, text = self.peek()
        if kind == "id" and text not in TYPES and text not in ("true", "false") and self.peek(1)[1] != "(":
            self.next()
            node = ("id", text)
            while True:
                if self.accept("."):
                    node = ("member", node, self.next()[1])
                elif self.accept("["):
                    idx = self.expr()
                    self.expect("]")
                    node = ("index", node, idx)
                else:
                    return node
        return ("text", self.unary())

    def binop(self, op, a, b):
        if op == "/":
            return f"_R.div({a}, {b})"
        if op == "%":
            return f"_R.mod({a}, {b})"
        if op == "<<":
            return f"_R.shl({a}, {b})"
        if op == ">>":
            return f"_R.shr({a}, {b})"
        if op == "&&":
            return f"({a} and {b})"
        if op == "||":
            return f"({a} or {b})"
        return f"({a} {op} {b})"

    LEVELS = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"), ("+", "-"), ("*", "/", "%")]

    def expr(self):
        cond = self.binary(0)
        if self.accept("?"):
            a = self.expr()
            self.expect(":")
            b = self.expr()
            return f"({a} if {cond} else {b})"
        return cond

    def binary(self, level):
        if level == len(self.LEVELS):
            return self.unary()
        lhs = self.binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[level]:
            op = self.next()[1]
            lhs = self.binop(op, lhs, self.binary(level + 1))
        return lhs

    def unary(self):
        kind, text = self.peek()
        if kind == "op" and text in ("-", "!", "~", "+"):
            self.next()
            x = self.unary()
            return {"-": f"(-{x})", "!": f"_R.lnot({x})", "~": f"(~{x})", "+": x}[text]
        if text in ("++", "--"):
            raise Skip("prefix increment in an expression")
        return self.postfix(self.primary())

    def postfix(self, e):
        while True:
            if self.accept("."):
                e = f"{e}.{self.next()[1]}"
            elif self.accept("["):
                idx = self.expr()
                self.expect("]")
                e = f"{e}[{idx}]"
            else:
                return e

    def primary(self):
        kind, text = self.next()
        if kind == "num":
            if text.lower().startswith("0x"):
                value = int(text.rstrip("uU"), 16)
                return self.hoist(f"_R.u32({value})") if text[-1] in "uU" else str(value)
            if re.fullmatch(r"\d+[uU]", text):
                return self.hoist(f"_R.u32({text[:-1]})")
            if re.fullmatch(r"\d+", text):
                return text
            return self.hoist(f"_R.f32({text.rstrip('fF')})")
        if text == "(":
            e = self.expr()
            self.expect(")")
            return f"({e})"
        if kind == "id":
            if text in ("true", "false"):
                return "True" if text == "true" else "False"
            if self.at("("):
                self.next()
                asts, args = [], []
                while not self.accept(")"):
                    mark = self.i
                    node = self.unary_ast() if (self.peek()[0] == "id" and self.peek()[1] not in TYPES) else None
                    if node is not None and self.peek()[1] in (",", ")") and node[0] != "text":
                        asts.append(node)
                        args.append(self.emit_ast(node))
                    else:
                        self.i = mark
                        asts.append(None)
                        args.append(self.expr())
                    self.accept(",")
                if text in SCALAR:
                    return f"_R.vconvert({SCALAR[text]!r}, {args[0]})"
                m = re.fullmatch(r"([iub]?vec)([234])", text)
                if m:
                    return f"_T.vec{m.group(2)}{VEC[m.group(1)]}({', '.join(args)})"
                if text in BUILTINS:
                    return f"{BUILTINS[text]}({', '.join(args)})"
                for k, boxed in enumerate(self.sig.get(text, ())):
                    if boxed:
                        if asts[k] is None:
                            raise Skip("out argument that is not a variable")
                        self.tmp += 1
                        tmp = f"_b{self.tmp}"
                        self.pre.append(f"{tmp} = _R.Box({args[k]})")
                        self.post.append((asts[k], f"{tmp}.v"))
                        args[k] = tmp
                if text not in self.sig:
                    raise Skip(f"call of unknown function {text}")
                return f"{pyname(text)}({', '.join(args)})"
            return self.lookup(text)
        raise Skip(f"unexpected {text!r} in expression")


class Module:
    """cpp-preprocess `entry` (with -D defines) in `directory`, translate, exec.  Uniform block members, textures, samplers and the
    gl_* builtins are plain globals of the namespace, set by the caller."""

    def __init__(self, directory, entry, defines):
        prologue = "".join(f"#define {k} {v}\n" for k, v in defines.items())
        text = open(f"{directory}/{entry}").read()
        pp = subprocess.run(["cpp", "-P", "-undef", "-nostdinc", "-I", directory, "-x", "c", "-"], input=prologue + text, capture_output=True, text=True, check=True).stdout
        self.source = pp
        tr = Translator(pp)
        self.python = tr.module()
        self.skipped = tr.skipped
        self.ns = {"_R": R, "_T": T, "_G": G, "_ONCE": (0,)}
        exec(compile(self.python, f"<glsl:{entry}>", "exec"), self.ns)
