"""AST (parser.py) -> Python source.  One Python function per WGSL function, value semantics kept by copying structs
and arrays on assignment, vectors immutable.  TEST INFRASTRUCTURE ONLY (see runtime.py)."""
import re

from . import parser

BUILTINS = {
    "min": "_R.w_min", "max": "_R.w_max", "abs": "_R.w_abs", "floor": "_R.w_floor", "ceil": "_R.w_ceil", "round": "_R.w_round", "trunc": "_R.w_trunc",
    "sqrt": "_R.w_sqrt", "fract": "_R.w_fract", "clamp": "_R.w_clamp", "saturate": "_R.w_saturate", "mix": "_R.w_mix", "sign": "_R.w_sign", "step": "_R.w_step",
    "sin": "_R.w_sin", "cos": "_R.w_cos", "exp": "_R.w_exp", "exp2": "_R.w_exp2", "log2": "_R.w_log2", "pow": "_R.w_pow", "inverseSqrt": "_R.w_inverse_sqrt",
    "select": "_R.select", "any": "_R.w_any", "all": "_R.w_all", "dot": "_R.dot", "cross": "_R.cross", "length": "_R.length", "distance": "_R.distance",
    "normalize": "_R.normalize", "reflect": "_R.reflect", "transpose": "_R.transpose",
    "pack2x16float": "_R.pack2x16float", "unpack2x16float": "_R.unpack2x16float", "pack2x16unorm": "_R.pack2x16unorm", "unpack2x16unorm": "_R.unpack2x16unorm",
    "pack4x8snorm": "_R.pack4x8snorm", "unpack4x8snorm": "_R.unpack4x8snorm", "pack4x8unorm": "_R.pack4x8unorm", "unpack4x8unorm": "_R.unpack4x8unorm",
    "textureLoad": "_T.texture_load", "textureStore": "_T.texture_store", "textureDimensions": "_T.texture_dimensions",
    "textureSampleLevel": "_T.texture_sample_level", "textureGather": "_T.texture_gather", "textureNumLevels": "_T.texture_num_levels",
    "arrayLength": "_T.array_length", "dpdx": "_T.dpdx", "dpdy": "_T.dpdy",
}
SCALARS = {"f32", "i32", "u32", "bool"}
PYKEYWORDS = {"lambda", "from", "in", "is", "not", "and", "or", "def", "class", "pass", "global", "del", "with", "as", "import", "yield", "None", "True", "False",
              "try", "except", "raise", "while", "print", "len", "id", "min", "max", "abs", "type", "hash", "input", "iter", "next", "filter", "map", "round"}


def pyname(n):
    return n + "_" if n in PYKEYWORDS else n


class Translator:
    def __init__(self, decls):
        self.decls = decls
        self.structs = {d[1]: d for d in decls if d[0] == "struct"}
        self.aliases = {d[1]: d[2] for d in decls if d[0] == "alias"}
        self.consts = []
        self.lines = []
        self.tmp = 0
        # pointer parameters: a struct pointee is passed by reference, anything else through a _R.Box
        self.box_sig = {}
        for d in decls:
            if d[0] == "fn":
                self.box_sig[d[1]] = [p[1][1] == "ptr" and p[1][2][1][1] not in self.structs for p in d[2]]
        self.box_params = set()
        self.pre, self.post = [], []
        self.scopes = []            # WGSL block scoping: name -> Python name, innermost last (shadowing declarations are renamed)
        self.used_names = set()

    def push_scope(self):
        self.scopes.append({})

    def pop_scope(self):
        self.scopes.pop()

    def declare(self, name):
        py = pyname(name)
        if any(name in sc for sc in self.scopes[:-1]) or (py in self.used_names and name not in self.scopes[-1]):
            k = 1
            while f"{py}__{k}" in self.used_names:
                k += 1
            py = f"{py}__{k}"
        self.scopes[-1][name] = py
        self.used_names.add(py)
        return py

    def lookup(self, name):
        for sc in reversed(self.scopes):
            if name in sc:
                return sc[name]
        return pyname(name)         # module scope

    # ------------------------------------------------------------ types
    def type_expr(self, ty):
        """Python expression that evaluates to the runtime type descriptor (_T.*)"""
        _, name, args = ty
        if name in self.aliases:
            return self.type_expr(self.aliases[name])
        if name in SCALARS:
            return f"_T.Scalar({name!r})"
        m = re.fullmatch(r"vec([234])", name)
        if m:
            return f"_T.Vec({m.group(1)}, {args[0][1]!r})"
        m = re.fullmatch(r"mat([234])x([234])", name)
        if m:
            return f"_T.Mat({m.group(1)}, {m.group(2)})"
        if name == "array":
            n = "None" if len(args) < 2 else str(int(re.sub(r"[iu]$", "", args[1][1]), 0))
            return f"_T.Array({self.type_expr(args[0])}, {n})"
        if name in ("ptr",):
            return self.type_expr(args[1])
        if name in self.structs:
            return f"S_{name}.TYPE"
        return f"_T.Opaque({name!r})"

    def zero_expr(self, ty):
        return f"{self.type_expr(ty)}.zero()"

    def ctor(self, ty, args):
        _, name, targs = ty
        a = ", ".join(args)
        m = re.fullmatch(r"vec([234])", name)
        if m:
            return f"_T.vec{m.group(1)}{targs[0][1]}({a})"
        m = re.fullmatch(r"mat([234])x([234])", name)
        if m:
            return f"_T.mat{m.group(1)}x{m.group(2)}({a})"
        if name == "array":
            return f"[{a}]"
        if name == "bitcast":
            t = targs[0]
            tn = t[2][0][1] if t[1].startswith("vec") else t[1]
            return f"_R.bitcast({tn!r}, {a})"
        raise NotImplementedError(name)

    # ------------------------------------------------------------ expressions
    def const(self, text):
        if text.startswith(("0x", "0X")):
            suffix = text[-1] if text[-1] in "iu" else ""
            value = int(text.rstrip("iu"), 16)
            return f"_R.u32({value})" if suffix == "u" else (f"_R.i32({value})" if suffix == "i" else str(value))
        if re.fullmatch(r"\d+[iu]?", text):
            if text.endswith("u"):
                return self.hoist(f"_R.u32({text[:-1]})")
            if text.endswith("i"):
                return self.hoist(f"_R.i32({text[:-1]})")
            return text                      # abstract int: adapts to the other operand (numpy weak scalar)
        return self.hoist(f"_R.f32({text.rstrip('f')})")

    def hoist(self, expr):
        if expr not in self.consts:
            self.consts.append(expr)
        return f"_k{self.consts.index(expr)}"

    def is_lvalue_path(self, e):
        return e[0] in ("id", "member", "index") or (e[0] == "paren" and self.is_lvalue_path(e[1])) or (e[0] == "un" and e[1] == "*")

    def expr(self, e):
        k = e[0]
        if k == "num":
            return self.const(e[1])
        if k == "bool":
            return "True" if e[1] else "False"
        if k == "id":
            return self.lookup(e[1])
        if k == "paren":
            return f"({self.expr(e[1])})"
        if k == "member":
            return f"{self.expr(e[1])}.{pyname(e[2])}"
        if k == "index":
            return f"{self.expr(e[1])}[{self.expr(e[2])}]"
        if k == "un":
            op, x = e[1], e[2]
            if op == "*" and x[0] == "id" and x[1] in self.box_params:
                return f"{self.lookup(x[1])}.v"
            if op in ("&", "*"):
                return self.expr(x)          # pointers to structs are references to the mutable object
            if op == "!":
                return f"_R.lnot({self.expr(x)})"
            if op == "~":
                return f"(~{self.expr(x)})"
            return f"(-{self.expr(x)})"
        if k == "bin":
            op, a, b = e[1], self.expr(e[2]), self.expr(e[3])
            if op == "&&":
                return f"({a} and {b})"
            if op == "||":
                return f"({a} or {b})"
            if op == "/":
                return f"_R.div({a}, {b})"
            if op == "%":
                return f"_R.mod({a}, {b})"
            if op == "<<":
                return f"_R.shl({a}, {b})"
            if op == ">>":
                return f"_R.shr({a}, {b})"
            return f"({a} {op} {b})"
        if k == "construct":
            return self.ctor(e[1], [self.expr(a) for a in e[2]])
        if k == "call":
            name, args = e[1], [self.expr(a) for a in e[2]]
            if name in SCALARS:
                return f"_R.vconvert({name!r}, {args[0]})"
            if name in self.structs:
                return f"S_{name}.make({', '.join(args)})"
            if name == "workgroupBarrier" or name == "storageBarrier":
                return "(yield)"
            if name in BUILTINS:
                return f"{BUILTINS[name]}({', '.join(args)})"
            for i, boxed in enumerate(self.box_sig.get(name, ())):
                a = e[2][i]
                if boxed and a[0] == "un" and a[1] == "&":       # box the variable for the call, write it back after the statement
                    self.tmp += 1
                    tmp = f"_b{self.tmp}"
                    self.pre.append(f"{tmp} = _R.Box({self.expr(a[2])})")
                    self.post.append((a[2], f"{tmp}.v"))
                    args[i] = tmp
            return f"{pyname(name)}({', '.join(args)})"
        raise NotImplementedError(k)

    def value(self, e):
        """expression used where WGSL copies (initialiser, right-hand side, argument stored by the callee)"""
        s = self.expr(e)
        return f"_R.cp({s})" if self.is_lvalue_path(e) else s

    # ------------------------------------------------------------ statements
    def emit(self, depth, text):
        self.lines.append("    " * depth + text)

    def assign(self, depth, target, value_src):
        """target = value with WGSL lvalue rules (vector components / swizzles of an immutable V are rebuilt)"""
        t = target
        while t[0] == "paren":
            t = t[1]
        if t[0] == "un" and t[1] == "*":      # (*p) = value: overwrite the pointee in place
            if t[2][0] == "id" and t[2][1] in self.box_params:
                self.emit(depth, f"{self.lookup(t[2][1])}.v = {value_src}")
            else:
                self.emit(depth, f"{self.expr(t[2])}.set_value({value_src})")
            return
        if t[0] == "id":
            self.emit(depth, f"{self.lookup(t[1])} = {value_src}")
        elif t[0] == "member":
            base, name = t[1], t[2]
            if re.fullmatch(r"[xyzw]{1,4}|[rgba]{1,4}", name):
                self.tmp += 1
                tmp = f"_t{self.tmp}"
                self.emit(depth, f"{tmp} = {self.expr(base)}")
                self.emit(depth, f"if {tmp}.__class__ is _R.V:")
                self.assign(depth + 1, base, f"{tmp}.with_({name!r}, {value_src})")
                self.emit(depth, "else:")
                self.emit(depth + 1, f"{tmp}.{pyname(name)} = {value_src}")
            else:
                self.emit(depth, f"{self.expr(base)}.{pyname(name)} = {value_src}")
        elif t[0] == "index":
            base, idx = t[1], t[2]
            self.tmp += 1
            tmp = f"_t{self.tmp}"
            self.emit(depth, f"{tmp} = {self.expr(base)}")
            self.emit(depth, f"if {tmp}.__class__ is _R.V:")
            self.assign(depth + 1, base, f"{tmp}.with_index({self.expr(idx)}, {value_src})")
            self.emit(depth, "else:")
            self.emit(depth + 1, f"{tmp}[{self.expr(idx)}] = {value_src}")
        else:
            raise NotImplementedError(f"assignment to {t[0]}")

    def stmt(self, depth, s, loops):
        if s[0] in ("var", "assign", "expr", "return"):
            self.pre, self.post = [], []
            mark = len(self.lines)
            self.stmt_inner(depth, s, loops)
            body = self.lines[mark:]
            del self.lines[mark:]
            for line in self.pre:
                self.emit(depth, line)
            post = self.post
            self.pre, self.post = [], []
            if post and s[0] == "return":
                raise NotImplementedError("return of a call that takes a boxed pointer")
            self.lines.extend(body)
            for target, src in post:
                self.assign(depth, target, src)
            return
        self.stmt_inner(depth, s, loops)

    def stmt_inner(self, depth, s, loops):
        k = s[0]
        if k == "block":
            if not s[1]:
                self.emit(depth, "pass")
            self.push_scope()
            for x in s[1]:
                self.stmt(depth, x, loops)
            self.pop_scope()
        elif k == "var":
            _, kind, name, ty, init = s
            if init is not None:
                src = self.value(init)
                if ty is not None and ty[1] in SCALARS and init[0] == "num":
                    src = f"_R.vconvert({ty[1]!r}, {src})"
            else:
                src = self.zero_expr(ty)
            self.emit(depth, f"{self.declare(name)} = {src}")      # (the initialiser was translated before the name became visible)
        elif k == "assign":
            _, op, target, value = s
            if op == "=":
                self.assign(depth, target, self.value(value))
            else:
                self.assign(depth, target, self.expr(("bin", op[:-1], target, ("paren", value))))
        elif k == "expr":
            self.emit(depth, self.expr(s[1]))
        elif k == "if":
            self.emit(depth, f"if {self.expr(s[1])}:")
            self.stmt(depth + 1, s[2], loops)
            if s[3] is not None:
                self.emit(depth, "else:")
                self.stmt(depth + 1, s[3], loops)
        elif k == "for":
            _, init, cond, update, body = s
            self.push_scope()
            if init is not None:
                self.stmt(depth, init, loops)
            flag = f"_brk{len(loops)}"
            self.emit(depth, f"while {self.expr(cond) if cond is not None else 'True'}:")
            self.emit(depth + 1, f"{flag} = False")
            self.emit(depth + 1, "for _once in _ONCE:")
            self.stmt(depth + 2, body, loops + [flag])
            self.emit(depth + 1, f"if {flag}: break")
            if update is not None:
                self.stmt(depth + 1, update, loops)
            self.pop_scope()
        elif k == "break":
            self.emit(depth, f"{loops[-1]} = True")
            self.emit(depth, "break")
        elif k == "continue":
            self.emit(depth, "continue")
        elif k == "return":
            self.emit(depth, "return" if s[1] is None else f"return {self.value(s[1])}")
        else:
            raise NotImplementedError(k)

    # ------------------------------------------------------------ module
    def module(self):
        out = []
        body = []
        for d in self.decls:
            self.lines = []
            if d[0] == "struct":
                _, name, fields = d
                names = [pyname(f[0]) for f in fields]
                self.emit(0, f"class S_{name}(_T.StructBase):")
                self.emit(1, f"__slots__ = {tuple(names)!r}")
                self.emit(1, f"FIELDS = {tuple(names)!r}")
                self.emit(0, f"S_{name}.TYPE = _T.Struct(S_{name}, [{', '.join('(%r, %s)' % (pyname(f[0]), self.type_expr(f[1])) for f in fields)}])")
            elif d[0] == "const":
                _, name, ty, init = d
                src = self.expr(init)
                if ty is not None and ty[1] in SCALARS:
                    src = f"_R.vconvert({ty[1]!r}, {src})"
                self.emit(0, f"{pyname(name)} = {src}")
            elif d[0] == "gvar":
                _, name, ty, init, space, attrs = d
                if "group" in attrs:
                    self.emit(0, f"RESOURCES[{name!r}] = ({int(attrs['group'][0])}, {int(attrs['binding'][0])}, {self.type_expr(ty)}, {tuple(space)!r})")
                elif "workgroup" in space:
                    self.emit(0, f"WORKGROUP_VARS[{name!r}] = {self.type_expr(ty)}")
                else:
                    self.emit(0, f"{pyname(name)} = {self.value(init) if init is not None else self.zero_expr(ty)}")
            elif d[0] == "fn":
                _, name, params, ret, block, attrs = d
                self.box_params = {p[0] for p, boxed in zip(params, self.box_sig[name]) if boxed}
                self.scopes, self.used_names = [{}], set()
                for prm in params:
                    self.declare(prm[0])
                self.emit(0, f"def {pyname(name)}({', '.join(self.lookup(p[0]) for p in params)}):")
                n0 = len(self.lines)
                used_globals = sorted(self.assigned_globals(block))
                if used_globals:
                    self.emit(1, "global " + ", ".join(used_globals))
                self.stmt(1, block, [])
                if len(self.lines) == n0:
                    self.emit(1, "pass")
                if "compute" in attrs:
                    builtins = {p[0]: p[2].get("builtin", [None])[0] for p in params}
                    self.emit(0, f"ENTRY_POINTS[{name!r}] = ({pyname(name)}, {tuple(int(re.sub('[iu]$', '', a)) for a in attrs.get('workgroup_size', ['1']))!r}, {builtins!r})")
            body.extend(self.lines)
        out.append("# generated by tests/tools/wgsl/translate.py - do not edit")
        for i, c in enumerate(self.consts):
            out.append(f"_k{i} = {c}")
        out.extend(body)
        return "\n".join(out) + "\n"

    def assigned_globals(self, block):
        """module-scope variables a function assigns to (Python needs them declared global)"""
        privates = {pyname(d[1]) for d in self.decls if d[0] == "gvar" and "group" not in d[5]}    # private and workgroup variables
        found = set()

        def walk(s):
            if isinstance(s, tuple):
                if s and s[0] == "assign":
                    t = s[2]
                    while t[0] in ("member", "index", "paren"):
                        t = t[1]
                    if t[0] == "id" and pyname(t[1]) in privates:
                        found.add(pyname(t[1]))
                for x in s:
                    walk(x)
            elif isinstance(s, list):
                for x in s:
                    walk(x)
        walk(block)
        return found


def translate(src):
    return Translator(parser.parse(src)).module()
