"""WGSL (naga 0.10 / bevy 0.9 era) parser for the subset the reference's shaders use: structs, module `let`
constants, resource / private / workgroup variables, functions with `ptr<function, T>` parameters, `var` / `let`,
assignment forms, `if` / `else`, `for`, `break` / `continue` / `return`.  Produces a small tuple AST.
TEST INFRASTRUCTURE ONLY (see runtime.py)."""
import re

GENERIC = {"vec2", "vec3", "vec4", "mat2x2", "mat3x3", "mat4x4", "mat3x4", "mat4x3", "array", "ptr", "bitcast", "binding_array", "texture_2d",
           "texture_storage_2d", "texture_2d_array", "atomic"}
TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>0[xX][0-9a-fA-F]+[iu]?|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fiu]?)
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>->|&&|\|\||==|!=|<=|>=|<<=|>>=|<<|>>|\+=|-=|\*=|/=|%=|&=|\|=|\^=|\+\+|--|[-+*/%&|^~!<>=.,;:(){}\[\]@])
""", re.X | re.S)


def tokenize(src):
    out, pos = [], 0
    while pos < len(src):
        m = TOKEN.match(src, pos)
        if not m:
            raise SyntaxError(f"bad character {src[pos]!r} at {pos}: {src[pos:pos + 40]!r}")
        pos = m.end()
        kind = m.lastgroup
        if kind != "ws":
            out.append((kind, m.group(kind)))
    out.append(("eof", ""))
    return out


class Parser:
    def __init__(self, src):
        self.t = tokenize(src)
        self.i = 0

    # ------------------------------------------------------------ token helpers
    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def at(self, text):
        return self.t[self.i][1] == text and self.t[self.i][0] != "num"

    def accept(self, text):
        if self.at(text):
            self.i += 1
            return True
        return False

    def expect(self, text):
        if not self.accept(text):
            ctx = " ".join(t[1] for t in self.t[max(0, self.i - 8):self.i + 4])
            raise SyntaxError(f"expected {text!r}, got {self.peek()[1]!r} near: {ctx}")

    def ident(self):
        kind, text = self.next()
        if kind != "id":
            raise SyntaxError(f"expected identifier, got {text!r}")
        return text

    def close_angle(self):
        """consume one '>' even if the tokenizer glued two of them (array<vec4<f32>>)"""
        kind, text = self.peek()
        if text == ">":
            self.i += 1
        elif text == ">>":
            self.t[self.i] = ("op", ">")
        elif text == ">=":
            self.t[self.i] = ("op", "=")
        else:
            raise SyntaxError(f"expected '>', got {text!r}")

    # ------------------------------------------------------------ types
    def type(self):
        name = self.ident()
        args = []
        if self.at("<"):
            self.next()
            while True:
                if self.peek()[0] == "num":
                    args.append(("num", self.next()[1]))
                else:
                    args.append(self.type())
                if not self.accept(","):
                    break
            self.close_angle()
        return ("type", name, tuple(args))

    def attributes(self):
        attrs = {}
        while self.accept("@"):
            name = self.ident()
            args = []
            if self.accept("("):
                while not self.at(")"):
                    args.append(self.next()[1])
                    self.accept(",")
                self.expect(")")
            attrs[name] = args
        return attrs

    # ------------------------------------------------------------ module
    def module(self):
        decls = []
        while self.peek()[0] != "eof":
            if self.accept(";"):
                continue
            attrs = self.attributes()
            if self.at("struct"):
                self.next()
                name = self.ident()
                self.expect("{")
                fields = []
                while not self.at("}"):
                    self.attributes()
                    fname = self.ident()
                    self.expect(":")
                    fields.append((fname, self.type()))
                    if not self.accept(","):
                        self.accept(";")
                self.expect("}")
                decls.append(("struct", name, fields))
            elif self.at("var"):
                self.next()
                space = []
                if self.accept("<"):
                    while not self.at(">"):
                        space.append(self.next()[1])
                        self.accept(",")
                    self.close_angle()
                name = self.ident()
                ty = None
                if self.accept(":"):
                    ty = self.type()
                init = self.expr() if self.accept("=") else None
                self.expect(";")
                decls.append(("gvar", name, ty, init, tuple(space), attrs))
            elif self.at("let") or self.at("const"):
                self.next()
                name = self.ident()
                ty = self.type() if self.accept(":") else None
                self.expect("=")
                init = self.expr()
                self.expect(";")
                decls.append(("const", name, ty, init))
            elif self.at("type"):
                self.next()
                name = self.ident()
                self.expect("=")
                decls.append(("alias", name, self.type()))
                self.accept(";")
            elif self.at("fn"):
                self.next()
                name = self.ident()
                self.expect("(")
                params = []
                while not self.at(")"):
                    pattrs = self.attributes()
                    pname = self.ident()
                    self.expect(":")
                    params.append((pname, self.type(), pattrs))
                    self.accept(",")
                self.expect(")")
                ret = None
                if self.accept("->"):
                    self.attributes()
                    ret = self.type()
                decls.append(("fn", name, params, ret, self.block(), attrs))
            else:
                raise SyntaxError(f"unexpected {self.peek()[1]!r} at module scope")
        return decls

    # ------------------------------------------------------------ statements
    def block(self):
        self.expect("{")
        stmts = []
        while not self.at("}"):
            stmts.append(self.statement())
        self.expect("}")
        return ("block", stmts)

    def statement(self):
        if self.at("{"):
            return self.block()
        if self.accept(";"):
            return ("block", [])
        if self.at("var") or self.at("let"):
            s = self.var_statement()
            self.expect(";")
            return s
        if self.accept("if"):
            return self.if_rest()
        if self.accept("for"):
            self.expect("(")
            init = None
            if not self.at(";"):
                init = self.var_statement() if (self.at("var") or self.at("let")) else self.simple_statement()
            self.expect(";")
            cond = None if self.at(";") else self.expr()
            self.expect(";")
            update = None if self.at(")") else self.simple_statement()
            self.expect(")")
            return ("for", init, cond, update, self.block())
        if self.accept("return"):
            value = None if self.at(";") else self.expr()
            self.expect(";")
            return ("return", value)
        if self.accept("break"):
            self.expect(";")
            return ("break",)
        if self.accept("continue"):
            self.expect(";")
            return ("continue",)
        s = self.simple_statement()
        self.expect(";")
        return s

    def if_rest(self):
        cond = self.expr()
        then = self.block()
        other = None
        if self.accept("else"):
            other = ("block", [self.if_rest()]) if self.accept("if") else self.block()
        return ("if", cond, then, other)

    def var_statement(self):
        kind = self.next()[1]
        if self.at("<"):        # var<function>
            while not self.at(">"):
                self.next()
            self.close_angle()
        name = self.ident()
        ty = self.type() if self.accept(":") else None
        init = self.expr() if self.accept("=") else None
        return ("var", kind, name, ty, init)

    def simple_statement(self):
        if self.at("_"):
            self.next()
            self.expect("=")
            return ("expr", self.expr())
        lhs = self.unary()
        kind, text = self.peek()
        if text in ("=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>="):
            self.next()
            return ("assign", text, lhs, self.expr())
        if text in ("++", "--"):
            self.next()
            return ("assign", text[0] + "=", lhs, ("num", "1"))
        return ("expr", lhs)

    # ------------------------------------------------------------ expressions
    LEVELS = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"), ("+", "-"), ("*", "/", "%")]

    def expr(self, level=0):
        if level == len(self.LEVELS):
            return self.unary()
        lhs = self.expr(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[level]:
            op = self.next()[1]
            lhs = ("bin", op, lhs, self.expr(level + 1))
        return lhs

    def unary(self):
        kind, text = self.peek()
        if kind == "op" and text in ("-", "!", "~", "&", "*"):
            self.next()
            return ("un", text, self.unary())
        return self.postfix(self.primary())

    def postfix(self, e):
        while True:
            if self.accept("."):
                e = ("member", e, self.ident())
            elif self.accept("["):
                idx = self.expr()
                self.expect("]")
                e = ("index", e, idx)
            else:
                return e

    def call_args(self):
        self.expect("(")
        args = []
        while not self.at(")"):
            args.append(self.expr())
            self.accept(",")
        self.expect(")")
        return args

    def primary(self):
        kind, text = self.peek()
        if kind == "num":
            self.next()
            return ("num", text)
        if text == "(":
            self.next()
            e = self.expr()
            self.expect(")")
            return ("paren", e)
        if kind == "id":
            if text in ("true", "false"):
                self.next()
                return ("bool", text == "true")
            if text in GENERIC and self.peek(1)[1] == "<":
                ty = self.type()
                return ("construct", ty, self.call_args())
            self.next()
            if self.at("("):
                return ("call", text, self.call_args())
            return ("id", text)
        raise SyntaxError(f"unexpected {text!r} in expression")


def parse(src):
    return Parser(src).module()
