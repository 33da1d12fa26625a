#!/usr/bin/env python3
"""tools/extract_scene_literals.py — pins for host/scenes.cpp from the reference's scene authoring code.

    python tools/extract_scene_literals.py [/root/reference/src/main.rs] > tests/golden/scene_literals.json

Runs in the BUILD container only (the reference is not on the GPU box).  Reads the `init_scene_*` functions of the reference's main.rs with a
small expression parser for the Rust subset they use (struct literals, path calls, method calls, + - * /, `as f64`, `x.to_radians()`),
EVALUATES the constant expressions (`2.0 + radius` with `let radius = 0.6;` becomes 2.6) and writes DATA ONLY: numbers, enum tags
(surface type, lens shape, element kind), asset file names, matrix operation names with their arguments, loop bounds and gen_range ranges.
No source text is kept.  tests/test_host_layer.py compares the result with the hr_scene_desc that host/scenes.cpp builds.

Per scene:
  camera     eye, target, up (as written, before .normalize()), fov, lens, aperture, focus      (Camera::new arguments, camera.rs:45-64)
  seed       the ISAAC-64 seed words of the scene's generator, or null
  fixed      the elements of the `elements: vec![...]` literal, in order
  added      `scene.add(...)` statements outside loops, in order of appearance
  loops      every `while count < N` placement loop in order: N, the (lo, hi) of its gen_range calls in evaluation order, the element kind
             and the constant parts of its material
  skybox     directory of the six faces, intensity (Skybox::one = (1, 1, 1))
  order      the order in which fixed / added / loop elements join Scene.elements ("fixed", "add:k", "loop:k")
"""
import json
import math
import re
import sys


class Sym:
    """A value only known at run time (a gen_range draw, the loop counter), as an expression tree the test can evaluate:
    ["draw", k] = the k-th gen_range of the loop body (evaluation order), ["count"], ["+" | "-" | "*" | "/", a, b], ["neg", a],
    ["to_radians", a]; leaves are numbers."""
    def __init__(self, tree):
        self.tree = tree

    @staticmethod
    def t(v):
        return v.tree if isinstance(v, Sym) else float(v)

    def __add__(self, o): return Sym(["+", self.tree, Sym.t(o)])
    def __radd__(self, o): return Sym(["+", Sym.t(o), self.tree])
    def __sub__(self, o): return Sym(["-", self.tree, Sym.t(o)])
    def __rsub__(self, o): return Sym(["-", Sym.t(o), self.tree])
    def __mul__(self, o): return Sym(["*", self.tree, Sym.t(o)])
    def __rmul__(self, o): return Sym(["*", Sym.t(o), self.tree])
    def __truediv__(self, o): return Sym(["/", self.tree, Sym.t(o)])
    def __rtruediv__(self, o): return Sym(["/", Sym.t(o), self.tree])
    def __neg__(self): return Sym(["neg", self.tree])


TOKEN = re.compile(r'\s*(?:(\d+\.\d*(?:[eE][-+]?\d+)?|\d+)|("(?:[^"\\]|\\.)*")|([A-Za-z_][A-Za-z0-9_]*)|(::|\.\.|==|[-+*/%(){}\[\],.:;&!<>=]))')


def tokenize(s):
    out, i = [], 0
    s = s.strip()
    while i < len(s):
        m = TOKEN.match(s, i)
        if not m:
            raise ValueError("cannot tokenize at %r" % s[i:i + 40])
        i = m.end()
        if m.group(1) is not None:
            out.append(("num", m.group(1)))
        elif m.group(2) is not None:
            out.append(("str", m.group(2)[1:-1]))
        elif m.group(3) is not None:
            out.append(("id", m.group(3)))
        else:
            out.append(("op", m.group(4)))
    return out


class Parser:
    def __init__(self, toks, env, draws):
        self.t, self.i, self.env, self.draws = toks, 0, env, draws

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", "")

    def eat(self, v=None):
        tok = self.peek()
        if v is not None and tok[1] != v:
            raise ValueError("expected %r, got %r at %d" % (v, tok, self.i))
        self.i += 1
        return tok

    def expr(self):
        v = self.additive()
        if self.peek()[1] == "==":             # `i % 2 == 0` in an if-expression
            self.eat()
            return v == self.additive()
        return v

    def additive(self):
        v = self.term()
        while self.peek()[1] in ("+", "-"):
            op = self.eat()[1]
            r = self.term()
            if isinstance(v, tuple) and v[0] == "vec" and isinstance(r, tuple) and r[0] == "vec":
                v = ("vec", [a + b if op == "+" else a - b for a, b in zip(v[1], r[1])])
            else:
                v = v + r if op == "+" else v - r
        return v

    def term(self):
        v = self.unary()
        while self.peek()[1] in ("*", "/", "%"):
            op = self.eat()[1]
            r = self.unary()
            if op == "%":
                v = math.fmod(v, r)
                continue
            if isinstance(v, tuple) and v[0] == "matrix" and isinstance(r, tuple) and r[0] == "matrix":
                v = ("matrix", v[1] + r[1])       # a product of Matrix44 factors: kept as the list of factors, in order
            elif isinstance(v, tuple) and v[0] == "vec":
                v = ("vec", [x * r if op == "*" else x / r for x in v[1]])      # Vector3 * f64
            elif isinstance(v, tuple) and v[0] == "hsv" and op == "*":
                v = ("hsv", v[1], (v[2] if len(v) > 2 else 1.0) * r)             # hsv_to_rgb(..) * f64
            else:
                v = v * r if op == "*" else v / r
        return v

    def unary(self):
        if self.peek()[1] == "-":
            self.eat()
            return -self.unary()
        if self.peek()[1] == "&":
            self.eat()
            return self.unary()
        return self.postfix(self.primary())

    def args(self, close=")"):
        out = []
        while self.peek()[1] != close:
            out.append(self.expr())
            if self.peek()[1] == ",":
                self.eat()
        self.eat(close)
        return out

    def postfix(self, v):
        while True:
            tok = self.peek()
            if tok[1] == "." and self.peek(1)[0] == "id" and self.peek(2)[1] != "(":      # field access: camera.eye
                self.eat()
                name = self.eat()[1]
                v = v[2][name] if isinstance(v, tuple) and v[0] == "struct" and name in v[2] else Sym(["unknown", name])
            elif tok[1] == "." and self.peek(1)[0] == "id":
                self.eat()
                name = self.eat()[1]
                self.eat("(")
                a = self.args()
                if name == "to_radians":
                    v = Sym(["to_radians", v.tree]) if isinstance(v, Sym) else v * (math.pi / 180.0)   # f64::to_radians: x * (PI / 180)
                elif name == "normalize":
                    v = ("normalize", v)
                elif name in ("sin", "cos", "fract") and not isinstance(v, Sym):
                    v = math.sin(v) if name == "sin" else math.cos(v) if name == "cos" else v - math.floor(v)
                elif name == "gen_range":
                    self.draws.append([a[0], a[1]])
                    v = Sym(["draw", len(self.draws) - 1])
                else:
                    v = Sym(["unknown", name])
            elif tok[1] == "as":
                self.eat()
                self.eat()
            else:
                return v

    def primary(self):
        kind, val = self.peek()
        if kind == "num":
            self.eat()
            # `180.0.to_radians()`: the tokenizer may have swallowed the method's dot into the number ("180.0." never happens: \d+\.\d* stops at the second dot)
            return float(val)
        if kind == "str":
            self.eat()
            return val
        if val == "(":
            self.eat()
            v = self.expr()
            self.eat(")")
            return v
        if kind == "id" and val == "if":       # if cond { a } else { b }: both arms are parsed, the condition (a constant here) picks one
            self.eat()
            cond = self.expr()
            self.eat("{")
            a = self.expr()
            self.eat("}")
            self.eat("else")
            self.eat("{")
            b = self.expr()
            self.eat("}")
            return a if cond else b
        if kind == "id":
            path = [self.eat()[1]]
            while self.peek()[1] == "::":
                self.eat()
                path.append(self.eat()[1])
            name = "::".join(path)
            if self.peek()[1] == "!" and path == ["vec"]:
                self.eat()
                self.eat("[")
                return self.args("]")
            if self.peek()[1] == "(":
                self.eat()
                return self.call(name, self.args())
            if self.peek()[1] == "{" and path[0][0].isupper():
                self.eat()
                fields = {}
                while self.peek()[1] != "}":
                    f = self.eat()[1]
                    self.eat(":")
                    fields[f] = self.expr()
                    if self.peek()[1] == ",":
                        self.eat()
                self.eat("}")
                return ("struct", name, fields)
            if name in self.env:
                return self.env[name]
            if name == "config::PI":
                return math.pi
            if name == "config::PI2":
                return 2.0 * math.pi
            if name.startswith(("SurfaceType::", "LensShape::")):
                return ("struct", name, {})
            return Sym(["unknown", name])
        raise ValueError("unexpected token %r" % (self.peek(),))

    def call(self, name, a):
        if name in ("Vector3::new", "Color::new"):
            return ("vec", a)
        if name in ("Vector3::from_one", "Color::from_one"):
            return ("vec", [a[0]] * 3)
        if name == "Vector3::zero":
            return ("vec", [0.0, 0.0, 0.0])
        if name == "Color::one":
            return ("vec", [1.0, 1.0, 1.0])
        if name.split("::")[-1] == "hsv_to_rgb":
            return ("hsv", a[0][1])
        if name.startswith("Matrix44::"):
            return ("matrix", [[name.split("::")[1]] + a])
        if name == "Box::new":
            return a[0]
        if name in ("BvhMesh::from_mesh",):
            return a[0]
        if name == "ObjLoader::load":
            return ("mesh", a[0], a[1], a[2])
        if name.startswith("Texture::"):
            return ("tex", name.split("::")[1], a)
        if name.startswith("Skybox::"):
            return ("sky", name.split("::")[1], a)
        return ("call", name, a)


def num(v):
    return v.tree if isinstance(v, Sym) else float(v)


def vec(v):
    if isinstance(v, tuple) and v[0] == "normalize":
        v = v[1]
    assert v[0] == "vec", v
    return [num(x) for x in v[1]]


def tex(t):
    kind, a = t[1], t[2]
    if kind in ("white", "black"):
        return {"t": kind}
    if kind == "from_color":
        if a[0][0] == "hsv":
            d = {"t": "hsv", "hsv": [num(x) for x in a[0][1]]}
            if len(a[0]) > 2:
                d["scale"] = num(a[0][2])
            return d
        return {"t": "color", "color": vec(a[0])}
    if kind == "from_path":
        return {"t": "image", "image": a[0]}
    if kind == "new":
        return {"t": "image", "image": a[0], "color": vec(a[1])}
    raise ValueError(kind)


def material(m):
    f = m[2]
    s = f["surface"]
    tag = s[1].split("::")[1]
    param = None
    for k in ("f0", "refractive_index"):
        if k in s[2]:
            param = num(s[2][k])
    return {"surface": tag, "param": param, "albedo": tex(f["albedo"]), "emission": tex(f["emission"]), "roughness": tex(f["roughness"])}


def element(e):
    if e[0] == "struct" and e[1] == "Sphere":
        return {"kind": "sphere", "center": vec(e[2]["center"]), "radius": num(e[2]["radius"]), "material": material(e[2]["material"])}
    if e[0] == "struct" and e[1] == "Cuboid":
        bb = e[2]["aabb"][2]
        return {"kind": "cuboid", "min": vec(bb["min"]), "max": vec(bb["max"]), "material": material(e[2]["material"])}
    if e[0] == "mesh":
        ops = [[op[0]] + [num(x) for x in op[1:]] for op in e[2][1]] if isinstance(e[2], tuple) and e[2][0] == "matrix" else None
        return {"kind": "mesh", "model": e[1], "matrix": ops, "material": material(e[3])}
    raise ValueError("unknown element %r" % (e[:2],))


def balanced(s, start, open_ch, close_ch):
    depth, i, in_str = 0, start, False
    while i < len(s):
        c = s[i]
        if c == '"':
            in_str = not in_str
        elif not in_str:
            if c == open_ch:
                depth += 1
            elif c == close_ch:
                depth -= 1
                if depth == 0:
                    return i
        i += 1
    raise ValueError("unbalanced")


def parse_expr(text, env, draws=None):
    d = [] if draws is None else draws
    p = Parser(tokenize(text), env, d)
    v = p.expr()
    return v


def extract_scene(body):
    out = {"seed": None}
    m = re.search(r"let seed[^=]*=\s*&\[([^\]]*)\]", body)
    if m:
        out["seed"] = [int(x) for x in m.group(1).split(",")]
    # top-level numeric lets (before the scene literal): `let radius = 0.6;`
    env = {}
    cam_at = body.index("Camera::new(")
    cam_end = balanced(body, cam_at + len("Camera::new"), "(", ")")
    scene_at = body.index("Scene {")
    scene_end = balanced(body, scene_at + len("Scene "), "{", "}")
    for m in re.finditer(r"let (?:mut )?(\w+)(?::\s*\w+)?\s*=\s*([^;{}]+);", body[:scene_at]):
        if m.start() > cam_at and m.start() < cam_end:
            continue
        if m.group(1) in ("seed", "rng", "camera", "scene"):
            continue
        try:
            v = parse_expr(m.group(2), env)
            if isinstance(v, float):
                env[m.group(1)] = v
        except ValueError:
            pass
    p = Parser(tokenize(body[cam_at + len("Camera::new("):cam_end] + ")"), env, [])
    a = p.args(")")
    out["camera"] = {"eye": vec(a[0]), "target": vec(a[1]), "up": vec(a[2]), "fov": num(a[3]), "lens": a[4][1].split("::")[1], "aperture": num(a[5]), "focus": num(a[6])}
    # `camera.eye` / `camera.forward` in element expressions (camera.rs:45-50: forward = (target - eye).normalize())
    eye, tgt = out["camera"]["eye"], out["camera"]["target"]
    fw = [t - e for t, e in zip(tgt, eye)]
    ln = math.sqrt(sum(x * x for x in fw))
    env["camera"] = ("struct", "Camera", {"eye": ("vec", list(eye)), "forward": ("vec", [x / ln for x in fw])})
    sc = parse_expr(body[scene_at:scene_end + 1], env)
    out["fixed"] = [element(e) for e in sc[2]["elements"]]
    sky = sc[2]["skybox"]
    faces = [x for x in sky[2] if isinstance(x, str)]
    inten = [x for x in sky[2] if not isinstance(x, str)]
    out["skybox"] = {"dir": faces[0].rsplit("/", 1)[0], "faces": [f.rsplit("/", 1)[1] for f in faces],
                     "intensity": vec(inten[0]) if inten else [1.0, 1.0, 1.0]}
    # after the literal: `scene.add(...)` statements and `while count < N { ... }` placement loops, in order
    rest = body[scene_end + 1:]
    out["added"], out["loops"], out["order"] = [], [], ["fixed"]
    i = 0
    while True:
        ma = re.compile(r"scene\.add\(").search(rest, i)
        mw = re.compile(r"while count < (\d+)\s*\{").search(rest, i)
        mc = re.compile(r"let count = (\d+);\s*while i < count\s*\{").search(rest, i)     # a counted loop (no draws): unrolled here
        cands = [m for m in (ma, mw, mc) if m]
        if not cands:
            break
        m = min(cands, key=lambda x: x.start())
        if m is mc:
            end = balanced(rest, m.end() - 1, "{", "}")
            blk = rest[m.end():end]
            call = blk.index("scene.add(")
            cend = balanced(blk, call + len("scene.add"), "(", ")")
            for it in range(int(m.group(1))):
                lenv = dict(env)
                lenv["i"], lenv["count"] = float(it), float(m.group(1))
                for lm in re.finditer(r"let (\w+)\s*=\s*([^;]+);", blk[:call]):
                    lenv[lm.group(1)] = parse_expr(lm.group(2), lenv)
                out["added"].append(element(parse_expr(blk[call + len("scene.add("):cend], lenv)))
                out["order"].append("add:%d" % (len(out["added"]) - 1))
            i = end
        elif m is ma:
            end = balanced(rest, m.end() - 1, "(", ")")
            out["added"].append(element(parse_expr(rest[m.end():end], env)))
            out["order"].append("add:%d" % (len(out["added"]) - 1))
            i = end
        else:
            end = balanced(rest, m.end() - 1, "{", "}")
            blk = rest[m.end():end]
            lenv, draws = dict(env), []
            lenv["count"] = Sym(["count"])
            call = blk.index("add_with_check_collisions(")
            for lm in re.finditer(r"let (\w+)\s*=\s*([^;]+);", blk[:call]):
                lenv[lm.group(1)] = parse_expr(lm.group(2), lenv, draws)
            cend = balanced(blk, call + len("add_with_check_collisions"), "(", ")")
            el = element(parse_expr(blk[call + len("add_with_check_collisions("):cend], lenv, draws))
            out["loops"].append({"count": int(m.group(1)), "draws": [[num(a), num(b)] for a, b in draws], "element": el})
            out["order"].append("loop:%d" % (len(out["loops"]) - 1))
            i = end
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/main.rs"
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {"_note": "numbers and enum tags extracted from the reference's scene builders by tools/extract_scene_literals.py (constant expressions evaluated; "
                    "null = a value only known at run time: a gen_range draw or the loop counter); data only, no source text"}
    for m in re.finditer(r"fn init_scene_(\w+)\(\)[^{]*\{", src):
        name = m.group(1)
        end = balanced(src, m.end() - 1, "{", "}")
        try:
            out[name] = extract_scene(src[m.end():end])
        except (ValueError, KeyError, AssertionError, TypeError, IndexError) as e:
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    json.dump(out, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
