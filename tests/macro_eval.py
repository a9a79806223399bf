"""Evaluates the generated `-D` macro expressions numerically in float64 (pure Python) - used to check the code
generator against finite differences without any compiled code."""
import math
import re

_FUNCS = dict(sin=math.sin, cos=math.cos, tan=math.tan, asin=math.asin, acos=math.acos, atan=math.atan, atan2=math.atan2,
              exp=math.exp, log=math.log, sqrt=math.sqrt, fabs=abs, sinh=math.sinh, cosh=math.cosh, tanh=math.tanh,
              pow=math.pow, fmod=math.fmod, fmin=min, fmax=max, sign=lambda x: (x > 0) - (x < 0))
_LITERAL = re.compile(r"(?<![A-Za-z_0-9])((?:\d+\.\d*|\.\d+|\d+)(?:e[+-]?\d+)?)f\b")
_TERNARY = re.compile(r"\?")


def parse_macros(argument_string):
    out = {}
    for tok in argument_string.split():
        if tok.startswith("-D"):
            k, _, v = tok[2:].partition("=")
            out[k] = v
    return out


def _ternaries(s):
    """C `(c ? a : b)` (always fully parenthesised by the generator) -> Python conditional expression"""
    out, i = [], 0
    while i < len(s):
        if s[i] != "(":
            out.append(s[i])
            i += 1
            continue
        depth, j = 0, i
        while True:
            depth += s[j] == "("
            depth -= s[j] == ")"
            if depth == 0:
                break
            j += 1
        inner = s[i + 1:j]
        depth, q, c = 0, -1, -1
        for k, ch in enumerate(inner):
            depth += ch == "("
            depth -= ch == ")"
            if depth == 0 and ch == "?" and q < 0:
                q = k
            elif depth == 0 and ch == ":" and q >= 0 and c < 0:
                c = k
        if q >= 0:
            out.append("((" + _ternaries(inner[q + 1:c]) + ") if (" + _ternaries(inner[:q]) + ") else (" + _ternaries(inner[c + 1:]) + "))")
        else:
            out.append("(" + _ternaries(inner) + ")")
        i = j + 1
    return "".join(out)


def to_python(expr):
    e = _LITERAL.sub(r"\1", expr).replace("cfg->", "cfg_")
    if _TERNARY.search(e):
        e = _ternaries(e)
    return e


class MacroSet:
    def __init__(self, argument_string):
        self.m = parse_macros(argument_string)
        self.temps = []
        t = self.m.get("TEMPORARIES0", "DUMMY")
        if t != "DUMMY":
            # split "pv0=expr,pv1=expr" at top-level commas
            depth, start, parts = 0, 0, []
            for i, ch in enumerate(t):
                depth += ch == "("
                depth -= ch == ")"
                if ch == "," and depth == 0:
                    parts.append(t[start:i])
                    start = i + 1
            parts.append(t[start:])
            for p in parts:
                name, _, e = p.partition("=")
                self.temps.append((name, compile(to_python(e), name, "eval")))
        self._cache = {}

    def env(self, pos, vel=None, cfg=None, dpos=None):
        env = dict(_FUNCS)
        env.update(v1=pos[0], v2=pos[1], v3=pos[2], v4=pos[3], rs=1.0, c=1.0)
        if vel is not None:
            env.update(iv1=vel[0], iv2=vel[1], iv3=vel[2], iv4=vel[3])
        if dpos is not None:
            env.update(dv1=dpos[0], dv2=dpos[1], dv3=dpos[2], dv4=dpos[3])
        for k, v in (cfg or {}).items():
            env["cfg_" + k] = v
        for name, code in self.temps:
            try:
                env[name] = eval(code, {"__builtins__": {}}, env)
            except (ZeroDivisionError, ValueError, OverflowError):
                env[name] = float("nan")
        return env

    def value(self, name, env):
        if name not in self._cache:
            self._cache[name] = compile(to_python(self.m[name]), name, "eval")
        return eval(self._cache[name], {"__builtins__": {}}, env)

    def has(self, name):
        return name in self.m

    def metric(self, pos, cfg=None):
        """full symmetric 4x4 as nested lists"""
        env = self.env(pos, cfg=cfg)
        g = [[0.0] * 4 for _ in range(4)]
        if self.has("GENERIC_BIG_METRIC"):
            for i in range(4):
                for j in range(i, 4):
                    g[i][j] = g[j][i] = self.value(f"F{i * 4 + j + 1}_I", env)
        else:
            for i in range(4):
                g[i][i] = self.value(f"F{i + 1}_I", env)
        return g

    def partial(self, pos, k, i, j, cfg=None):
        """d g_ij / d v_k from the F*_P macros"""
        env = self.env(pos, cfg=cfg)
        if self.has("GENERIC_BIG_METRIC"):
            a, b = min(i, j), max(i, j)
            return self.value(f"F{k * 16 + a * 4 + b + 1}_P", env)
        if i != j:
            return 0.0
        return self.value(f"F{i * 4 + k + 1}_P", env)   # diagonal layout [var*4 + wrt]

    def accel(self, pos, vel, cfg=None):
        env = self.env(pos, vel=vel, cfg=cfg)
        return [self.value(f"GEO_ACCEL{i}", env) for i in range(4)]
