"""TEST INFRASTRUCTURE ONLY.  Builds the CPU oracle oracle/restate.cpp for one macro string:
`g++ -O2 @macros.rsp restate.cpp -shared` -> oracle/_build/librestate_<hash>.so (git-ignored).

Unlike oracle/_ref (the reference's cl.cl, container only) this travels: g++ exists on the GPU box, so
bench.py's cpu_baseline leg and __graft_entry__.smoke() can build and run it there.
"""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
SRC = os.path.join(HERE, "restate.cpp")


def build(argument_string, opt="-O2"):
    os.makedirs(OUT, exist_ok=True)
    with open(SRC, "rb") as f:
        src = f.read()
    key = hashlib.sha1(src + b"\0" + argument_string.encode() + b"\0" + opt.encode()).hexdigest()[:16]
    so = os.path.join(OUT, f"librestate_{key}.so")
    if os.path.exists(so):
        return so
    rsp = os.path.join(OUT, f"{key}.rsp")
    with open(rsp, "w") as f:
        for tok in argument_string.split():
            if tok.startswith("-D"):
                f.write('"' + tok.replace("\\", "\\\\").replace('"', '\\"') + '"\n')
    tmp = so + f".tmp{os.getpid()}"
    # no -ffast-math: NaN/Inf tests must stay meaningful (IS_DEGENERATE, cl.cl:68)
    subprocess.check_call(["g++", "-std=c++17", opt, "-fPIC", "-shared", "-w", "-ffp-contract=off", "@" + rsp, SRC, "-o", tmp,
                           "-lpthread"])
    os.replace(tmp, so)
    return so
