"""Weights a slice of disassembly by the gfx950 VALU issue rates measured with tools/ubench/valu_rate.hip
(cycles per wave64 instruction: full rate 2, half rate 4, transcendental 8).   usage: isa_cost.py file first last"""
import re
import sys

HALF = ("v_cmp", "v_cndmask", "v_min_f32", "v_max_f32", "v_med3", "v_cvt", "v_rndne", "v_alignbit", "v_bfi", "v_fract", "v_pk_",
        "v_mad_u64", "v_bitop3", "v_min_u32", "v_max_u32", "v_ffbh", "v_bfe", "v_floor", "v_trunc", "v_ceil", "v_readlane", "v_readfirstlane")
QUARTER = ("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")


def cost(op):
    if op.startswith(QUARTER):
        return 8
    if op.startswith(HALF):
        return 4
    return 2


def main():
    path, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lines = open(path).read().splitlines()[first - 1:last]
    total, classes, salu = 0, {2: 0, 4: 0, 8: 0}, 0
    for ln in lines:
        m = re.match(r"\s*([vs]_[a-z0-9_]+)", ln)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("s_"):
            salu += 1
            continue
        c = cost(op)
        classes[c] += 1
        total += c
    print(f"VALU instructions {sum(classes.values())} (full {classes[2]}, half {classes[4]}, quarter {classes[8]}), SALU {salu}, "
          f"issue cycles {total}")


if __name__ == "__main__":
    main()
