#!/usr/bin/env python3
"""Static instruction histogram per kernel of a gfx950 .s file (hipcc -S --cuda-device-only)."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for idx, (i, name) in enumerate(starts):
    ins = []
    for l in lines[i + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        t = l.strip()
        if not l.startswith("\t") or t.startswith(".") or t.startswith(";") or not t:
            continue
        ins.append(t.split()[0])
    c = collections.Counter(ins)
    key = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    short = re.sub(r"_ZN8cddp_dev\d+", "", name)[:64]
    print("%-64s total %5d fma64 %4d mul64 %4d add64 %4d ld %3d st %3d wait %3d mov %4d acc %4d div %3d rcp %3d cnd %4d salu %4d br %3d" % (
        short, len(ins), c["v_fma_f64"], key("v_mul_f64"), key("v_add_f64"), key("global_load"), key("global_store"),
        c["s_waitcnt"], key("v_mov"), key("v_accvgpr"), key("v_div"), key("v_rcp"), key("v_cndmask"),
        sum(v for k, v in c.items() if k.startswith("s_")), key("s_cbranch") + key("s_branch")))
