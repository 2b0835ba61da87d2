#!/usr/bin/env python3
"""Static instruction mix of one function in a hipcc -S assembly listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --offload-device-only -S -o k.s iris_kernels.hip
    python tools/isa_mix.py k.s k_regwaveILi0
"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z][\w.]*:", l) and key in l and not l.startswith(".L"))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ins = [l.split()[0] for l in lines[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    groups = collections.Counter()
    for k in ins:
        if k.startswith(("v_div", "v_rcp")): g = "div(v_div*/v_rcp)"
        elif k.startswith(("v_sqrt", "v_rsq")): g = "sqrt"
        elif "_f64" in k: g = "f64"
        elif k.startswith("v_"): g = "valu other"
        elif k.startswith("s_"): g = "salu"
        elif k.startswith("ds_"): g = "lds"
        elif k.startswith(("global", "buffer", "flat", "scratch")): g = "vmem"
        else: g = "other"
        groups[g] += 1
    print(lines[start].split(":")[0], "instructions:", len(ins))
    for g, n in groups.most_common():
        print("  %-20s %6d" % (g, n))
    c = collections.Counter(ins)
    print("  top:", ", ".join("%s %d" % kv for kv in c.most_common(25)))


if __name__ == "__main__":
    main()
