#!/usr/bin/env python3
"""Attributes the instructions of one kernel in a `hipcc -S -gline-tables-only` listing to the source
functions they were inlined from (innermost line only).  Static counts; on straight-line shading code
they are close to the dynamic ones.

    python tools/isa_lines.py k.s k_regwaveILi0 iris_amd/csrc/iris_device.h iris_amd/csrc/iris_kernels.hip
"""
import collections
import re
import sys


def function_spans(path):
    spans = []
    for i, l in enumerate(open(path), 1):
        m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:IRIS_DEV|__device__|__global__)\s+.*?(\w+)\s*\(", l)
        if m:
            spans.append((i, m.group(1)))
    return spans


def main():
    asm, key, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    lines = open(asm).read().split("\n")
    fileno = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            fileno[int(m.group(1))] = m.group(2).split("/")[-1]
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"\s+md5', l)
        if m:
            fileno[int(m.group(1))] = m.group(2).split("/")[-1]
    spans = {f.split("/")[-1]: function_spans(f) for f in files}
    start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z][\w.]*:", l) and key in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cur = ("?", 0)
    per_fn = collections.Counter()
    per_line = collections.Counter()
    for l in lines[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (fileno.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not l.startswith("\t") or l.strip().startswith((".", ";")):
            continue
        op = l.split()[0]
        f, ln = cur
        fn = "?"
        for s, name in spans.get(f, []):
            if s <= ln:
                fn = name
        per_fn[(f, fn)] += 1
        per_line[(f, ln)] += 1
    total = sum(per_fn.values())
    print("total", total)
    for (f, fn), n in per_fn.most_common(45):
        print("%6d %5.1f%%  %s:%s" % (n, 100.0 * n / total, f, fn))
    print("-- hottest lines")
    src = {f.split("/")[-1]: open(f).read().split("\n") for f in files}
    for (f, ln), n in per_line.most_common(60):
        text = src[f][ln - 1].strip()[:110] if f in src and 0 < ln <= len(src[f]) else ""
        print("%6d  %s:%d  %s" % (n, f, ln, text))


if __name__ == "__main__":
    main()
