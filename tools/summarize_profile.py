#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into a short text summary (kept under profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


print("== kernel-trace stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-48s calls %6s  total %12.3f ms  avg %12.3f us  %6s%%" % (
            short(row["Name"])[:48], row["Calls"], float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e3, row["Percentage"]))
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k in seen:
            continue
        seen.add(k)
        print("  %-46s VGPR %s accum %s SGPR %s LDS %s scratch %s grid %s wg %s" % (
            k[:46], row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
            row.get("Scratch_Size"), row.get("Grid_Size"), row.get("Workgroup_Size")))
print("== PMC counters per kernel: sum over all dispatches of all passes that collected the counter, and per dispatch ==")
agg = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(set)
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        ndisp[(k, row["Counter_Name"])].add((f, row["Dispatch_Id"]))  # a counter may be collected in several passes
for k, cs in agg.items():
    if not k.startswith("k_"):
        continue
    print(k[:60])
    for c, v in sorted(cs.items()):
        n = len(ndisp[(k, c)])
        print("   %-28s %18.0f   (%d dispatches, %.4g per dispatch)" % (c, v, n, v / max(1, n)))
