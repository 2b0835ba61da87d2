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
# The path-tracing stage of a step may be many dispatches that overlap on several streams (queue kernels): their durations
# add up to more than the stage takes.  The comparable figure is the span from the first start to the last end of the
# stage's dispatches between two k_resolve dispatches -- what bench.py's HIP events around the stage measure.
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    spans, lo, hi, n = [], None, None, 0
    for row in rows:
        k = short(row["Kernel_Name"])
        if k.startswith("k_resolve"):
            if lo is not None:
                spans.append(((hi - lo) / 1e6, n))
            lo = hi = None
            n = 0
        elif k.startswith("k_") and not k.startswith(("k_sky_mask", "k_upsample", "k_dbg")):
            a, b = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            lo, hi, n = (a if lo is None else min(lo, a)), (b if hi is None else max(hi, b)), n + 1
    if spans:
        print("== path-tracing stage per step: first start to last end of its dispatches (ms; dispatches) ==")
        print("   " + "  ".join("%.3f (%d)" % s for s in spans) + "   mean of all but the first (warm-up): %.3f" %
              (sum(s[0] for s in spans[1:]) / max(1, len(spans) - 1)))
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

# ---- HBM traffic of the dominant kernel per launch, for bench.py's roofline.traffic (profiles/traffic.json) ----
# MI355X_MICROARCH.md, "HBM": FETCH_SIZE and WRITE_SIZE come from separate --pmc passes (they do not fit one), both in KiB;
# on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so it is doubled; WRITE_SIZE is taken as it is.  The
# doubling is calibrated for wide coalesced reads; this kernel's reads are mostly L2 hits (the scene), so the figure is an
# upper bound for them -- stated in the entry.
if len(sys.argv) > 3 and sys.argv[2] != "-":
    import json
    key, dest = sys.argv[2], sys.argv[3]
    # The path-tracing stage of one bench step is one dispatch of k_regwave / k_wavefront, or many dispatches of the queue kernels
    # (k_wfq_*); everything is reported PER STEP: counter sums over the stage's kernels divided by the number of steps the pass
    # ran, which is the number of k_resolve dispatches (one per step and chunk).
    stage = [k for k in agg if k.startswith("k_") and not k.startswith(("k_resolve", "k_sky_mask", "k_upsample", "k_dbg"))]

    def per_step(counter):
        n_steps = max(1, len(ndisp[("k_resolve", counter)]))
        return sum(agg[k].get(counter, 0.0) for k in stage) / n_steps, n_steps

    if stage and all(any(c in agg[k] for k in stage) for c in ("FETCH_SIZE", "WRITE_SIZE")):
        fetch, write = per_step("FETCH_SIZE")[0] * 1024.0, per_step("WRITE_SIZE")[0] * 1024.0
        best = max(stage, key=lambda k: agg[k].get("SQ_BUSY_CYCLES", 0.0))
        entry = {key: {"bytes_per_launch": round(2.0 * fetch + write), "fetch_size_bytes_raw": round(fetch), "write_size_bytes": round(write),
                       "kernel": best if len(stage) == 1 else " + ".join(sorted(stage)),
                       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `%s`, per %s; "
                                 "FETCH_SIZE doubled (gfx950 counts 128-byte read requests as 64 bytes)" %
                                 (" ".join(sys.argv[4:] or ["bench.py"]), "dispatch" if len(stage) == 1 else "step (all dispatches of the stage's kernels)")}}
        # issue-side figures of the same stage, recomputable from the counters printed above:
        #   lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
        #   wave-cycles waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES
        #   VALU busy = SQ_INSTS_VALU x 3 cycles (mean issue cost of this instruction mix, DESIGN.md "Instruction prices") /
        #               (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs summed over the stage's dispatches
        e = entry[key]
        tv, av = per_step("SQ_THREAD_CYCLES_VALU")[0], per_step("SQ_ACTIVE_INST_VALU")[0]
        wa, wc = per_step("SQ_WAIT_ANY")[0], per_step("SQ_WAVE_CYCLES")[0]
        iv, gui = per_step("SQ_INSTS_VALU")[0], per_step("GRBM_GUI_ACTIVE")[0]
        if av > 0:
            e["lane_utilisation"] = round(tv / (64.0 * av), 4)
        if wc > 0:
            e["wave_cycles_waiting"] = round(wa / wc, 4)
        if gui > 0:
            e["valu_busy"] = round(iv * 3.0 / (1024.0 * gui / 8.0), 4)
            e["valu_insts_per_step"] = round(iv)
        e["counters_source"] = "rocprofv3 --pmc passes of the same command (tools/profile_gpu.sh): lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU); " \
                               "wave_cycles_waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES; valu_busy = SQ_INSTS_VALU x 3 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)"
        json.dump(entry, open(dest, "w"), indent=1)
        print("== traffic entry ==")
        print(json.dumps(entry))
