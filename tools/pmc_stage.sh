#!/bin/bash
# Per-kernel PMC totals PER BENCH STEP (sum over a kernel's dispatches / number of k_resolve dispatches): tools/pmc_stage.sh <tag> [bench args]
# (run through gpurun; IRIS_HIP_LIB selects a build)
TAG=$1; shift
OUT=gpurun_out/pmcs_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config-c --no-relaxed $@"
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$i -o pmc -- python3 $BENCH > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float)); nd = defaultdict(set)
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add((f, r["Dispatch_Id"]))
for k in sorted(agg):
    if not k.startswith("k_") or k.startswith(("k_sky", "k_resolve")): continue
    print(k)
    c = agg[k]
    for name in sorted(c):
        steps = max(1, len(nd[("k_resolve", name)]))
        print("   %-24s %14.4g per step  (%d dispatches per step)" % (name, c[name] / steps, len(nd[(k, name)]) / steps))
    steps = lambda n: max(1, len(nd[("k_resolve", n)]))
    if c.get("SQ_ACTIVE_INST_VALU"): print("   lane utilisation %.3f" % (c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])))
    if c.get("SQ_WAVE_CYCLES"): print("   waiting %.3f   waves/SIMD while busy %.2f" % (c["SQ_WAIT_ANY"] / steps("SQ_WAIT_ANY") / (c["SQ_WAVE_CYCLES"] / steps("SQ_WAVE_CYCLES")), 0.0))
    if c.get("GRBM_GUI_ACTIVE"):
        cyc = c["GRBM_GUI_ACTIVE"] / steps("GRBM_GUI_ACTIVE") / 8.0
        print("   kernel cycles per step %.4g (%.2f ms at 2.4 GHz); VALU busy (x3 cycles) %.3f; mean waves/SIMD %.2f" % (
            cyc, cyc / 2.4e6, c["SQ_INSTS_VALU"] / steps("SQ_INSTS_VALU") * 3 / (1024 * cyc), c["SQ_WAVE_CYCLES"] / steps("SQ_WAVE_CYCLES") * 4 / (1024 * cyc)))
PY
