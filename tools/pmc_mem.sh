#!/bin/bash
# Memory-pipeline counters (TA / TCP / TD) per kernel and bench step: tools/pmc_mem.sh <tag> [bench args]
TAG=$1; shift
OUT=gpurun_out/pmcm_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config-c --no-relaxed $@"
i=0
for pass in "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TAGRAM0_REQ_sum SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
            "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN2_sum"; do
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
    for name in sorted(agg[k]):
        steps = max(1, len(nd[("k_resolve", name)]))
        print("   %-40s %14.4g per step" % (name, agg[k][name] / steps))
PY
