#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + PMC passes for bench.py's workload.
# Usage: tools/profile_gpu.sh <tag> <traffic key or -> [bench args...]   -> gpurun_out/prof_<tag>/ (+ gpurun_out/traffic_<tag>.json)
set -o pipefail
TAG=${1:-r01}; shift
KEY=${1:--}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config-c --no-relaxed --no-other-configs $@"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $BENCH > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" \
            "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $BENCH > $OUT/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; tail -3 $OUT/pmc_$name.log; }
done
python3 tools/summarize_profile.py $OUT $KEY gpurun_out/traffic_$TAG.json $BENCH > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
