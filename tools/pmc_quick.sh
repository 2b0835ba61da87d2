#!/bin/bash
# quick two-pass PMC comparison: tools/pmc_quick.sh <tag> [bench args]
TAG=$1; shift
OUT=gpurun_out/pmcq_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="bench.py --steps 1 --warmup 1 --no-cpu-baseline --spp 256 $@"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_a -o pmc -- python3 $BENCH > $OUT/a.log 2>&1 || tail -3 $OUT/a.log
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_b -o pmc -- python3 $BENCH > $OUT/b.log 2>&1 || tail -3 $OUT/b.log
python3 tools/summarize_profile.py $OUT | sed -n '/PMC/,$p' | grep -v resolve | grep -B1 -A14 "^k_" | grep -v "^k_resolve"
