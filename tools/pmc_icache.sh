#!/bin/bash
# Instruction-cache counters of the path-tracing kernel: tools/pmc_icache.sh <tag> [bench args]   (through gpurun)
TAG=$1; shift
OUT=gpurun_out/pmci_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config-c --no-relaxed --no-other-configs $@"
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_a -o pmc -- python3 $BENCH > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
python3 tools/summarize_profile.py $OUT | sed -n '/PMC/,$p' | grep -A8 "^k_regwave\|^k_wfq"
