#!/bin/bash
# A/B of two builds of the library on one box: kernel time + VALU/SALU instruction counts of the path-tracing kernel.
#   tools/pmc_ab.sh <tag> <lib.so> [bench args]      (run through gpurun; IRIS_HIP_LIB selects the build)
TAG=$1; LIB=$2; shift; shift
OUT=gpurun_out/ab_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export IRIS_HIP_LIB=$PWD/$LIB
BENCH="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config-c --no-relaxed $@"
timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config-c --no-relaxed "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', 'Msamples/s', d['value'], 'kernel ms', d['roofline']['kernel_ms_per_launch'])"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_a -o pmc -- python3 $BENCH > $OUT/a.log 2>&1 || tail -3 $OUT/a.log
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_b -o pmc -- python3 $BENCH > $OUT/b.log 2>&1 || tail -3 $OUT/b.log
python3 tools/summarize_profile.py $OUT 2>/dev/null | sed -n '/PMC/,$p' | grep -v resolve | grep -A14 "^k_regwave" | head -16
