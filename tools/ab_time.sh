#!/bin/bash
# Kernel time of one build of the library on several workloads: tools/ab_time.sh <tag> <lib.so> "<bench args 1>" "<bench args 2>" ...
TAG=$1; LIB=$2; shift; shift
export IRIS_HIP_LIB=$PWD/$LIB
for ARGS in "$@"; do
  timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config-c --no-relaxed --no-other-configs $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', '$ARGS', 'Msamples/s', d['value'], 'kernel ms', d['roofline']['kernel_ms_per_launch'])" || exit 1
done
