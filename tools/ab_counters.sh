#!/bin/bash
# Work counters and kernel time of one build on several workloads: tools/ab_counters.sh <lib.so> "<bench args>" ...
LIB=$1; shift
export IRIS_HIP_LIB=$PWD/$LIB
for ARGS in "$@"; do
  timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config-c --no-relaxed $ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['counters_per_step']; ms=d['roofline']['kernel_ms_per_launch']
print('$ARGS', 'kernel ms', ms, 'bounces/sample %.3f rays/sample %.3f tests/ray %.2f ns/bounce(256CU) %.4f frac %.4f' % (c['bounces']/c['samples'], c['rays']/c['samples'], c['sphere_tests']/c['rays'], ms*1e6/c['bounces'], d['roofline']['frac']))" || exit 1
done
