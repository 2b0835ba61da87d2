#!/bin/bash
# Timing sweep of the queue kernels on config D (run on the GPU box): tools/wfq_sweep.sh "<opt sets separated by ;>"
set -o pipefail
IFS=';' read -ra SETS <<< "$1"
for s in "${SETS[@]}"; do
  args=""
  for kv in $s; do args="$args --opt $kv"; done
  timeout -k 10 120 python bench.py --scene s2 --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --no-relaxed --no-config-c $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-50s value %8.1f  ms/step %8.3f  kernel ms %8.3f' % ('$s', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch']))" || exit 1
done
