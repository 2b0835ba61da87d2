#!/bin/bash
# A/B of experiment builds of the queue kernels on config D (run on the GPU box): tools/wfq_ab.sh "<extra bench args>" lib1.so lib2.so ...
# per build: trace / shade kernel totals per render from rocprofv3 --kernel-trace (one pipeline), and the bench's kernel ms with the default pipelines
set -o pipefail
export TMPDIR=/tmp
EXTRA=$1; shift
for LIB in "$@"; do
  export IRIS_HIP_LIB=$PWD/$LIB
  tag=$(basename $LIB .so)
  rm -rf gpurun_out/ab_$tag
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag -o t -- python3 bench.py --scene s2 --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --no-relaxed --no-config-c --opt mode=3 --opt wfq_pipes=1 $EXTRA > gpurun_out/ab_$tag.log 2>&1 || { echo "$tag: profile run failed"; tail -3 gpurun_out/ab_$tag.log; exit 1; }
  two=$(timeout -k 10 120 python3 bench.py --scene s2 --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --no-relaxed --no-config-c --opt mode=3 $EXTRA 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_ms_per_launch'])") || exit 1
  python3 - $tag $two <<'PY'
import csv, glob, sys
tag, two = sys.argv[1], sys.argv[2]
tot = {}
for f in glob.glob("gpurun_out/ab_%s/**/*kernel_stats.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("k_wfq_trace", "k_wfq_shade", "k_resolve"):
            if k in r["Name"]: tot[k] = (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]))
n = max(1, tot.get("k_resolve", (0, 2))[1])
print("%-10s trace %7.2f ms  shade %7.2f ms per render (1 pipeline, %d passes) | default pipelines: %s ms" % (
    tag, tot["k_wfq_trace"][0] / n, tot["k_wfq_shade"][0] / n, tot["k_wfq_shade"][1] / n, two), flush=True)
PY
done
