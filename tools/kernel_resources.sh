#!/bin/bash
# Registers, scratch and LDS of every kernel in a built library, from its gfx950 code object: tools/kernel_resources.sh [lib.so] [name filter]
LIB=${1:-iris_amd/lib/libiris_hip.so}; FILTER=${2:-.}
T=$(mktemp -d); B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $LIB && $B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle && $B/llvm-readelf --notes $T/dev.co > $T/notes.txt
python3 - $T/notes.txt "$FILTER" <<'PY'
import re, subprocess, sys
t = open(sys.argv[1]).read()
for blk in t.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '').split('(')[0]
    if not re.search(sys.argv[2], name):
        continue
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    print('%-40s vgpr %3s sgpr %3s scratch %4s lds %6s' % (name[-40:], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
PY
rm -rf $T
