#!/bin/bash
# Registers, spills and scratch of the kernels in a built object file, from the code object's metadata:
#   tools/kernel_resources.sh graphsage_amd/csrc/build/sage2.o [name filter]
set -e
obj=$1; filt=${2:-}
tmp=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$obj" $tmp/fat.bin
$L/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.o
$L/llvm-readelf --notes $tmp/dev.o | python3 -c "
import sys, re, subprocess
txt = sys.stdin.read()
for b in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', b) or [None, '?'])[1]
    name = g('name')
    try:
        name = subprocess.run(['$L/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    if '$filt' in name:
        print('%-110s vgpr %s sgpr %s vspill %s sspill %s scratch %s lds %s' % (name[:110], g('vgpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
"
rm -rf $tmp
