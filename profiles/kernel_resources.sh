#!/bin/bash
# registers / LDS / scratch of the kernels of a built library, from the code object's metadata:
#   bash profiles/kernel_resources.sh [lib.so] [name filter]
LIB=${1:-$(dirname $0)/../tuturenderer_amd/libtutu_hip.so}
FILT=${2:-k_trace}
T=$(mktemp -d)
cp $LIB $T/lib.so
(cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
CO=$(ls $T/*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $CO | python3 -c "
import sys,re,subprocess
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk).group(1)
    dem=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    if '$FILT' not in dem: continue
    g=lambda k: re.search(r'\.'+k+r':\s+(\d+)',blk).group(1)
    print(f\"{dem[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}\")
"
rm -rf $T
