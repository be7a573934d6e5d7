#!/bin/bash
# VGPR / SGPR / spill / LDS usage of the kernels in one object file: tools/kernel_resources.sh build/hgt_edge_agg_mfma.o [name filter]
set -e
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | python3 -c "
import sys,re
cur={}
flt=sys.argv[1] if len(sys.argv)>1 else ''
def emit(c):
    if c.get('name') and flt in c['name']:
        print('%-90s vgpr %3s sgpr %3s spill v%s s%s lds %6s scratch %s' % (c['name'][:90], c.get('vgpr_count'), c.get('sgpr_count'), c.get('vgpr_spill_count'), c.get('sgpr_spill_count'), c.get('group_segment_fixed_size'), c.get('private_segment_fixed_size')))
for line in sys.stdin:
    m=re.match(r'\s*-?\s*\.(\w+):\s+(\S+)', line)
    if not m: continue
    k,v=m.groups()
    if k=='agpr_count' and cur: emit(cur); cur={}
    if k=='name' and 'name' in cur and '.' not in v: pass
    if k in ('name','vgpr_count','sgpr_count','vgpr_spill_count','sgpr_spill_count','group_segment_fixed_size','private_segment_fixed_size'):
        if k=='name' and not v.startswith('_Z'): continue
        cur[k]=v
emit(cur)
" "${2:-}"
rm -rf $T
