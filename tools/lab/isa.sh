#!/bin/bash
# tools/lab/isa.sh <out.s> [part=400] [extra -D flags]: gfx950 ISA + resource usage of one part of hgt_edge_agg_mfma.hip (dev layouts)
OUT=$1; PART=${2:-400}; shift; shift
cd "$(dirname "$0")/../../pyhgt_amd/csrc"
V=$(echo $PART | cut -c1); R=$(echo $PART | cut -c2); F=$(echo $PART | cut -c3)
/opt/rocm/bin/hipcc -DHGT_DEV_LAYOUTS -DHGT_LAB_KERNELS "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -I../../include \
    -DHGT_MFMA_PART_VEC=$V -DHGT_MFMA_PART_RTE=$R -DHGT_MFMA_PART_F16=$F -S --cuda-device-only hgt_edge_agg_mfma.hip -o $OUT 2>&1 | grep -E "error|warning: (?!unused)" | head -30
python "$(dirname "$0")/../../tools/lab/isa_summary.py" $OUT 2>/dev/null || python /root/repo/tools/lab/isa_summary.py $OUT
