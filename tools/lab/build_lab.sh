#!/bin/bash
# Lab builds of the aggregation kernel: the dev-layout library (d = 256 / 8 heads, d = 64 / 4 heads) with ONE part of
# hgt_edge_agg_mfma.hip (VEC = 4, no RTE, bf16 split = the c2 kernel) recompiled under extra -D switches.
#   tools/lab/build_lab.sh <name> "<-D flags>" [part=400]      ->  pyhgt_amd/lib_lab_<name>/libhgt_hip.so   (load with HGT_LIB_PATH)
set -e
NAME=$1; FLAGS=$2; PART=${3:-400}
cd "$(dirname "$0")/../../pyhgt_amd/csrc"
make BUILD=build_lab LIBDIR=../lib_lab LAB=1 EXTRA=-DHGT_DEV_LAYOUTS -j8 > /dev/null
V=$(echo $PART | cut -c1); R=$(echo $PART | cut -c2); F=$(echo $PART | cut -c3)
OBJ=build_lab/lab_${NAME}_p$PART.o
/opt/rocm/bin/hipcc -DHGT_DEV_LAYOUTS -DHGT_LAB_KERNELS $FLAGS --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function \
    -I../../include -DHGT_MFMA_PART_VEC=$V -DHGT_MFMA_PART_RTE=$R -DHGT_MFMA_PART_F16=$F -c hgt_edge_agg_mfma.hip -o $OBJ \
    -Rpass-analysis=kernel-resource-usage 2> build_lab/lab_${NAME}_p$PART.rpass || { tail -30 build_lab/lab_${NAME}_p$PART.rpass; exit 1; }
mkdir -p ../lib_lab_$NAME
OBJS=$(ls build_lab/hgt_*.o build_lab/lab_hgt_*.o | grep -v "hgt_edge_agg_mfma_p$PART.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_lab_$NAME/libhgt_hip.so $OBJS $OBJ
grep -A12 "Function Name: .*k_edge_aggregate_update_mfma.*Li4ELi8E" build_lab/lab_${NAME}_p$PART.rpass | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" | head -12
