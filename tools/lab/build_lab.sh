#!/bin/bash
# Experiment build (never shipped): pyhgt_amd/lib_lab/libhgt_hip.so = the product objects + the wide persistent Q|K|V kernel of
# tools/lab/hgt_gemm_wide.hip in N variants (one object per line of the variant list: "<id> <-D flags>"), selected at run time by
# HGT_WD_VARIANT=<id> (unset = the product kernel).  Usage: tools/lab/build_lab.sh variants.txt [extra flags for hgt_gemm_bf16x3]
#   HGT_LIB_PATH=$PWD/pyhgt_amd/lib_lab/libhgt_hip.so HGT_WD_VARIANT=14 python tools/bench_linear.py --which bf16x3
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/pyhgt_amd/csrc; B=$C/build_lab; L=$ROOT/pyhgt_amd/lib_lab
LIST=${1:-$ROOT/tools/lab/variants.txt}; shift || true
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -I$ROOT/include -I$C -DHGT_LAB_WIDE"
mkdir -p $B $L; rm -f $B/*.o
make -C $C -j8 > /dev/null
ids=""
while read v flags; do
  [ -z "$v" ] && continue
  ids="$ids $v"
  /opt/rocm/bin/hipcc $FL -DWD_SUFFIX=_v$v $flags -c $ROOT/tools/lab/hgt_gemm_wide.hip -o $B/wide_v$v.o &
done < $LIST
/opt/rocm/bin/hipcc $FL "$@" -c $C/hgt_gemm_bf16x3.hip -o $B/hgt_gemm_bf16x3.o &
wait
DECL=""; CALL=""; for v in $ids; do DECL="$DECL DECL($v)"; CALL="$CALL CALL($v)"; done
/opt/rocm/bin/hipcc $FL "-DWD_DECLS=$DECL" "-DWD_CALLS=$CALL" -c $ROOT/tools/lab/hgt_gemm_wide_lab.hip -o $B/lab_dispatch.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libhgt_hip.so $(ls $C/build/*.o | grep -v hgt_gemm_bf16x3.o) $B/*.o
ls -la $L/libhgt_hip.so
