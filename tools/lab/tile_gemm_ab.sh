#!/bin/bash
# tools/lab/tile_gemm_ab.sh "<rows k n_out>" lib...: tools/lab/tile_gemm.py on several lab libraries (pyhgt_amd/lib_lab_<name>)
cd ${GRAFT_REPO_ROOT:-$(pwd)}; ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
shape=$1; shift
for n in "$@"; do
    case $n in prod) L=pyhgt_amd/lib/libhgt_hip.so;; *) L=pyhgt_amd/lib_lab_$n/libhgt_hip.so;; esac
    rm -rf /tmp/tg; HGT_LIB_PATH=$ROOT/$L timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python $ROOT/tools/lab/tile_gemm.py run $shape > /tmp/tg.log 2>&1 || tail -3 /tmp/tg.log
    echo "== $n $shape"; python $ROOT/tools/lab/tile_gemm.py show /tmp/tg | sed -n 2,4p
done
