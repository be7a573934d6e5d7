#!/bin/bash
# tools/lab/tile_gemm.sh "<rows k n_out>" ... : cold / warm durations of the tile and slab typed linears (tools/lab/tile_gemm.py)
cd ${GRAFT_REPO_ROOT:-$(pwd)}; ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
for shape in "$@"; do
    rm -rf /tmp/tg; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python $ROOT/tools/lab/tile_gemm.py run $shape > /tmp/tg.log 2>&1 || tail -3 /tmp/tg.log
    echo "== $shape"; python $ROOT/tools/lab/tile_gemm.py show /tmp/tg
done
