#!/bin/bash
# tools/lab/latency_ab.sh <precision> lib...: the latency-regime timelines (tools/trace_latency.py: c3, c5, mag4) on several lab libraries
cd ${GRAFT_REPO_ROOT:-$(pwd)}; ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
prec=$1; shift
for n in "$@"; do
    case $n in prod) L=pyhgt_amd/lib/libhgt_hip.so;; *) L=pyhgt_amd/lib_lab_$n/libhgt_hip.so;; esac
    for wl in c3:1 c5:2 mag4:4; do
        w=${wl%%:*}; nl=${wl##*:}
        rm -rf /tmp/la; HGT_LIB_PATH=$ROOT/$L timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/la -o t -- python $ROOT/tools/trace_latency.py run $w $prec > /tmp/la.log 2>&1 || tail -3 /tmp/la.log
        echo "== $n $w $prec"; python $ROOT/tools/trace_latency.py show /tmp/la $nl | cut -c1-90 | tail -12
    done
done
