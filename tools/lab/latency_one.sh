#!/bin/bash
# tools/lab/latency_one.sh <workload:layers> <precision> <grep pattern> lib...: one workload's timeline on several lab libraries, the kernels matching the pattern
cd ${GRAFT_REPO_ROOT:-$(pwd)}; ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
wl=$1; prec=$2; pat=$3; shift 3
w=${wl%%:*}; nl=${wl##*:}
for n in "$@"; do
    case $n in prod) L=pyhgt_amd/lib/libhgt_hip.so;; *) L=pyhgt_amd/lib_lab_$n/libhgt_hip.so;; esac
    rm -rf /tmp/la; HGT_LIB_PATH=$ROOT/$L timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/la -o t -- python $ROOT/tools/trace_latency.py run $w $prec > /tmp/la.log 2>&1 || tail -3 /tmp/la.log
    echo "== $n $w $prec"; python $ROOT/tools/trace_latency.py show /tmp/la $nl | cut -c1-100 | grep -E "$pat|sum of" | tail -4
done
