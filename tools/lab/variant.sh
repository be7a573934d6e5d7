#!/bin/bash
# tools/lab/variant.sh <name> "<-D flags>" <source.hip>...: the product library with the named sources recompiled under extra switches
#   -> pyhgt_amd/lib_lab_<name>/libhgt_hip.so (load with HGT_LIB_PATH; tools/lab/latency_ab.sh <precision> prod <name>...)
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../../pyhgt_amd/csrc"
make -j8 > /dev/null
SKIP=""; NEW=""
for s in "$@"; do
    b=${s%.hip}; o=build/lab_${NAME}_$b.o
    /opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -I../../include -c $s -o $o \
        -Rpass-analysis=kernel-resource-usage 2> build/lab_${NAME}_$b.rpass || { grep -E "error" -A5 build/lab_${NAME}_$b.rpass | head -40; exit 1; }
    SKIP="$SKIP build/$b.o"; NEW="$NEW $o"
done
OBJS=""
for o in build/hgt_*.o; do case " $SKIP " in *" $o "*) ;; *) OBJS="$OBJS $o";; esac; done
mkdir -p ../lib_lab_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_lab_$NAME/libhgt_hip.so $OBJS $NEW
echo built lib_lab_$NAME
