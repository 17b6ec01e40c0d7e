#!/bin/bash
# build_variant.sh NAME SRC [extra hipcc flags]: links an alternative libidiff (the object of SRC's basename replaced by a build
# of SRC with the extra flags) to instancediff_amd/variants/libidiff_NAME.so, selectable with IDIFF_LIB=... (A/B kernel experiments)
set -e
NAME=$1; SRC=$2; shift 2
CS=/root/repo/instancediff_amd/csrc
OBJ=$(basename "$SRC" .hip).o
mkdir -p /root/repo/instancediff_amd/variants /tmp/variants/$NAME
# the copy is compiled from the sources' directory (relative includes) and removed whatever the compiler says: a stray *.hip there
# would be swept into the library by the Makefile's wildcard
cp "$SRC" $CS/_variant_$NAME.hip
trap 'rm -f $CS/_variant_$NAME.hip' EXIT
EXTRA=""  # the Makefile's per-object flags (missing until r05: variants of conv_wino4.hip were built with SLP packing on)
case "$OBJ" in conv_wino4.o|conv_wino4h.o|conv_wino4_wgrad.o|conv_select.o) EXTRA="-fno-slp-vectorize";; sde.o) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function $EXTRA "$@" -c $CS/_variant_$NAME.hip -o /tmp/variants/$NAME/$OBJ
rm -f $CS/_variant_$NAME.hip
OBJS=$(ls $CS/*.o | grep -v "/$OBJ")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/instancediff_amd/variants/libidiff_$NAME.so $OBJS /tmp/variants/$NAME/$OBJ
echo built $NAME
