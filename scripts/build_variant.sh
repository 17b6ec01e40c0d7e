#!/bin/bash
# build_variant.sh NAME SRC [extra hipcc flags]: links an alternative libidiff (conv_wino.o replaced by a build of SRC) to
# instancediff_amd/variants/libidiff_NAME.so, selectable with IDIFF_LIB=... (A/B kernel experiments)
set -e
NAME=$1; SRC=$2; shift 2
CS=/root/repo/instancediff_amd/csrc
mkdir -p /root/repo/instancediff_amd/variants /tmp/variants/$NAME
cp "$SRC" $CS/_variant_$NAME.hip
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function "$@" -c $CS/_variant_$NAME.hip -o /tmp/variants/$NAME/conv_wino.o
rm -f $CS/_variant_$NAME.hip
OBJS=$(ls $CS/*.o | grep -v conv_wino.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/instancediff_amd/variants/libidiff_$NAME.so $OBJS /tmp/variants/$NAME/conv_wino.o
echo built $NAME
