#!/bin/bash
# build_variant2.sh NAME [SRC [extra hipcc flags]] [-- ORDER...]: like build_variant.sh, but the link keeps the Makefile's object order
# (sorted) with SRC's object replaced IN PLACE, and SRC is compiled with the Makefile's per-object flags.  With "-- a.o b.o ..." the named
# objects are linked FIRST in that order (link-order experiments: no effect, every translation unit is its own code object), the rest
# sorted behind.  ALWAYS build a control first: the unchanged source through this script must give the tree's numbers.
set -e
NAME=$1; shift
CS=/root/repo/instancediff_amd/csrc
mkdir -p /root/repo/instancediff_amd/variants /tmp/variants/$NAME
SRC=""; FLAGS=(); FIRST=()
if [ $# -gt 0 ] && [ "$1" != "--" ]; then SRC=$1; shift; fi
while [ $# -gt 0 ] && [ "$1" != "--" ]; do FLAGS+=("$1"); shift; done
if [ $# -gt 0 ]; then shift; FIRST=("$@"); fi
REPL=""
if [ -n "$SRC" ]; then
  REPL=$(basename "$SRC" .hip).o
  cp "$SRC" $CS/_variant_$NAME.hip
  trap 'rm -f $CS/_variant_$NAME.hip' EXIT
  # the Makefile's per-object flags (r05: the first version of this script and build_variant.sh compiled conv_wino4.hip WITHOUT its
  # -fno-slp-vectorize: every variant of that kernel came out 0.3-0.5 ms/step slower than the tree for that reason alone)
  EXTRA=""
  case "$REPL" in conv_wino4.o|conv_wino4h.o|conv_wino4_wgrad.o|conv_select.o) EXTRA="-fno-slp-vectorize";; sde.o) EXTRA="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $EXTRA "${FLAGS[@]}" -c $CS/_variant_$NAME.hip -o /tmp/variants/$NAME/$REPL
  rm -f $CS/_variant_$NAME.hip
fi
OBJS=()
for f in "${FIRST[@]}"; do if [ "$f" = "$REPL" ]; then OBJS+=(/tmp/variants/$NAME/$REPL); else OBJS+=($CS/$f); fi; done
for o in $(cd $CS && ls *.o | sort); do
  skip=0; for f in "${FIRST[@]}"; do [ "$f" = "$o" ] && skip=1; done; [ $skip = 1 ] && continue
  if [ "$o" = "$REPL" ]; then OBJS+=(/tmp/variants/$NAME/$REPL); else OBJS+=($CS/$o); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/instancediff_amd/variants/libidiff_$NAME.so "${OBJS[@]}"
echo built $NAME: ${OBJS[@]##*/}
