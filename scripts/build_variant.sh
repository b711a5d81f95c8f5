#!/bin/bash
# Build a variant of libdewi_hip.so with extra compile flags for ONE source file (A/B experiments).
# usage: bash scripts/build_variant.sh <name> <source.hip> <flags...>   -> <pkg>/lib_<name>/libdewi_hip.so
set -e
name=$1; src=$2; shift 2
P=dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd
make -s -C $P/csrc -j8
mkdir -p $P/lib_$name $P/build/var_$name
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function --offload-arch=gfx950 "$@" \
  -c $P/csrc/$src -o $P/build/var_$name/$base.o
objs=""
for o in $P/build/*.o; do
  if [ "$(basename $o)" = "$base.o" ]; then objs="$objs $P/build/var_$name/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o $P/lib_$name/libdewi_hip.so
echo built $P/lib_$name/libdewi_hip.so
