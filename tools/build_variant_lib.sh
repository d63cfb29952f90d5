#!/bin/bash
# beta-recsys_amd/libhiprec_<name>.so: the product sources with extra -D flags, for A/B timing on the GPU box
# (HIPREC_LIB=libhiprec_<name>.so selects it).  Reuses the product build's objects for the files the flags do not
# touch: bash tools/build_variant_lib.sh <name> "<flags>" file.hip [file.hip ...]
set -e
name=$1; flags=$2; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
objdir=$root/build/variant_$name
mkdir -p $objdir
objs=""
for f in $root/beta-recsys_amd/csrc/*.hip; do
  b=$(basename $f)
  if [[ " $* " == *" $b "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wall -Wno-unused-function $flags -c $f -o $objdir/$b.o
    objs="$objs $objdir/$b.o"
  else
    objs="$objs $root/build/libhiprec/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $root/beta-recsys_amd/libhiprec_$name.so
echo "built libhiprec_$name.so"
