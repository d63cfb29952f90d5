#!/bin/bash
# beta-recsys_amd/libhiprec_debug.so: the library with its timing switches compiled in (HIPREC_OWNED_DBG of
# csrc/mf_owned.hip, HIPREC_SLICED_EXP of csrc/spmm_sliced.hip, the fused NCF launch's in-kernel timestamps).  Use with HIPREC_LIB=libhiprec_debug.so; the product
# library (python -c "import __graft_entry__ as g; g.build()") has none of them.
set -e
cd "$(dirname "$0")/../beta-recsys_amd/csrc"
out=../libhiprec_debug.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wall -Wno-unused-function \
  -DHIPREC_OWNED_DEBUG -DHIPREC_SLICED_DEBUG -DHIPREC_NCF_DEBUG -shared *.hip -o "$out"
echo "built $out"
