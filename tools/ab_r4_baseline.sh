#!/bin/bash
# Build the SpMM kernels of an earlier commit (git rev $1, default f33e508 = end of round 4) as library variant $2 (default "r4") for tools/ab_fused.py: the A/B partner
# of the long-row split / heavy-tiles-first kernels of round 5 (balanced graphs must not pay for them).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
rev=${1:-f33e508}
name=${2:-r4}                # variant name: build/variants/$name (ab_fused.py takes it as its argument)
src=$root/build/${name}src/tm-gcn_amd/csrc
mkdir -p $src $root/build/variants/$name
srcs="spmm.hip spmm_gemm.hip pointwise.hip"
git -C $root cat-file -e $rev:tm-gcn_amd/csrc/pools.hip 2>/dev/null && srcs="$srcs pools.hip"      # (round 5: the pools have their own file)
for f in Makefile common.h spmm_row.h async_stage.h $srcs; do
  git -C $root show $rev:tm-gcn_amd/csrc/$f > $src/$f
done
mkdir -p $root/build/${name}src/include $root/build/${name}src/tools
git -C $root show $rev:include/tmgcn.h > $root/build/${name}src/include/tmgcn.h
cp $root/tools/check_reserved_vgprs.py $root/build/${name}src/tools/
d=$root/build/variants/$name
make -s -C $src -j4 OBJDIR=$d LIB=$d/libtmgcn_hip.so SRCS="$srcs" $d/libtmgcn_hip.so 2>&1 | grep -E "error|VIOLATION|Error" || true
test -f $d/libtmgcn_hip.so && echo "built $name ($rev)"
