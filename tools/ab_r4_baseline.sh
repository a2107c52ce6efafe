#!/bin/bash
# Build the ROUND-4 SpMM kernels (git rev $1, default f33e508) as library variant "r4" for tools/ab_fused.py: the A/B partner
# of the long-row split / heavy-tiles-first kernels of round 5 (balanced graphs must not pay for them).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
rev=${1:-f33e508}
src=$root/build/r4src/tm-gcn_amd/csrc
mkdir -p $src $root/build/variants/r4
for f in Makefile common.h spmm_row.h async_stage.h spmm.hip spmm_gemm.hip pointwise.hip; do
  git -C $root show $rev:tm-gcn_amd/csrc/$f > $src/$f
done
mkdir -p $root/build/r4src/include $root/build/r4src/tools
git -C $root show $rev:include/tmgcn.h > $root/build/r4src/include/tmgcn.h
cp $root/tools/check_reserved_vgprs.py $root/build/r4src/tools/
d=$root/build/variants/r4
make -s -C $src -j4 OBJDIR=$d LIB=$d/libtmgcn_hip.so SRCS="spmm.hip spmm_gemm.hip pointwise.hip" $d/libtmgcn_hip.so 2>&1 | grep -E "error|VIOLATION|Error" || true
test -f $d/libtmgcn_hip.so && echo "built r4 ($rev)"
