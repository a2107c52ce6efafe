#!/bin/bash
# Build tuning variants of the kernel library into build/variants/<name>/libtmgcn_hip.so
# usage: tools/ab_variants.sh name "-DTMGCN_FUSED_OCC=4 -DTMGCN_FUSED_U=8" [name2 "flags2" ...]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  d=$root/build/variants/$name
  mkdir -p $d
  make -s -C $root/tm-gcn_amd/csrc -j4 OBJDIR=$d LIB=$d/libtmgcn_hip.so EXTRA="$flags" SRCS="spmm.hip spmm_gemm.hip gemm.hip pointwise.hip edge_head.hip loss.hip pools.hip" $d/libtmgcn_hip.so 2>&1 | grep -E "error|VIOLATION|Error" || true
  test -f $d/libtmgcn_hip.so || { echo "FAILED to build $name"; exit 1; }
  echo "built $name: $flags"
done
