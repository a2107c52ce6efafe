#!/bin/bash
# usage: resusage.sh file.hip  -> kernel, VGPRs, scratch bytes/lane, occupancy, LDS
here=$(cd "$(dirname "$0")" && pwd)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$here/../../include -I$here -c "$1" -o /dev/null \
  -mllvm -pragma-unroll-threshold=200000 -mllvm -amdgpu-mfma-vgpr-form=1 -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" |
  sed -E 's/.*(Function Name: [^ ]*|VGPRs: [0-9]*|ScratchSize \[bytes\/lane\]: [0-9]*|Occupancy \[waves\/SIMD\]: [0-9]*|LDS Size \[bytes\/block\]: [0-9]*).*/\1/' |
  paste - - - - - | awk '{print $3, "vgpr="$5, "scratch="$8, "occ="$11, "lds="$15}'
