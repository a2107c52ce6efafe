#!/bin/bash
cp tm-gcn_amd/libtmgcn_hip.so /tmp/orig.so
for v in trace_dyn trace_rev; do
  cp build/variants/$v/libtmgcn_hip.so tm-gcn_amd/libtmgcn_hip.so
  for c in chess S1 S2z2; do
    timeout 200 python3 tools/l12_trace.py $c 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', '$c', 'span', d['kernel_span_us'], 'life_mean', d['block_lifetime_mean'], 'sum0', d['row_block_0']['tiles_summed'], 'gath0', d['row_block_0']['gathers_parked'][2], 'tail', d['tail_slab_and_tickets'][2])"
  done
done
cp /tmp/orig.so tm-gcn_amd/libtmgcn_hip.so
