set -x
export TMPDIR=/tmp
OUT=gpurun_out/r3c; mkdir -p $OUT
AB_R=8000000 AB_POWER=1 python3 tools/ab_gemm.py dw_m16 > $OUT/ab_dw_m16_R8M.txt 2>&1; echo "ab_gemm 8M rc=$?" >> $OUT/status.log
AB_R=32000000 python3 tools/ab_gemm.py dw_m16 > $OUT/ab_dw_m16_R32M.txt 2>&1; echo "ab_gemm 32M rc=$?" >> $OUT/status.log
AB_T=4 python3 tools/ab_fused.py occ2_u4 occ2_u8 occ3_u4 qt > $OUT/ab_fused_variants.txt 2>&1; echo "ab_fused rc=$?" >> $OUT/status.log
cat $OUT/status.log; tail -20 $OUT/ab_dw_m16_R8M.txt; tail -12 $OUT/ab_dw_m16_R32M.txt; tail -24 $OUT/ab_fused_variants.txt
