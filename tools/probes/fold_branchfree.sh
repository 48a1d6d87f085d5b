#!/bin/bash
# the folded-LayerNorm consumer against the plain weight-stationary kernel, per compile-time ablation (`make ABL=n
# LIB=../lib/libconzic_hip_abl<n>.so`): every line pair is plain vs folded alternated inside ONE process, so the ratio is the
# robust figure (separate processes move by 2-3 % on one box).  Build the ablation libraries first:
#   for a in 1 2 3 4 7 8; do make -C conzic_amd/csrc ABL=$a LIB=../lib/libconzic_hip_abl$a.so; done   (and delete them afterwards: they travel with every gpurun)
cd /root/repo; mkdir -p gpurun_out/r05f
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k folded 2>&1 | tail -3 > gpurun_out/r05f/tests.txt
for rep in 1 2; do
for lib in "" _abl8 _abl1 _abl2 _abl3 _abl4 _abl7; do
  [ -f conzic_amd/lib/libconzic_hip$lib.so ] || continue
  for m in 100000 156000; do
    CZC_LIB_PATH=$PWD/conzic_amd/lib/libconzic_hip$lib.so python tools/ab_gemm.py $m 2048 512 1 0 6:0:0,6:0:7 10 | awk -v L="lib$lib" '{print L, $1, $7, "median", $9}' | paste - - | awk '{printf "%s %s plain %s folded %s ratio %.4f\n", $1, $2, $5, $10, $10/$5}'
  done
done
done > gpurun_out/r05f/ab_fc1.txt 2>&1
python tools/ab_gemm.py 156000 1536 512 0 0 6:0:0,6:0:7 10 > gpurun_out/r05f/ab_qkv.txt 2>&1
cat gpurun_out/r05f/tests.txt gpurun_out/r05f/ab_fc1.txt gpurun_out/r05f/ab_qkv.txt
