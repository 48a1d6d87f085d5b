#!/bin/bash
# the weight-stationary kernels' epilogue on packed fp32 pairs (libconzic_hip_pk.so, built from the commit before) against scalar fp32 (product library): per library
# the weight-stationary kernel and the untouched 128 x 128 kernel alternate in ONE process on the same shape; compare the ratio wreg / tiled between the libraries
cd /root/repo; O=gpurun_out/r05k; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "folded or wreg or x16" 2>&1 | tail -2 > $O/tests.txt
for rep in 1 2; do for lib in "" _pk; do
  [ -f conzic_amd/lib/libconzic_hip$lib.so ] || continue
  for spec in "156000 2048 512 1 0 6:0:0,6:0:7,0:0:0" "156000 1536 512 0 0 6:0:0,6:0:7,0:0:0"; do
    CZC_LIB_PATH=$PWD/conzic_amd/lib/libconzic_hip$lib.so python tools/ab_gemm.py $spec 8 | awk -v L="lib$lib" '{print L, $2, $6, "median", $9}' | paste - - - | awk '{printf "%s %s plain %s folded %s tiled %s  plain/tiled %.4f folded/tiled %.4f\n", $1, $2, $5, $10, $15, $5/$15, $10/$15}'
  done
done; done > $O/ab.txt 2>&1
cat $O/tests.txt $O/ab.txt
