#!/bin/bash
# gemm256x with the branch-free steady K loop (product library) against the form before it (libconzic_hip_old.so built from the
# previous commit in a worktree): per library the ping-pong kernel and the untouched 128 x 128 kernel alternate in ONE process on
# the same shape; compare the ratio ping-pong / tiled between the libraries
cd /root/repo; O=gpurun_out/r05p; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -3 > $O/tests.txt
for rep in 1 2; do
for lib in "" _old; do
  [ -f conzic_amd/lib/libconzic_hip$lib.so ] || continue
  for spec in "156000 512 2048 0 6 7:0:6,0:0:6" "60000 512 2048 0 6 7:0:6,0:0:6" "156000 512 2048 0 1 7:0:1,0:0:1"; do
    CZC_LIB_PATH=$PWD/conzic_amd/lib/libconzic_hip$lib.so python tools/ab_gemm.py $spec 8 | awk -v L="lib$lib" '{print L, $1, $2, $3, $6, "median", $9}' | paste - - | awk '{printf "%s %s %s %s %s pp %s tiled %s ratio %.4f\n", $1, $2, $3, $4, $5, $7, $14, $7/$14}'
  done
done
done > $O/ab.txt 2>&1
cat $O/tests.txt $O/ab.txt
