set -x
mkdir -p gpurun_out/r03i
cd /root/repo
export TMPDIR=/tmp
export CZC_LIB_PATH=/root/repo/conzic_amd/lib/libconzic_hip_exp.so
timeout 300 python tools/ab_gemm.py 312000 512 2048 0 1 7,7:4096,7:8192,7:12288,7:2048,7:3328,7:1792 8 > gpurun_out/r03i/ab_fc2_epi.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 512 512 0 1 7,7:4096,7:8192,7:12288 8 > gpurun_out/r03i/ab_out_epi.log 2>&1
unset CZC_LIB_PATH
cat gpurun_out/r03i/ab_*.log
cd /tmp && export CZC_NORMAL_EXIT=1 && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python bench.py --streams 1 --steps 1 --warmup 0 --no-profile --no-cpu-baseline --no-alt --no-invariance > $O/pmc_mfma.log 2>&1
ls -la $O/pmc_mfma
python tools/probes/pmc_mfma_summary.py $O/pmc_mfma > $O/pmc_mfma_summary.json 2> $O/pmc_mfma_summary.err
find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*kernel_trace.csv" -delete
cat $O/pmc_mfma_summary.json; tail -n 3 $O/pmc_mfma_summary.err; tail -n 2 $O/pmc_mfma.log | cut -c1-600
