set -x
mkdir -p gpurun_out/r03h
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -q -m gpu -k "bitwise or every_row_count or depend_on_the_batch or fp16" 2>&1 | tail -30 > gpurun_out/r03h/pytest.log
tail -12 gpurun_out/r03h/pytest.log
bash tools/probes/r03_profiles.sh > gpurun_out/r03h/profiles.log 2>&1
tail -60 gpurun_out/r03h/profiles.log | cut -c1-250
