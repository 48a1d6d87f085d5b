set -x
mkdir -p gpurun_out/r03m
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r03m/pytest.log
tail -5 gpurun_out/r03m/pytest.log
timeout 600 python bench.py > gpurun_out/r03m/bench_default.json 2> gpurun_out/r03m/bench_default.err
timeout 600 python bench.py --opt test:w_dbg=8 --no-alt --no-cpu-baseline --no-invariance > gpurun_out/r03m/bench_oldepi.json 2>> gpurun_out/r03m/bench_default.err
timeout 600 python bench.py --no-alt --no-cpu-baseline --no-invariance > gpurun_out/r03m/bench_newepi.json 2>> gpurun_out/r03m/bench_default.err
cut -c1-260 gpurun_out/r03m/bench_default.json; cut -c1-200 gpurun_out/r03m/bench_oldepi.json; cut -c1-200 gpurun_out/r03m/bench_newepi.json
