set -x
mkdir -p gpurun_out/r03g
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -70 > gpurun_out/r03g/pytest.log
timeout 600 python bench.py > gpurun_out/r03g/bench_default.json 2> gpurun_out/r03g/bench_default.err
timeout 300 python bench.py --gamma 5 --len 12 --images 64 --steps 2 --no-alt --no-cpu-baseline > gpurun_out/r03g/bench_cfg4.json 2>> gpurun_out/r03g/bench_default.err
tail -60 gpurun_out/r03g/pytest.log; cut -c1-300 gpurun_out/r03g/bench_default.json; tail -3 gpurun_out/r03g/bench_default.err
