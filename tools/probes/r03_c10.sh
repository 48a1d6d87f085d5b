set -x
cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/probes/pmc_bench_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
tail -60 gpurun_out/pmc_traffic.log
