# FETCH_SIZE / WRITE_SIZE passes over bench.py (separate runs: the TCC has 4 counter slots) -> gpurun_out/pmc/pmc_{fetch,write}/
set -x
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 10 --warmup 2 --cpu-rows 0 --no-extras"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/pmc_fetch -o f -- $B > gpurun_out/pmc/b1.json 2> gpurun_out/pmc/b1.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/pmc_write -o w -- $B > gpurun_out/pmc/b2.json 2> gpurun_out/pmc/b2.err
find gpurun_out/pmc -name "*counter_collection.csv" | head
