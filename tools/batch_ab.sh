#!/bin/bash
# chain batches (ON the GPU box): the chain parity tests and the batch tests, the single-chain timings (must not have moved), then
# the aggregate curve of K chains in one launch against K streams.
# `batch_ab.sh prof`: instead, rocprofv3 of ONE launch of 256 chains -- kernel stats, then the FETCH_SIZE / WRITE_SIZE passes
# (separate runs, --kernel-trace only) -> gpurun_out/batch_prof/.
set -o pipefail
mkdir -p gpurun_out
if [ "$1" = prof ]; then
  R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out/batch_prof"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
  cd /tmp
  CIAO_KS=256 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o s -- python3 "$R/tools/lambda_path.py" svrg batch > "$O/stats.log" 2>&1; echo "stats rc=$?"
  for c in FETCH_SIZE WRITE_SIZE; do
  CIAO_KS=256 timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_$c" -o p -- python3 "$R/tools/lambda_path.py" svrg batch > "$O/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
  done
  grep "chain_dma" "$O/stats/s_kernel_stats.csv" | cut -c1-200
  grep "chain_dma" "$O"/pmc_FETCH_SIZE/p_counter_collection.csv | awk -F, '{print $7, $(NF-3), $(NF-2)}' | tail -4
  grep "chain_dma" "$O"/pmc_WRITE_SIZE/p_counter_collection.csv | awk -F, '{print $7, $(NF-3), $(NF-2)}' | tail -4
  exit 0
fi
timeout -k 10 900 python -m pytest tests/test_gpu_chain_batch.py tests/test_gpu_parity.py -q -m gpu -x -k "batch or svrg or saga or finito or chain or wave_spec" > gpurun_out/batch_tests.log 2>&1
rc=$?
tail -5 gpurun_out/batch_tests.log
[ $rc -eq 0 ] || exit $rc
{
echo "single chain: $(python tools/chain_time.py) || $(python tools/saga_time.py | tail -1)"
echo "single chain, saga on chain_dma_kernel: $(CIAO_OPTS=chain_no_ws=1 python tools/saga_time.py | tail -1)"
echo "== SVRG, K chains in one launch"; python tools/lambda_path.py svrg batch 2>&1 >/dev/null | grep "K="
echo "== SAGA (N=200k, a table per chain), K chains in one launch"; CIAO_N=200000 python tools/lambda_path.py saga batch 2>&1 >/dev/null | grep "K="
echo "== SVRG, K streams (default hardware queues)"; CIAO_KS=1,4,16,64,256 python tools/lambda_path.py svrg 2>&1 >/dev/null | grep "K="
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/batch_ab.txt
