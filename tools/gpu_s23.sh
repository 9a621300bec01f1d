#!/bin/bash
# single-wave chain (option chain_one_wave) against the four-wave chain: correctness on the chain tests, then timings
set -o pipefail
mkdir -p gpurun_out
CIAO_TEST_OPTS=chain_one_wave=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "svrg_epochs or saga_steps or finito_steps or lfinito" > gpurun_out/s23_tests.log 2>&1
rc=$?
tail -4 gpurun_out/s23_tests.log
[ $rc -eq 0 ] || exit $rc
for o in "" "chain_one_wave=1"; do
  echo "== opts: ${o:-none}"
  CIAO_OPTS=$o python tools/chain_time.py 2>/dev/null
  CIAO_D=512 CIAO_OPTS=$o python tools/chain_time.py 2>/dev/null
  CIAO_OPTS=$o python tools/saga_time.py 2>/dev/null | tail -1
done 2>&1 | tee gpurun_out/s23_ab.txt
