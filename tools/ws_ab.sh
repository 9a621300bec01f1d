#!/bin/bash
# A/B on one box: the wave-specialised chain (default) against chain_dma_kernel (chain_no_ws=1), bitwise check first.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/ws_check.py > gpurun_out/ws_check.log 2>&1; echo "ws_check rc=$?" >> gpurun_out/ws_check.log
tail -8 gpurun_out/ws_check.log
for rep in 1 2; do
  echo "ws      : $(timeout -k 10 200 python tools/chain_time.py) || $(timeout -k 10 300 python tools/saga_time.py | tail -1)"
  echo "dma     : $(CIAO_OPTS=chain_no_ws=1 timeout -k 10 200 python tools/chain_time.py) || $(CIAO_OPTS=chain_no_ws=1 timeout -k 10 300 python tools/saga_time.py | tail -1)"
done 2>&1 | tee gpurun_out/ws_ab.txt
