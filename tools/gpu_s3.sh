#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/s3"
mkdir -p "$O"
cd "$R"
echo "== batches eager, own stream"; CIAO_OWN_STREAM=1 timeout -k 10 300 python tools/finito_batch_time.py 16 64 256 1024 4096 2>&1 | grep "r=" | tee "$O/batch_eager.log"
echo "== batches graph, own stream"; CIAO_OWN_STREAM=1 CIAO_OPTS=graph_batches=1 timeout -k 10 300 python tools/finito_batch_time.py 16 64 256 1024 4096 2>&1 | grep "r=\|rror" | tee "$O/batch_graph.log"
for bpc in 2 3 4; do echo "== split_blocks_per_cu=$bpc"; CIAO_OPTS=split_blocks_per_cu=$bpc timeout -k 10 300 python tools/finito_batch_time.py 1024 4096 16384 2>&1 | grep "r=" | tee "$O/batch_bpc$bpc.log"; done
echo "== wave-per-row instead"; CIAO_OPTS=split_max_rows=0 timeout -k 10 300 python tools/finito_batch_time.py 1024 4096 16384 2>&1 | grep "r=" | tee "$O/batch_fast.log"
for bpc in 2 4 8; do echo "== wave-per-row sweep_blocks_per_cu=$bpc"; CIAO_OPTS=split_max_rows=0,sweep_blocks_per_cu=$bpc timeout -k 10 300 python tools/finito_batch_time.py 4096 16384 2>&1 | grep "r=" | tee "$O/batch_fast$bpc.log"; done
