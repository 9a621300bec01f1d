#!/bin/bash
# crossover between a batch run as a chain and a batch-parallel launch, after the chain work
mkdir -p gpurun_out
for cfg in "1024 " "2048 " "4096 " "8192 " "128 1" "256 1"; do
  set -- $cfg
  for o in chain_max_batch=64 chain_max_batch=0; do
    echo "d=$1 f64=${2:-0} $o: $(CIAO_D=$1 CIAO_F64=$2 CIAO_OPTS=$o python tools/finito_batch_time.py 8 12 16 24 32 48 64 2>/dev/null | sed 's/, [0-9]* GB.*//' | tr '\n' ' ')"
  done
done | tee gpurun_out/s34_crossover.txt
