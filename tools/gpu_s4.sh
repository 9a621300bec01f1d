#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/s4"
mkdir -p "$O"
cd "$R"
for v in product st_t1p1 st_t2p1 st_t0p2 st_t2p2; do
  if [ $v = product ]; then unset CIAO_HIP_LIB; else export CIAO_HIP_LIB="$R/build/$v/libciao_hip.so"; fi
  echo "== $v batches"; timeout -k 10 300 python tools/finito_batch_time.py 64 256 1024 4096 16384 2>&1 | grep "r=" | sort -u | tee "$O/batch_$v.log"
  echo "== $v table modes"; TABLE_ONLY=saga_init_f32_d1024,finito_init_f32_d4096,finito_batch_r65536_f32_d4096,saga_init_f64_d1024,finito_batch_r65536_f64_d1024 timeout -k 10 300 python tools/table_modes.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); [print(k, round(v['alg_GBps'])) for k,v in d.items()]" | tee "$O/table_$v.log"
done
