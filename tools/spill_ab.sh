#!/bin/bash
# Same-box A/B of the round-5 register work (arguments read through the kernel-argument segment, pinned hot fields, LDS-DMA
# destinations as base + immediate): the product library against the round-4 library (build/r04base/libciao_hip.so, built from the
# round-4 commit).  Every figure carries a digest of the resulting state: the two libraries must agree BITWISE.
# sharded SVRG fp64, fp64 SAGA d = 1024, adaptive Finito d = 1024 -- plus the unsharded chains, the one-wave
# shapes and the 16 KiB-row chains that park registers in AGPRs.
set -o pipefail
mkdir -p gpurun_out
B="${CIAO_AB_LIB:-build/r04base/libciao_hip.so}"
run() {  # label, env..., script
  local label="$1"; shift
  echo "new  $label: $(env "$@" 2>&1 | grep -v amdgpu.ids | tr '\n' ';')"
  echo "r04  $label: $(env CIAO_HIP_LIB=$B "$@" 2>&1 | grep -v amdgpu.ids | tr '\n' ';')"
}
{
for rep in 1 2; do
  run "svrg d=1024"            python tools/chain_time.py
  run "svrg d=1024 sharded"    CIAO_SHARDED=1 python tools/chain_time.py
  run "svrg d=256 (one wave)"  CIAO_D=256 python tools/chain_time.py
  run "svrg d=2048 (J=4)"      CIAO_D=2048 python tools/chain_time.py
  run "afinito"                python tools/af_time.py
done
run "saga routes f64+f32 d=1024" CIAO_M=200000 python tools/saga_ab.py
run "saga routes d=2048"         CIAO_D=2048 CIAO_N=400000 CIAO_M=200000 python tools/saga_ab.py
run "finito r=1 chain d=4096"    python tools/finito_batch_time.py 1 2
} 2>&1 | tee gpurun_out/spill_ab.txt
