#!/bin/bash
# finalize_kernel column-width experiment: experiment builds with CIAO_FIN_BYTES in {128, 64, 256}, time Finito batches.
# GPU box.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
for fb in 128 64 256; do
  lib=$("$here/exp_build.sh" fin$fb "-DCIAO_FIN_BYTES=$fb")
  echo "== CIAO_FIN_BYTES=$fb"; (cd "$here/.." && CIAO_HIP_LIB=$lib python tools/finito_batch_time.py 64 256 1024)
done
