#!/bin/bash
# finalize_kernel column-width experiment: rebuild with CIAO_FIN_BYTES in {128, 64, 256}, time Finito batches. GPU box.
set -e
cd "$(dirname "$0")/../ciaoalgorithms.jl_amd/csrc"
for fb in 128 64 256; do
  rm -f rows_f64.o rows_f32.o
  make -s -j8 EXTRA="-DCIAO_FIN_BYTES=$fb" ../libciao_hip.so >/dev/null 2>&1
  echo "== CIAO_FIN_BYTES=$fb"; (cd ../.. && python tools/finito_batch_time.py 64 256 1024)
done
rm -f rows_f64.o rows_f32.o
make -s -j8 ../libciao_hip.so >/dev/null 2>&1
