#!/bin/bash
# rows_small_kernel experiments (plain vs non-temporal loads). GPU box.
set -e
cd "$(dirname "$0")/../ciaoalgorithms.jl_amd/csrc"
for fl in "" "-DCIAO_SMALL_PLAIN"; do
  rm -f rows_f64.o rows_f32.o
  make -s -j8 EXTRA="$fl" ../libciao_hip.so >/dev/null 2>&1
  for cfg in "50 f64" "50 f32"; do set -- $cfg
    (cd ../.. && python bench.py --rows-per-gpu 4000000 --d $1 --dtype $2 --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('flags[$fl]', j['config']['d'], j['dtype'], round(j['roofline']['achieved']), j['roofline']['kernel'])")
  done
done
rm -f rows_f64.o rows_f32.o
make -s -j8 ../libciao_hip.so >/dev/null 2>&1
