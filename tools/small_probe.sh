#!/bin/bash
# rows_small_kernel experiments (plain vs non-temporal loads), experiment builds under build/. GPU box.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
for fl in "-DCIAO_SMALL_NT" "-DCIAO_SMALL_PLAIN"; do
  lib=$("$here/exp_build.sh" small$(echo $fl | tr -cd 'A-Z_') "$fl")
  for cfg in "50 f64" "50 f32"; do set -- $cfg
    (cd "$here/.." && CIAO_HIP_LIB=$lib python bench.py --rows-per-gpu 4000000 --d $1 --dtype $2 --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('flags[$fl]', j['config']['d'], j['dtype'], round(j['roofline']['achieved']), j['roofline']['kernel'])")
  done
done
