#!/bin/bash
# Round 5: the parity log that the stated scales are derived from (tools/retune_scales.py) -- the whole GPU suite once, then the seeded random
# chain / shape tests under four more seeds with 2500 / 400 cases each, merged into ONE log (the worst error per call site): a scale is 10 x the
# largest error over all of them.  (ON the GPU box; CIAO_PARITY_CALIBRATE=1: nothing asserts on the bounds while they are being measured.)
#   gpurun --timeout 1200 -- 'bash tools/exp/calibrate_seeds.sh'  ->  gpurun_out/parity_observed.json;  python tools/retune_scales.py gpurun_out/parity_observed.json --write
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
export CIAO_PARITY_CALIBRATE=1
rm -f gpurun_out/parity_observed.json
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -2 || exit 1
export CIAO_PARITY_LOG_MERGE=1
for seed in 11 424242 2027 31337; do
  CIAO_FUZZ_CASES=2500 CIAO_FUZZ_SHAPES=400 CIAO_FUZZ_SEED=$seed timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_random_chain_configurations or test_random_shapes" 2>&1 | tail -1 || exit 1
done
