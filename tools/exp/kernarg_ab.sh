#!/bin/bash
# Round 5: same-box A/B of the by-value kernel argument against the kernel-argument-segment reads (CIAO_KERNARG0) in the THROUGHPUT
# kernels (rows_multi / rows_small / rows_smallb / rows_smallm / rows_long / proshi_chain / chain_wide), where the round-5 record
# runs showed the fp32 sweep 22 % slower than round 4's.  (ON the GPU box.)
#   tools/exp_build.sh byvalue "-DCIAO_KERNARG_BYVALUE" <units to reuse>;  gpurun -- 'bash tools/exp/kernarg_ab.sh'
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out/kernarg_ab"; mkdir -p "$O"; cd "$R"
B="${CIAO_AB_LIB:-$R/build/byvalue/libciao_hip.so}"
for rep in 1 2; do
  for which in new byvalue; do
    if [ $which = new ]; then unset CIAO_HIP_LIB; else export CIAO_HIP_LIB="$B"; fi
    timeout -k 10 200 python3 bench.py --dtype f32 --steps 10 --warmup 2 --no-cpu --no-extras --no-chains 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.readline()); print('$which rep$rep f32 sweep', j['roofline']['kernel'], j['roofline']['kernel_avg_ms'], 'ms')" || exit 1
  done
done
for which in new byvalue; do
  if [ $which = new ]; then unset CIAO_HIP_LIB; else export CIAO_HIP_LIB="$B"; fi
  timeout -k 10 500 python3 tools/run_extras.py > "$O/extras_$which.json" 2> "$O/extras_$which.err" || { tail -5 "$O/extras_$which.err"; exit 1; }
done
for which in new byvalue; do
  if [ $which = new ]; then unset CIAO_HIP_LIB; else export CIAO_HIP_LIB="$B"; fi
  echo "== $which: small rows (tools/small_modes_time.py), long rows (tools/long_rows_time.py)"
  timeout -k 10 300 python3 tools/small_modes_time.py 2>/dev/null | grep -v amdgpu | cut -c1-230 || exit 1
  timeout -k 10 200 python3 tools/long_rows_time.py 2>/dev/null | grep -v amdgpu | cut -c1-230 || exit 1
  CIAO_TABLE=1 timeout -k 10 200 python3 tools/long_rows_time.py 2>/dev/null | grep -v amdgpu | cut -c1-230 || exit 1
done
python3 - "$O/extras_byvalue.json" "$O/extras_new.json" <<'PY'
import json, sys
def flat(d, p=""):
    out = {}
    if isinstance(d, dict):
        for k, v in d.items(): out.update(flat(v, p + "/" + str(k)))
    elif isinstance(d, (int, float)) and not isinstance(d, bool): out[p] = d
    return out
a, b = (flat(json.load(open(f))) for f in sys.argv[1:3])
print("# extras: by-value -> kernel-argument segment, entries that moved by more than 3 %")
for k in sorted(a):
    if k in b and a[k] and abs(b[k] / a[k] - 1) > 0.03 and any(t in k for t in ("us_per", "alg_GBps", "seconds", "per_s")):
        print("%5.2f  %-70s %.4g -> %.4g" % (b[k] / a[k], k, a[k], b[k]))
PY
