#!/bin/bash
# Round 5, VERDICT r4 item 8: which kernels of the built library does the GPU test suite actually LAUNCH?  (ON the GPU box.)
# rocprofv3 --kernel-trace of `pytest -m gpu` (+ the timing tools that drive dispatch corners the tests do not), every process's trace
# reduced to (kernel symbol, launches); tools/reached_kernels.py sets that against the kernels the library holds (tools/kernel_meta.py).
#   gpurun --timeout 1100 -- 'bash tools/exp/reached_kernels.sh'   ->  gpurun_out/reached/launched.tsv, reached.txt;  then
#   grep 'ciao::\|sample_uniform_kernel' gpurun_out/reached/launched.tsv > profiles/r05_launched_kernels.tsv  (tests/test_kernel_reach_record.py holds the
#   built library against it on every CPU test run)
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out/reached"; T=/tmp/reached_trace; rm -rf "$T"; mkdir -p "$O" "$T"; cd /tmp; export TMPDIR=/tmp
cd "$R"
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d "$T" -o "t_%pid%" -- python3 -m pytest tests -m gpu -q -p no:cacheprovider > "$O/pytest.log" 2>&1
rc=$?; echo "pytest under rocprofv3 rc=$rc"; tail -2 "$O/pytest.log"
python3 - "$T" "$O/launched.tsv" <<'PY'
import collections, csv, glob, sys
csv.field_size_limit(1 << 30)
n = collections.Counter()
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
for f in files:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            n[r["Kernel_Name"]] += 1
with open(sys.argv[2], "w") as out:
    for k, v in sorted(n.items()):
        out.write("%d\t%s\n" % (v, k))
print("%d trace files, %d launches, %d distinct kernels" % (len(files), sum(n.values()), len(n)))
PY
python3 tools/reached_kernels.py "$O/launched.tsv" > "$O/reached.txt"; tail -5 "$O/reached.txt"
exit $rc
