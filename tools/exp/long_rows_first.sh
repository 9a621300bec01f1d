set -o pipefail
mkdir -p gpurun_out/long
timeout -k 10 500 python -m pytest tests/test_gpu_long_rows.py "tests/test_gpu_parity.py::test_rows_longer_than_lds" -x -q -p no:cacheprovider > gpurun_out/long/tests.log 2>&1; rc=$?; tail -5 gpurun_out/long/tests.log; [ $rc -eq 0 ] || exit $rc
rm -f gpurun_out/long/time_final.txt
for cfg in "CIAO_D=32768" "CIAO_D=32768 CIAO_OPTS=long_rows=0" "CIAO_D=32768 CIAO_OPTS=long_j=4" "CIAO_D=32768 CIAO_OPTS=split_blocks_per_cu=1" "CIAO_D=10000" "CIAO_D=10000 CIAO_OPTS=long_j=4" "CIAO_D=16384" "CIAO_D=65536" "CIAO_D=131072" "CIAO_D=262144" "CIAO_D=65536 CIAO_F32=1" "CIAO_D=65536 CIAO_F32=1 CIAO_OPTS=long_rows=0" "CIAO_D=20000 CIAO_F32=1" "CIAO_D=20000 CIAO_F32=1 CIAO_OPTS=long_j=4" "CIAO_D=262144 CIAO_F32=1" "CIAO_D=32768 CIAO_TABLE=1 CIAO_GB=4" "CIAO_D=32768 CIAO_TABLE=1 CIAO_GB=4 CIAO_OPTS=long_rows=0" "CIAO_D=65536 CIAO_F32=1 CIAO_TABLE=1 CIAO_GB=4"; do
  env $cfg timeout -k 10 120 python tools/long_rows_time.py 2>&1 | tail -1 | tee -a gpurun_out/long/time_final.txt || exit 1
done
cd /tmp && export TMPDIR=/tmp && CIAO_D=32768 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/long/prof -o l -- python3 $GRAFT_REPO_ROOT/tools/long_rows_time.py > $GRAFT_REPO_ROOT/gpurun_out/long/prof.log 2>&1; echo "rocprof rc=$?"
