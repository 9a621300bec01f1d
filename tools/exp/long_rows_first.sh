set -o pipefail
mkdir -p gpurun_out/long
true
for cfg in "CIAO_D=32768" "CIAO_D=32768 CIAO_OPTS=long_rows=0" "CIAO_D=32768 CIAO_OPTS=long_j=8" "CIAO_D=32768 CIAO_OPTS=split_blocks_per_cu=2" "CIAO_D=32768 CIAO_OPTS=split_blocks_per_cu=3" "CIAO_D=32768 CIAO_OPTS=long_j=8,split_blocks_per_cu=1" "CIAO_D=16384" "CIAO_D=131072" "CIAO_D=131072 CIAO_OPTS=long_j=8" "CIAO_D=10000" "CIAO_D=65536 CIAO_F32=1" "CIAO_D=65536 CIAO_F32=1 CIAO_OPTS=long_rows=0" "CIAO_D=65536 CIAO_F32=1 CIAO_OPTS=long_j=8" "CIAO_D=20000 CIAO_F32=1" "CIAO_D=32768 CIAO_TABLE=1 CIAO_GB=4" "CIAO_D=32768 CIAO_TABLE=1 CIAO_GB=4 CIAO_OPTS=long_rows=0"; do
  env $cfg timeout -k 10 120 python tools/long_rows_time.py 2>&1 | tail -2 | tee -a gpurun_out/long/time.txt || exit 1
done
