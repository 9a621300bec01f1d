set -o pipefail
mkdir -p gpurun_out/long
for cfg in "CIAO_D=32768" "CIAO_D=32768 CIAO_OPTS=long_j=8" "CIAO_D=32768 CIAO_OPTS=split_blocks_per_cu=2" "CIAO_D=16384" "CIAO_D=131072" "CIAO_D=131072 CIAO_OPTS=long_j=8" "CIAO_D=10000" "CIAO_D=65536 CIAO_F32=1" "CIAO_D=65536 CIAO_F32=1 CIAO_OPTS=long_j=8" "CIAO_D=20000 CIAO_F32=1" "CIAO_D=32768 CIAO_TABLE=1 CIAO_GB=4"; do
  env $cfg timeout -k 10 120 python tools/long_rows_time.py 2>&1 | tail -1 | tee -a gpurun_out/long/time4.txt || exit 1
done
