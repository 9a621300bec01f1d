#!/bin/bash
# instruction counters of the SVRG chain kernel (evidence for DESIGN 3.2: instructions per step of the one wave per SIMD)
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/chain_pmc"
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/$tag" -o c -- python3 "$R/tools/chain_time.py" > "$O/$tag.log" 2>&1
  echo "rc=$? $tag"
done
cd "$R"
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/chain_pmc/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chain_dma_kernel" in r["Kernel_Name"]:
            key = r["Kernel_Name"].split("(")[0][-60:]
            acc[key][r["Counter_Name"]] = max(acc[key][r["Counter_Name"]], float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, dict(v))
PY
