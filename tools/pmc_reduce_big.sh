#!/bin/bash
# usage: tools/pmc_reduce_big.sh <tag> <case>   e.g. tools/pmc_reduce_big.sh syn120 120:30:4096:0.49995
# rocprofv3 PMC passes (each in its own run, with --kernel-trace only) over tools/reduce_big_one.py: the flushing LIST kernel on long rows
tag=$1; c=$2
out=gpurun_out/pq_$tag
mkdir -p $out; export TMPDIR=/tmp
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
         "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  name=$(echo $pmc | tr ' ' '+' | cut -c1-50)
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc_$name -- python3 tools/reduce_big_one.py $c 1 5 > $out/pmc_$name.log 2>&1 || echo "pmc pass $pmc failed" >> $out/errors.log
  echo "pass $name done"
done
python3 - <<PY
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "flush_kernel" in k or "children_wave" in k:
            res[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print(f"   {c:28s} launches {len(v):3d}  min {v[0]:.4g}  median {v[len(v)//2]:.4g}  max {v[-1]:.4g}")
PY
