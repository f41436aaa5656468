#!/bin/bash
# usage: tools/pmc_one.sh <tag> "<counters>" <program args...>   -- one rocprofv3 PMC pass, prints per-kernel means
tag=$1; ctr=$2; shift 2
out=gpurun_out/pmc1_$tag
mkdir -p $out; export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python3 "$@" > $out/log.txt 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pynqs" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print(k[0], k[1], sum(v) / len(v), len(v))
PY
