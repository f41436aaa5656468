#!/bin/bash
# usage: tools/pmc_rowout.sh <debug flags...>  -- SQ instruction counters of the semi-stochastic front end (tools/reduce_rowout_time.py) per PYNQS_OP_DEBUG ablation
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r4/pmc_rowout
mkdir -p $out
cd /tmp
for dbg in "$@"; do
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    name=$(echo $c | cut -d' ' -f1)
    PYNQS_OP_DEBUG=$dbg timeout -k 5 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/d${dbg}_$name -- python3 $GRAFT_REPO_ROOT/tools/reduce_rowout_time.py > $out/d${dbg}_$name.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4/pmc_rowout"
for d in sorted(glob.glob(out + "/d*_SQ_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "reduce" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        print(os.path.basename(d), k, {c: round(x / cnt[(k, c)] / 1e6, 2) for c, x in v.items()}, "(millions per launch)")
PY
