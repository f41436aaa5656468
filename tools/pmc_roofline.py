#!/usr/bin/env python3
"""Profile one bench.py workload with rocprofv3 and write what bench.py's `roofline` field quotes (run on the GPU box, from the repo root):

    python3 tools/pmc_roofline.py --workload fe2s2_reduce_vmc_step [--walkers N] [--tag r04_reduce_step] [--kernel substring]

Passes (each its own `rocprofv3 ... -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra ...`; counters never share a
run with a trace domain other than --kernel-trace):
    --kernel-trace --stats                          per-kernel calls / average duration
    --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
    --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
    --pmc FETCH_SIZE        --pmc WRITE_SIZE        (separately, as the microarchitecture guide prescribes)
    --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
Output: profiles/<tag>.txt (everything, per kernel) and profiles/pmc_<pmc_name>.json (the dominant kernel's per-launch numbers + the sha256 of
the native sources they were measured on: bench.py prints `"stale": true` instead of a fraction when the tree has changed since).
HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE counts 64-byte requests that each fill a 128-byte line on gfx950:
profiles/r03_fetch_size_calibration.txt)."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pynqs_amd.build import source_hash  # noqa: E402

PASSES = ["SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES",
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU",
          "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True)
    ap.add_argument("--walkers", type=int, default=0)
    ap.add_argument("--tag", default=None)
    ap.add_argument("--kernel", default=None, help="substring of the dominant kernel's name (default: the pynqs kernel with the largest total time)")
    ap.add_argument("--pmc-name", default=None, help="profiles/pmc_<this>.json (default: the workload's name)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--keep-raw", action="store_true", help="keep rocprofv3's CSVs under gpurun_out/pmc_<tag>/ (tens of MB per workload; gpurun copies at most 64 MiB back)")
    ap.add_argument("--passes", default=None, help="other counter passes instead of the default five, ';' between passes (e.g. 'SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS'): everything lands in profiles/<tag>.txt")
    a, extra = ap.parse_known_args()
    passes = [c.strip() for c in a.passes.split(";") if c.strip()] if a.passes else PASSES
    tag = a.tag or f"r04_{a.workload}"
    out = os.path.join(ROOT, "gpurun_out", "pmc_" + tag)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--workload", a.workload] + (["--walkers", str(a.walkers)] if a.walkers else []) + extra
    runs = [("trace", ["--kernel-trace", "--stats"], ["--steps", str(a.steps), "--warmup", "5"])] + \
           [("pmc%d" % i, ["--kernel-trace", "--pmc"] + c.split(), ["--steps", "3", "--warmup", "1"]) for i, c in enumerate(passes)]
    for name, flags, steps in runs:
        d = os.path.join(out, name)
        cmd = ["rocprofv3"] + flags + ["--output-format", "csv", "-d", d, "--"] + bench + steps
        with open(os.path.join(out, name + ".log"), "w") as log:
            rc = subprocess.call(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=600)
        print(f"[pmc_roofline] {name}: rc {rc}", flush=True)
    # ---- summarise ----
    lines, stats = [], {}
    # per-dispatch durations (the MEDIAN is what is quoted: a workload's first launches -- sizing calls with other buffer shapes, a call that still
    # carries a de-duplication table -- are not the steady state, and a run has only a few dozen launches)
    durs = defaultdict(list)
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            durs[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    med = lambda v: sorted(v)[len(v) // 2] if v else None  # noqa: E731
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            stats[r["Name"]] = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), total_ns=float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["AverageNs"]) * int(r["Calls"]))
    lines.append(f"== {tag}: rocprofv3 --kernel-trace --stats of `{' '.join(bench[1:])} --steps {a.steps} --warmup 5` ==")
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"])[:25]:
        mk_ = max(durs, key=lambda q: len(os.path.commonprefix([q, k]))) if durs else None
        lines.append(f"{k[:120]:120s} calls={v['calls']:5d} avg_us={v['avg_ns'] / 1e3:10.2f} median_us={(med(durs[mk_]) if mk_ else 0) / 1e3:10.2f} total_ms={v['total_ns'] / 1e6:9.3f}")
    pmc, meta = defaultdict(lambda: defaultdict(list)), {}
    for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "pynqs" not in k:
                continue
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {x: r.get(x) for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count")}
    lines += ["", "== rocprofv3 --pmc (separate passes), MEDIAN per dispatch =="]
    means = {}
    for k, d in pmc.items():
        means[k] = {c: med(v) for c, v in d.items()}   # (medians: see above)
        lines.append(k[:160])
        lines.append("   " + " ".join(f"{x}={y}" for x, y in meta[k].items()))
        for c, m in sorted(means[k].items()):
            lines.append(f"   {c:28s} {m:18.1f}   (n={len(d[c])})")
    pyn = {k: v for k, v in stats.items() if "pynqs" in k}
    dom = None
    if a.kernel:
        cands = [k for k in pyn if a.kernel in k]
        dom = max(cands, key=lambda k: pyn[k]["total_ns"]) if cands else None
    elif pyn:
        dom = max(pyn, key=lambda k: pyn[k]["total_ns"])
    sha = source_hash()
    lines += ["", f"native sources sha256 {sha}"]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", tag + ".txt"), "w").write("\n".join(lines) + "\n")
    if dom is None:
        print("no pynqs kernel in the trace", file=sys.stderr)
        return 1
    # counters are keyed by the demangled name too: match by the longest common prefix
    mk = max(means, key=lambda k: len(os.path.commonprefix([k, dom]))) if means else None
    m = means.get(mk, {})
    fetch, write = m.get("FETCH_SIZE"), m.get("WRITE_SIZE")
    hit, miss = m.get("TCC_HIT_sum"), m.get("TCC_MISS_sum")
    walkers = a.walkers or 8192
    js = {"source": f"profiles/{tag}.txt (tools/pmc_roofline.py --workload {a.workload}: rocprofv3 --kernel-trace --stats + separate --pmc passes of the tree with this sha256)",
          "csrc_sha256": sha, "workload": a.workload, "walkers": walkers, "kernel": dom,
          "rocprof_kernel_avg_ns": (med(durs[max(durs, key=lambda k: len(os.path.commonprefix([k, dom])))]) if durs else None) or stats[dom]["avg_ns"],
          "rocprof_kernel_mean_ns": stats[dom]["avg_ns"], "rocprof_kernel_calls": stats[dom]["calls"], "statistic": "median per dispatch (durations and counters)",
          "valu_insts_per_launch": m.get("SQ_INSTS_VALU"), "salu_insts_per_launch": m.get("SQ_INSTS_SALU"), "lds_insts_per_launch": m.get("SQ_INSTS_LDS"),
          "wave_cycles": m.get("SQ_WAVE_CYCLES"), "wait_any_cycles": m.get("SQ_WAIT_ANY"), "busy_cycles": m.get("SQ_BUSY_CYCLES"),
          "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
          "hbm_bytes_per_launch": (2 * fetch + write) * 1024 if fetch is not None and write is not None else None,
          "correction": "FETCH_SIZE x 2: the counter counts 64-byte requests, every request fills a 128-byte line (profiles/r03_fetch_size_calibration.txt); WRITE_SIZE as is. Both count L2 misses, i.e. traffic to the Infinity Cache / HBM side",
          "TCC_HIT": hit, "TCC_MISS": miss, "TCC_REQ": m.get("TCC_REQ_sum"), "l2_hit_rate": hit / (hit + miss) if hit is not None and miss else None}
    name = a.pmc_name or a.workload
    json.dump(js, open(os.path.join(ROOT, "profiles", f"pmc_{name}.json"), "w"), indent=1)
    print(json.dumps(js, indent=1))
    if not a.keep_raw:
        import shutil

        shutil.rmtree(out, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
