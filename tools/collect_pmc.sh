#!/bin/bash
# HBM traffic of the three kernels of the regular neighbour launch (k_neighbours2<false, 1|2> + k_sim) per step, the way MI355X_MICROARCH.md prescribes: counters in
# their own rocprofv3 passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2), no trace domains mixed in.
# Usage (on the GPU box, from the repo root):  bash tools/collect_pmc.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pmc_fetch $ROOT/gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_fetch -- python3 $ROOT/tools/run_steps.py c2 300 40 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_write -- python3 $ROOT/tools/run_steps.py c2 300 40 > /dev/null 2>&1
python3 - <<PY
import csv, glob, json
def avg(d, counter):
    rows = [r for f in glob.glob("$ROOT/gpurun_out/" + d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
            if ("k_neighbours2<false" in r["Kernel_Name"] or r["Kernel_Name"].startswith("k_sim")) and r["Counter_Name"] == counter]
    # the regular neighbour launch is three kernels (pick + rest + re-simulation) per step: sum them, per step
    steps = len({r["Dispatch_Id"] for r in rows if "<false, 2>" in r["Kernel_Name"]}) or len({r["Dispatch_Id"] for r in rows})
    return sum(float(r["Counter_Value"]) for r in rows) / max(1, steps), steps
f, nf = avg("pmc_fetch", "FETCH_SIZE")
w, nw = avg("pmc_write", "WRITE_SIZE")
out = dict(kernel="k_neighbours2<false, PICK> + k_neighbours2<false, REST> + k_sim", workload="c2, 4096 neighbours/launch, after 300 warm-up steps", launches=nf,
           FETCH_SIZE_KB=f, WRITE_SIZE_KB=w,
           note="FETCH_SIZE/WRITE_SIZE are in KB (rocprofv3 derived metrics).  On gfx950 FETCH_SIZE counts 128-B requests as 64 B "
                "for wide streaming reads (MI355X_MICROARCH.md, HBM); this kernel reads narrow scattered lines, so both the raw "
                "and the doubled figure are given; bytes_per_launch uses the doubled (upper) one.",
           fetch_bytes_raw=f * 1024, fetch_bytes_doubled=2 * f * 1024, write_bytes=w * 1024,
           bytes_per_launch=2 * f * 1024 + w * 1024)
json.dump(out, open("$ROOT/profiles/pmc_traffic.json", "w"), indent=1)
json.dump(out, open("$ROOT/gpurun_out/pmc_traffic.json", "w"), indent=1)  # gpurun merges only gpurun_out/ back: copy it into profiles/
print(json.dumps(out))
PY
