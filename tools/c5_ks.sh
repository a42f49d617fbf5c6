#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MGL_NO_AUTOBUILD=1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ks_c5 && mkdir -p $R/gpurun_out/ks_c5
cat > /tmp/c5run.py <<PY
import sys
sys.path.insert(0, "$R")
from megalania_amd import binding, corpus
data,_=corpus.config_input("c5")
sa=binding.SA(data, neighbours_per_step=4096, pb=2, max_bucket_scan=4096, iters_per_epoch=len(data), accept="bulk")
sa.run(60); sa.set_accept_mode("single"); sa.run(12); sa.close()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ks_c5/ks -- python3 /tmp/c5run.py > $R/gpurun_out/ks_c5/log 2>&1
cd $R
python3 tools/pmc_to_json.py gpurun_out/ks_c5 c5 10 gpurun_out/ks_c5/summary.json > /dev/null 2>&1
python3 - <<PY
import json
d = json.load(open("gpurun_out/ks_c5/summary.json"))
print("c5: span per step %.1f us" % d["span_us_per_step"])
for k, v in list(d["kernels"].items())[:8]:
    print(f'{k[:50]:50s} x{v["launches_per_step"]:.2f} avg {v["avg_us"]:9.1f} us  per step {v["us_per_step"]:9.1f} us')
PY
find gpurun_out/ks_c5/ks -type f -delete
