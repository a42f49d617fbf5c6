#!/bin/bash
# Registers, scratch and spills of every kernel of the library, as the compiler reports them (gfx950): the table to look at
# after any change to a kernel -- a few more live values in a wrapper cost the pick kernel 54 spilled VGPRs once.
#   bash tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage -o /dev/null $R/megalania_amd/csrc/mgl_api.hip 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//' |
  python3 -c '
import subprocess, sys
rows, cur = [], None
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith("Function Name:"):
        cur = [ln.split(":", 1)[1].strip()]; rows.append(cur)
    elif cur is not None:
        cur.append(ln)
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, nm in zip(rows, names):
    print(nm.split("(")[0].replace("void ", ""), "|", " | ".join(r[1:]))
'
