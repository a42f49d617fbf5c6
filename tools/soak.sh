#!/bin/bash
# Reference schedule (main.c:66-70: 3 phases x 200 epochs x N iterations) through the C driver on c2-shaped input.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 - <<PY
from megalania_amd import corpus
open("gpurun_out/soak_c2.bin","wb").write(corpus.config_input("c2")[0])
PY
START=$(date +%s%N)
./megalania_amd/_build/megalania-hip -o gpurun_out/soak_c2.lzma gpurun_out/soak_c2.bin 2> gpurun_out/soak_c2.err
END=$(date +%s%N)
tail -2 gpurun_out/soak_c2.err
xz -dc --format=lzma gpurun_out/soak_c2.lzma | cmp - gpurun_out/soak_c2.bin && echo "round trip ok: $(stat -c %s gpurun_out/soak_c2.lzma) bytes in $(( (END - START) / 1000000 )) ms (600 epochs x 25 steps x 4096 neighbours)"
rm -f gpurun_out/soak_c2.bin
