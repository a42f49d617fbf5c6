#!/bin/bash
# round-2 baseline evidence on the unchanged kernels: GPU tests, c3 kernel stats, c3 phase shares, c3 PMC traffic
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export MGL_NO_AUTOBUILD=1
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_tests0.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r02_tests0.log
tail -3 gpurun_out/r02_tests0.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02_ks_c3 $R/gpurun_out/r02_pmc_c3_fetch $R/gpurun_out/r02_pmc_c3_write $R/gpurun_out/r02_pmc_c3_sq
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_ks_c3 -- python3 $R/tools/run_steps.py c3 40 40 16384 > $R/gpurun_out/r02_ks_c3.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_c3_fetch -- python3 $R/tools/run_steps.py c3 40 10 16384 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_c3_write -- python3 $R/tools/run_steps.py c3 40 10 16384 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/r02_pmc_c3_sq -- python3 $R/tools/run_steps.py c3 40 10 16384 > /dev/null 2>&1
cd $R
python3 tools/phase_profile.py c3 40 > gpurun_out/r02_phase_c3.log 2>&1
python3 tools/phase_profile.py c2 300 > gpurun_out/r02_phase_c2.log 2>&1
for K in "k_neighbours2<false, 1>" "k_neighbours2<false, 2>" "k_sim"; do python3 tools/pmc_summary.py gpurun_out/r02_pmc_c3_sq "$K"; python3 tools/pmc_summary.py gpurun_out/r02_pmc_c3_fetch "$K"; python3 tools/pmc_summary.py gpurun_out/r02_pmc_c3_write "$K"; done > gpurun_out/r02_pmc_c3_summary.txt 2>&1
python3 - <<PY > gpurun_out/r02_ks_c3_summary.txt
import csv,glob
for f in glob.glob("gpurun_out/r02_ks_c3/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[:24]: print(f'{r["Name"][:60]:60s} calls={r["Calls"]:>6s} avg={float(r["AverageNs"])/1000:9.1f}us pct={r["Percentage"]}')
PY
cat gpurun_out/r02_ks_c3_summary.txt gpurun_out/r02_phase_c3.log gpurun_out/r02_phase_c2.log
# keep the big raw dirs out of the 64 MiB merge: only the CSVs that matter
find gpurun_out/r02_ks_c3 gpurun_out/r02_pmc_c3_fetch gpurun_out/r02_pmc_c3_write gpurun_out/r02_pmc_c3_sq -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete
find gpurun_out -name '*counter_collection.csv' -size +20M -delete
