set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_v5
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v5 -- python3 $R/bench.py --steps 500 --warmup 100 --no-cpu > $R/gpurun_out/bench_v5_prof.log 2>&1
bash $R/tools/collect_pmc.sh > $R/gpurun_out/pmc_v5.log 2>&1
cd $R
python3 bench.py > gpurun_out/bench_r01_v5.json
python3 bench.py --config c3 --steps 200 --warmup 20 --no-cpu > gpurun_out/bench_r01_v5_c3.json
python3 bench.py --config c5 --steps 300 --warmup 30 --no-cpu > gpurun_out/bench_r01_v5_c5.json
tail -c 600 gpurun_out/bench_r01_v5.json
