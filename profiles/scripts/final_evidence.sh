# final evidence of the round's second session: default bench line, kernel stats of configs[3] at N = 1 and emulated N = 8
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err || { tail -30 gpurun_out/final/bench_default.err; exit 1; }
tail -c 600 gpurun_out/final/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --no-extra --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/final/prof_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof_c3e8 -o c3e8 -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --emulate-world 8 --no-extra --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/final/prof_c3e8.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/final/prof_c3 -name "*kernel_stats.csv" | head -1) gpurun_out/final/c3_kernel_stats.csv
cp $(find gpurun_out/final/prof_c3e8 -name "*kernel_stats.csv" | head -1) gpurun_out/final/c3e8_kernel_stats.csv
rm -rf gpurun_out/final/prof_c3 gpurun_out/final/prof_c3e8
grep "^{" gpurun_out/final/prof_c3.log | cut -c1-300
grep "^{" gpurun_out/final/prof_c3e8.log | cut -c1-300
