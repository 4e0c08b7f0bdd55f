set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_c2 -o c2 -- python3 $R/bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 > $R/gpurun_out/final/prof_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_c3 -o c3 -- python3 $R/bench.py --config 3 --no-extra --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/final/prof_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_c3e8 -o c3e8 -- python3 $R/bench.py --config 3 --emulate-world 8 --no-extra --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/final/prof_c3e8.log 2>&1
cd $R
for t in c2 c3 c3e8; do cp $(find gpurun_out/final/prof_$t -name "*kernel_stats.csv" | head -1) gpurun_out/final/${t}_kernel_stats.csv; rm -rf gpurun_out/final/prof_$t; grep "^{" gpurun_out/final/prof_$t.log | cut -c1-260; done
head -4 gpurun_out/final/c2_kernel_stats.csv | cut -c1-200
