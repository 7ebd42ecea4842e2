cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g1_prof_cfg2 -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $R/gpurun_out/g1_cfg2_profiled.json 2> $R/gpurun_out/g1_cfg2_profiled.err; echo cfg2 rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g1_prof_cfg3 -o b -- python $R/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $R/gpurun_out/g1_cfg3_profiled.json 2> $R/gpurun_out/g1_cfg3_profiled.err; echo cfg3 rc=$?
cd $R
timeout -k 10 500 python bench.py --workload cfg4 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/g1_cfg4.json 2> gpurun_out/g1_cfg4.err; echo cfg4 rc=$?
timeout -k 10 500 python bench.py --workload cfg5 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/g1_cfg5.json 2> gpurun_out/g1_cfg5.err; echo cfg5 rc=$?
