# Final records of the other workloads (repository root on the GPU box): cfg3 whole (V2V, with both VAE legs),
# cfg4 and cfg5 (A14B, both experts on one GPU).
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/f_cfg3.json 2> gpurun_out/f_cfg3.err || exit 1
echo cfg3 done
timeout -k 10 500 python bench.py --workload cfg4 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/f_cfg4.json 2> gpurun_out/f_cfg4.err || exit 1
echo cfg4 done
timeout -k 10 500 python bench.py --workload cfg5 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/f_cfg5.json 2> gpurun_out/f_cfg5.err || exit 1
echo cfg5 done
