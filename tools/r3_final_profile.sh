# Final records of the round (run from the repository root on the GPU box): cfg2 bench, its kernel stats, the VAE's
# matrix-pipe counters on the conv kernel, the VAE per-kernel stats and the clock held during a VAE decode.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f_prof_cfg2 -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $R/gpurun_out/f_cfg2_profiled.json 2> $R/gpurun_out/f_cfg2_profiled.err || exit 1
echo profiled cfg2 done
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/f_pmc_vae -o p -- python $R/tools/bench_kernels.py vae_small > $R/gpurun_out/f_pmc_vae.log 2>&1 || exit 1
echo pmc vae done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f_prof_vae -o v -- python $R/tools/bench_kernels.py vae > $R/gpurun_out/f_prof_vae.log 2>&1 || exit 1
echo vae stats done
cd $R
timeout -k 10 120 python -u tools/clock_sample.py 6 > gpurun_out/f_clock_vae.txt 2>&1 || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/f_bench_cfg2.json 2> gpurun_out/f_bench_cfg2.err || exit 1
tail -c 600 gpurun_out/f_bench_cfg2.json
