R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_k20.json 2> gpurun_out/r02_bench_k20.err
timeout -k 10 900 python scripts/bench_configs.py --graphed > gpurun_out/r02_other_configs.jsonl 2> gpurun_out/r02_other_configs.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02_prof_bench $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w $R/gpurun_out/r02_prof_E
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_bench -o b -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r02_prof_bench.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_E -o e -- python3 $R/scripts/bench_configs.py --graphed E > $R/gpurun_out/r02_prof_E.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f -o f -- python3 $R/bench.py --steps 20 --warmup 8 --no-cpu-baseline --adam-steps 0 > $R/gpurun_out/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -o w -- python3 $R/bench.py --steps 20 --warmup 8 --no-cpu-baseline --adam-steps 0 > $R/gpurun_out/pmc_w.log 2>&1
echo done
