cd $GRAFT_REPO_ROOT
python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r02_e_bench.json 2> gpurun_out/r02_e_bench.err; tail -c 300 gpurun_out/r02_e_bench.err
python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r02_e_bench_c5.json 2> gpurun_out/r02_e_bench_c5.err
python bench.py --config c5 --quantization none --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r02_e_bench_c5_noq.json 2>> gpurun_out/r02_e_bench_c5.err
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
