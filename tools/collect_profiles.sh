#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines, rocprofv3 kernel-trace stats of the bench command and
# the PMC passes (separate runs, kernel filter - rocprofv3 segfaults in PyTorch's own kernels
# otherwise), summarised into gpurun_out/ (the raw databases stay in /tmp: too large to ship back).
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
set -u
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FILTER='gemm_|attn_|ln_mod|qkv_split|solver_step|dac_out|rows_add|latent_rows|gather_rows|add_periodic|cast_kernel|step_increment'

python $R/bench.py --steps 3 --warmup 1 > $OUT/${TAG}_bench_bs1.json 2> $OUT/${TAG}_bench_bs1.err
tail -c 600 $OUT/${TAG}_bench_bs1.json
python $R/bench.py --steps 2 --warmup 1 --bs 8 --no-cpu-baseline > $OUT/${TAG}_bench_bs8.json 2> $OUT/${TAG}_bench_bs8.err

# kernel-trace stats of the bench command itself (1 timed pass + the event-timed loop / decode)
rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/kt1.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/kt1 -name "*.db" | head -1) > $OUT/${TAG}_bench_bs1_kernel_stats.md
rocprofv3 --kernel-trace --stats -d /tmp/kt8 -o kt -- python $R/bench.py --steps 1 --warmup 0 --bs 8 --no-cpu-baseline > /tmp/kt8.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/kt8 -name "*.db" | head -1) > $OUT/${TAG}_bench_bs8_kernel_stats.md

# PMC passes on 2 loop iterations + 1 decode (profile_run.py), one counter set per run
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$FILTER" -d /tmp/pmc_$c -o p -- python $R/tools/profile_run.py --iters 2 > /tmp/pmc_$c.log 2>&1
done
python $R/tools/pmc_summary.py $(find /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE -name "*.db") > $OUT/${TAG}_pmc_mem.md
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-include-regex "$FILTER" -d /tmp/pmc_sq -o p -- python $R/tools/profile_run.py --iters 2 > /tmp/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pmc_sq -name "*.db") > $OUT/${TAG}_pmc_sq.md
ls -la $OUT | tail -12
