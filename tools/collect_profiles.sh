#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines, rocprofv3 kernel-trace stats of the bench command and
# the PMC passes (separate runs, kernel filter - rocprofv3 segfaults in PyTorch's own kernels
# otherwise), summarised into gpurun_out/ (the raw databases stay in /tmp: too large to ship back).
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r06 [tag-suffix]'
set -u
TAG=${1:-r06}${2:+_$2}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FILTER='gemm_|attn_|ln_mod|qkv_split|solver_step|dac_|rows_add|latent_rows|gather_rows|add_periodic|cast_kernel|step_increment|rows_to_planes'

# PMC passes on 2 loop iterations (profile_run.py --no-dac), one counter set per run (FETCH_SIZE and
# WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots")
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$FILTER" -d /tmp/pmc_$c -o p -- python $R/tools/profile_run.py --iters 2 --no-dac > /tmp/pmc_$c.log 2>&1
done
python $R/tools/pmc_summary.py --traffic-json $OUT/${TAG}_pmc_traffic.json --iters 2 --workload c2/bs1/bf16/xxl \
  $(find /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE -name "*.db") > $OUT/${TAG}_pmc_mem.md
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json   # bench.py reports roofline.traffic from the file whose source hash matches

rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-include-regex "$FILTER" -d /tmp/pmc_sq -o p -- python $R/tools/profile_run.py --iters 2 --no-dac > /tmp/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py --mfma-json $OUT/${TAG}_pmc_mfma.json --workload c2/bs1/bf16/xxl $(find /tmp/pmc_sq -name "*.db") > $OUT/${TAG}_pmc_sq.md
cp $OUT/${TAG}_pmc_mfma.json $R/profiles/${TAG}_pmc_mfma.json

# kernel-trace stats of the bench command itself (1 timed pass + the event-timed loop / decode) - BEFORE the headline run, so that its line
# carries roofline.frac_rocprof from this very collection (bench.py matches the kernel-source hash)
rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /tmp/kt1.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/kt1 -name "*.db" | head -1) --dominant-json $OUT/${TAG}_rocprof_dominant.json --workload c2/bs1/bf16/xxl > $OUT/${TAG}_bench_bs1_kernel_stats.md
cp $OUT/${TAG}_rocprof_dominant.json $R/profiles/${TAG}_rocprof_dominant.json
python $R/bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench_c2.json 2> $OUT/${TAG}_bench_c2.err
tail -c 400 $OUT/${TAG}_bench_c2.json
python $R/bench.py --config c3 --with-encoders --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_bench_c3.json 2> $OUT/${TAG}_bench_c3.err
python $R/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_bench_c4_1gpu.json 2> $OUT/${TAG}_bench_c4.err
python $R/bench.py --config c5 --steps 2 --warmup 1 --no-extra > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err
python $R/bench.py --precision fp16 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_bench_c2_fp16.json 2> $OUT/${TAG}_bench_fp16.err

rocprofv3 --kernel-trace --stats -d /tmp/kt8 -o kt -- python $R/bench.py --steps 1 --warmup 0 --bs 8 --no-cpu-baseline --no-extra > /tmp/kt8.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/kt8 -name "*.db" | head -1) > $OUT/${TAG}_bench_bs8_kernel_stats.md
rocprofv3 --kernel-trace --stats -d /tmp/kt3 -o kt -- python $R/bench.py --config c3 --with-encoders --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /tmp/kt3.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/kt3 -name "*.db" | head -1) --all > $OUT/${TAG}_bench_c3_kernel_stats.md   # --all: the encoders' torch glue kernels are part of this configuration
rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o kt -- python $R/bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /tmp/kt5.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/kt5 -name "*.db" | head -1) > $OUT/${TAG}_bench_c5_kernel_stats.md

# the same SQ counters at 8 clips per GPU (the bs=8 half of the metric): matrix-pipe utilisation of the large-grid kernels
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-include-regex "$FILTER" -d /tmp/pmc_sq8 -o p -- python $R/tools/profile_run.py --iters 2 --no-dac --bs 8 > /tmp/pmc_sq8.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pmc_sq8 -name "*.db") > $OUT/${TAG}_pmc_sq_bs8.md
# per-launch timeline of one DAC decode against the per-launch roofline bound (fp32 matrix peak / HBM), bs = 1 and 8
for b in 1 8; do
  rocprofv3 --kernel-trace -d /tmp/dt$b -o dt -- python $R/tools/dac_trace.py --bs $b > /tmp/dt$b.log 2>&1
  python $R/tools/dac_trace.py --bs $b --summarise /tmp/dt$b > $OUT/${TAG}_dac_trace_bs$b.md 2>&1
done
python $R/tools/attn_bench.py > $OUT/${TAG}_attn_bench.txt 2>&1
python $R/tools/wide_bench.py --m 4000 --cases w13,w2,lin2,fc1,fc2,proj --tiles 0,23,31 --rounds 12 > $OUT/${TAG}_wide_bench.txt 2>&1
python $R/tools/wide_bench.py --m 3000 --cases w13,w2,lin2,fc1,fc2,proj --tiles 0,23,31 --rounds 12 --fp8 >> $OUT/${TAG}_wide_bench.txt 2>&1
ls -la $OUT | tail -18
