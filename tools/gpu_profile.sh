# rocprofv3 runs for profiles/ (kernel stats, then PMC passes in their own runs)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
BENCH="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decoupled --no-f32 --no-rollout --no-warm"      # every k_solve launch of the trace is a cold headline launch
# (the full default bench line is taken separately: python bench.py > gpurun_out/bench_full.log)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $BENCH > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $BENCH > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $BENCH > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $BENCH > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $BENCH > $OUT/bench_tcc.log 2>&1
# the fp32 library (config 4's per-GPU share): kernel stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decoupled --no-hji --no-rollout --no-warm --precision f32 --batch 8192 > $OUT/bench_stats_f32.log 2>&1
tail -1 $OUT/bench_stats.log | cut -c1-300
