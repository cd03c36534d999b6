# rocprofv3 runs for profiles/ (kernel stats, then PMC passes in their own runs: gpurun refuses --pmc together with the trace domains).  PART = 1 | 2 | 3 (a gpurun
# call is capped at 20 minutes): 1 = the cold fp64 headline (stats + HBM + SQ + matrix-core counters), 2 = config 5 (lateral N = 50 + walls), config 3 (fp32 + HJI row) and
# the closed-loop rollouts (stats; HBM and SQ counters for configs 5 and 3), 3 = the fp32 library at 8192 per GPU (config 4's share).   usage: PART=1 bash tools/gpu_profile.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PART=${PART:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
HEAD="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decoupled --no-f32 --no-rollout --no-warm --no-hji"      # every k_solve launch of the trace is a cold headline launch
pmc() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $3 > $OUT/bench_$1.log 2>&1 || echo "pass $1 failed (counter not available here?)"; }
if [ "$PART" = "1" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $HEAD > $OUT/bench_stats.log 2>&1
  pmc pmc_fetch "FETCH_SIZE" "$HEAD"
  pmc pmc_write "WRITE_SIZE" "$HEAD"
  pmc pmc_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "$HEAD"
  pmc pmc_mfma "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "$HEAD"
  pmc pmc_sq2 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "$HEAD"
  pmc pmc_flops "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "$HEAD"      # round 6: the hardware's own count of executed fp64 arithmetic (flops = 64 x (ADD + MUL + TRANS + 2 FMA))
  pmc pmc_wr "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_WRITE_SECTORS_sum" "$HEAD"               # what k_solve's 45 MB of write-back are made of
  pmc pmc_wb "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_WRITEBACK_sum TCC_EA0_WR_UNCACHED_32B_sum" "$HEAD"
  pmc pmc_tcc "TCC_HIT_sum TCC_MISS_sum" "bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decoupled --no-f32 --no-rollout --no-warm"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_hji -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decoupled --no-f32 --no-rollout --no-warm > $OUT/bench_stats_hji.log 2>&1      # + the HJI lookups (three layouts)
  pmc pmc_fetch_hji "FETCH_SIZE" "bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decoupled --no-f32 --no-rollout --no-warm"
  pmc pmc_write_hji "WRITE_SIZE" "bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decoupled --no-f32 --no-rollout --no-warm"
fi
if [ "$PART" = "2" ]; then
  DEC="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32 --no-rollout --no-warm --no-hji"               # + config 5: k_nodes_dec, k_qp_dec, k_solve_lat
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dec -- python3 $DEC > $OUT/bench_stats_dec.log 2>&1
  pmc pmc_fetch_dec "FETCH_SIZE" "$DEC"
  pmc pmc_write_dec "WRITE_SIZE" "$DEC"
  pmc pmc_sq_dec "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "$DEC"
  pmc pmc_sq2_dec "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "$DEC"
  pmc pmc_flops_dec "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "$DEC"
  ROLL="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decoupled --no-f32 --no-warm --no-hji"            # + closed loop: k_nodes_warm, k_advance, warm k_solve
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_roll -- python3 $ROLL > $OUT/bench_stats_roll.log 2>&1
  C3="tools/gpu_config3_probe.py f32"   # config 3 alone: the fp32 library, B = 4096, safety row on the 13 x 13 x 9^5 grid, five cold steps (every k_solve launch of the trace is a config-3 launch)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $C3 > $OUT/bench_stats_c3.log 2>&1
  pmc pmc_fetch_c3 "FETCH_SIZE" "$C3"
  pmc pmc_write_c3 "WRITE_SIZE" "$C3"
  pmc pmc_sq_c3 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "$C3"
  pmc pmc_flops_c3 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "$C3"
  DECLOOP="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32 --no-warm --no-hji"                        # + the lateral closed loops (warm k_solve_lat, k_advance)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_decloop -- python3 $DECLOOP > $OUT/bench_stats_decloop.log 2>&1
fi
if [ "$PART" = "3" ]; then
  F32="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decoupled --no-hji --no-rollout --no-warm --precision f32 --batch 8192"      # config 4's per-GPU share
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -- python3 $F32 > $OUT/bench_stats_f32.log 2>&1
  pmc pmc_fetch_f32 "FETCH_SIZE" "$F32"
  pmc pmc_write_f32 "WRITE_SIZE" "$F32"
  pmc pmc_sq_f32 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "$F32"
  pmc pmc_mfma_f32 "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "$F32"
  pmc pmc_flops_f32 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "$F32"
fi
ls $OUT | head -40
