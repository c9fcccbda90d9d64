# Round-4 evidence pass on the GPU box (rocprofv3 directly in front of python3; --pmc passes are separate runs with
# --kernel-trace only).  Output under gpurun_out/r4_final/; the summaries worth keeping are copied to profiles/ by hand.
set -e
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r4_final; mkdir -p $O
export SIGSVGD_REVISION=$(cat $R/sigsvgd_amd/_exp/.revision 2>/dev/null || echo unknown)
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
SQ2="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --headline-only > $O/p_bench.log 2>&1
echo step1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_other -- python3 $R/scripts/profile_other_rows.py > $O/p_other.log 2>&1
echo step2 done
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/p_sqf -- python3 $R/scripts/one_gram.py 3 > $O/p_sqf.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/p_sqq -- python3 $R/scripts/dev/one_quad.py > $O/p_sqq.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/p_sqb -- python3 $R/scripts/dev/one_refined.py > $O/p_sqb.log 2>&1
echo step3 done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- python3 $R/scripts/one_gram.py 3 > $O/p_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_write -- python3 $R/scripts/one_gram.py 3 > $O/p_write.log 2>&1
echo step4 done
cd $R
python3 scripts/pmc_summary.py traffic $O/r04_pmc_hbm_traffic.csv $O/p_fetch $O/p_write
python3 scripts/pmc_summary.py sq $O/r04_sq_counters_gram_fast.csv $O/p_sqf
python3 scripts/pmc_summary.py sq $O/r04_sq_counters_gram_quad.csv $O/p_sqq
python3 scripts/pmc_summary.py sq $O/r04_sq_counters_refined.csv $O/p_sqb
for m in sym ordered; do python3 scripts/dev/phase_stamps.py 1024 64 7 $m 2>&1 | grep "phase stamps" >> $O/stamps.txt; done
python3 scripts/dev/phase_stamps.py 512 64 3 sym 2>&1 | grep "phase stamps" >> $O/stamps.txt
python3 scripts/dev/phase_stamps.py 256 128 14 sym 2>&1 | grep "phase stamps" >> $O/stamps.txt
for a in "100 10 2 dyadic4" "35 30 2 dyadic3" "150 10 2 dyadic4" "16 20 2 dyadic2"; do python3 scripts/dev/phase_stamps.py $a 2>&1 | grep "phase stamps band" | tail -1 >> $O/stamps.txt; done
echo step5 done
python3 scripts/shard_cost.py > $O/shard_cost.txt 2>&1 || true
tail -8 $O/shard_cost.txt
