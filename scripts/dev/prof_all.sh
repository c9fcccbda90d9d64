set -e
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/final; mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --headline-only > $O/p_bench.log 2>&1
echo step1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_other -- python3 $R/scripts/profile_other_rows.py > $O/p_other.log 2>&1
echo step2 done
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/p_sqf -- python3 $R/scripts/one_gram.py 3 > $O/p_sqf.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/p_sqq -- python3 $R/scripts/dev/one_quad.py > $O/p_sqq.log 2>&1
echo step3 done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- python3 $R/scripts/one_gram.py 3 > $O/p_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_write -- python3 $R/scripts/one_gram.py 3 > $O/p_write.log 2>&1
echo step4 done
cd $R
for m in sym ordered fwd; do python scripts/dev/phase_stamps.py 1024 64 7 $m 2>&1 | grep "phase stamps" >> $O/stamps.txt; done
python scripts/dev/phase_stamps.py 128 32 7 sym 2>&1 | grep "phase stamps" >> $O/stamps.txt
python scripts/dev/phase_stamps.py 256 128 14 sym 2>&1 | grep "phase stamps" >> $O/stamps.txt
python scripts/dev/phase_stamps.py 256 128 14 ordered 2>&1 | grep "phase stamps" >> $O/stamps.txt
echo step5 done
python scripts/shard_cost.py > $O/shard_cost.txt 2>&1 || true
tail -5 $O/shard_cost.txt
