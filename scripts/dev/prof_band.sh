# SQ counters of the band kernels on the notebook and maze shapes (both modes); rocprofv3 directly in front of python3
set -e
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/band_prof; rm -rf $O; mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
SQ2="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq1 -- python3 $R/scripts/dev/band_modes.py 100 10 2 4 35 30 2 3 > $O/sq1.log 2>&1
if [ "$1" = "all" ]; then
rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/sq2 -- python3 $R/scripts/dev/band_modes.py 100 10 2 4 35 30 2 3 > $O/sq2.log 2>&1
fi
cd $R
for k in sq1 sq2; do [ -d $O/$k ] && python3 scripts/pmc_summary.py sq $O/$k.csv $O/$k || true; done
