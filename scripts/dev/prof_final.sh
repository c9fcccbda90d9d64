set -e
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/final3; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_other -- python3 $R/scripts/profile_other_rows.py > $O/p_other.log 2>&1
echo step1 done
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 200 $O/bench.json
