# rocprofv3 on the 8(f) rows (pose chain + association at C3 batch scale): tools/_pose_prof.sh <tag>
# stats pass (one context, nothing beside a traced kernel) + an SQ counter pass for occupancy / issue numbers.
set -e
TAG=${1:-r06_pose_chain}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/pose_chain_bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/tools/pose_chain_bench.py --steps 5 --no-prof > $O/stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/sq -o p --output-format csv -- python3 $R/tools/pose_chain_bench.py --steps 2 --no-prof --pmc-calibrate > $O/sq.log 2>&1
cd $R
python3 tools/prof_summary.py $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_summary.csv > /dev/null
python3 tools/sq_summary.py $(find $O/sq -name "*counter_collection.csv" | head -1) $O/sq_counters.csv > /dev/null
rm -rf $O/stats $O/sq
cat $O/bench.json
cat $O/kernel_stats_summary.csv
