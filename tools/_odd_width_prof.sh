# rocprofv3 kernel stats of the front-end step at 1278 x 720 (padded internal rows): tools/_odd_width_prof.sh <tag>
set -e
TAG=${1:-r06_odd_width}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/odd_width_bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/tools/odd_width_bench.py --only-odd --steps 5 > $O/stats.log 2>&1
cd $R
python3 tools/prof_summary.py $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_summary.csv > /dev/null
rm -rf $O/stats
cat $O/bench.json
cat $O/kernel_stats_summary.csv
