# baseline / after profile of the grid extractor: tools/_grid_prof.sh <tag>
set -e
TAG=${1:-r06_grid_before}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/grid_bench.py --frames 64 > $O/bench_hard.json 2> $O/bench_hard.err
timeout -k 10 300 python3 tools/grid_bench.py --frames 64 --data easy > $O/bench_easy.json 2>> $O/bench_hard.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/tools/grid_bench.py --frames 64 --steps 5 --no-prof > $O/stats.log 2>&1
cd $R
python3 tools/prof_summary.py $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_summary.csv > /dev/null
rm -rf $O/stats
cat $O/bench_hard.json $O/bench_easy.json
cat $O/kernel_stats_summary.csv
