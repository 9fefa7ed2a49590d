# Randomised parity and determinism runs on the build as it stands: tools/soak_all.sh <tag> [seconds per fuzzer]
# Each leg writes its own file under gpurun_out/<tag>/ (and a line to summary.txt) so a long call shows progress.
TAG=${1:-r06_soak}
SECS=${2:-240}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python3 -c "import sys, json; sys.path.insert(0, '.'); import bench; print(json.dumps(bench.source_stamp(), indent=1))" > $O/source_stamp.txt
SEED=$(( $(date +%s) % 100000 ))
echo "seed base $SEED, $SECS s per fuzzer" > $O/summary.txt
for f in extract match ransac grid assoc pose; do
  timeout -k 10 $((SECS + 120)) python3 tests/fuzz_$f.py $((SEED + 1)) $SECS > $O/fuzz_$f.log 2>&1 || { echo "fuzz_$f FAILED" >> $O/summary.txt; tail -5 $O/fuzz_$f.log >> $O/summary.txt; exit 1; }
  echo "fuzz_$f: $(grep -v amdgpu.ids $O/fuzz_$f.log | tail -2 | tr "\n" ";")" >> $O/summary.txt
  SEED=$((SEED + 7))
done
timeout -k 10 400 python3 tools/pipeline_determinism.py 400 C3 4 hard > $O/pipeline_determinism.log 2>&1 || { echo "pipeline_determinism FAILED" >> $O/summary.txt; exit 1; }
echo "pipeline_determinism: $(tail -1 $O/pipeline_determinism.log)" >> $O/summary.txt
timeout -k 10 400 python3 tools/grid_determinism.py 600 64 > $O/grid_determinism.log 2>&1 || { echo "grid_determinism FAILED" >> $O/summary.txt; exit 1; }
echo "grid_determinism: $(tail -1 $O/grid_determinism.log)" >> $O/summary.txt
cat $O/summary.txt
