# rocprofv3 passes over the bench command: kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes, with the
# calibration copies), SQ counters.  Usage (on the GPU box): bash tools/prof_all.sh <tag> [C3|C5|C2]
# The counter passes and the per-kernel stats run ONE batch at a time on one context (--in-flight 1): rocprofv3 serialises
# dispatches under --pmc anyway, and a kernel's duration is its own only when no other batch shares the chip -- which is
# also how bench.py's HIP-event pass measures it.  The stats pass adds --alone (round 6): the k-d build, the blur and the
# generator stay on the main stream, so nothing runs beside a traced kernel and its duration agrees with the HIP-event pass.
# A second stats pass takes the default command (four batches in flight).  The SQ pass carries GRBM_GUI_ACTIVE (a launch's
# cycles, clock-free) and the calibration kernel (--pmc-calibrate): occupancy and a calibrated vector-pipe share come out.
set -e
TAG=${1:-r05}
WL=${2:-C3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
[ "$WL" != "C3" ] && O=$R/gpurun_out/${TAG}_$WL
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--workload $WL --cpu-pairs 0 --cpu-all-cores-pairs 0 --no-profile-pass --no-extras"
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py $B --in-flight 1 --alone --steps 5 --warmup 1 > $O/stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --stats -d $O/stats3 -o s --output-format csv -- python3 $R/bench.py $B --steps 9 --warmup 3 > $O/stats3.log 2>&1
echo stats3 done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o p --output-format csv -- python3 $R/bench.py $B --in-flight 1 --steps 2 --warmup 1 --pmc-calibrate > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o p --output-format csv -- python3 $R/bench.py $B --in-flight 1 --steps 2 --warmup 1 --pmc-calibrate > $O/write.log 2>&1
echo write done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/sq -o p --output-format csv -- python3 $R/bench.py $B --in-flight 1 --steps 2 --warmup 1 --pmc-calibrate > $O/sq.log 2>&1
echo sq done
cd $R
python3 tools/prof_summary.py $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_summary.csv > /dev/null
python3 tools/prof_summary.py $(find $O/stats3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_in_flight_summary.csv > /dev/null
python3 tools/pmc_summary.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_hbm_traffic.csv > /dev/null
python3 tools/sq_summary.py $(find $O/sq -name "*counter_collection.csv" | head -1) $O/sq_counters.csv
python3 -c "import sys, json; sys.path.insert(0, '.'); import bench; print(json.dumps(bench.source_stamp(), indent=1))" > $O/source_stamp.txt
# a timeline of one step, one batch at a time (the kernels of one batch in order, both queues)
python3 tools/timeline.py $(find $O/stats -name "*kernel_trace.csv" | head -1) > $O/step_timeline.txt 2>/dev/null || true
rm -rf $O/stats $O/stats3 $O/fetch $O/write $O/sq   # the raw traces are large; the summaries are what is kept
