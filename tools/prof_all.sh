set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-pairs 0 --cpu-all-cores-pairs 0 --no-profile-pass > $O/stats.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --cpu-all-cores-pairs 0 --no-profile-pass --pmc-calibrate > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --cpu-all-cores-pairs 0 --no-profile-pass --pmc-calibrate > $O/write.log 2>&1
echo write done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/sq -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --cpu-all-cores-pairs 0 --no-profile-pass > $O/sq.log 2>&1
echo sq done
cd $R
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
echo bench done
