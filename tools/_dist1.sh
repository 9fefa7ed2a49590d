# one-rank RCCL rehearsal of the N > 1 path at C3, both communicator arrangements + the plain run: tools/_dist1.sh
R=$GRAFT_REPO_ROOT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="--gpus 1 --steps 40 --warmup 8 --no-extras --no-profile-pass --cpu-pairs 0 --cpu-all-cores-pairs 0"
for rep in 1 2; do
for comm in per-rank per-context; do
  VSLAM_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29500 + RANDOM % 400)) timeout -k 10 300 python3 bench.py $B --comm $comm 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$comm', round(d['ms_per_step'],4), d['record_gather']['communicators_per_rank'], d['parity_in_bench']['bit_exact'])"
done
timeout -k 10 300 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', round(d['ms_per_step'],4))"
done
