# Step time of the bench's own timed loop, one FRESH PROCESS per configuration (so that every configuration makes its streams
# in the same order: which hardware queue a stream lands on depends on what the process created before it, and that mapping
# moves the step by up to 9 % -- tools/ab_env.py's in-process comparison is blind to it), repeated and interleaved.
#   bash tools/ab_proc.sh <reps> "NAME=VALUE ..." "NAME=VALUE ..." ...      ("-" = no variables)
# environment knobs exist in the EXPERIMENTS build of the library only (vslam_amd/build.py: libvslam_amd_exp.so)
export VSLAM_AMD_LIB=${VSLAM_AMD_LIB:-$(dirname $0)/../vslam_amd/libvslam_amd_exp.so}
REPS=$1; shift
B="--no-extras --cpu-pairs 0 --no-profile-pass --steps 60 --warmup 10"
for r in $(seq $REPS); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    ms=$(env $e python3 bench.py $B $BENCH_ARGS 2>/dev/null | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$ms  [$cfg]"
  done
done | sort -k2 | awk '{k=$0; sub(/^[^ ]+  /,"",k); v[k]=v[k]" "$1} END {for (k in v) print k, v[k]}'
