# A/B of the matcher at C3 and C5: kernel time (alone, its own launches, spreading pre-pass included) and step time for
# the FP4 form (default: 8 waves x 32 rows, pre-spread train rows) and its variants, and the int8 form.
# usage (on the GPU box): bash tools/ab_match.sh > gpurun_out/ab_match.txt
# environment knobs exist in the EXPERIMENTS build of the library only (vslam_amd/build.py: libvslam_amd_exp.so)
export VSLAM_AMD_LIB=${VSLAM_AMD_LIB:-$(dirname $0)/../vslam_amd/libvslam_amd_exp.so}
run() {   # label, then VAR=value ...
  label=$1; shift
  for wl in C3 C5; do
    env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --cpu-pairs 4 --cpu-all-cores-pairs 0 --no-extras \
      2>/dev/null > /tmp/ab_match.json
    python - "$wl" "$label" <<'P'
import json, sys
d = json.load(open("/tmp/ab_match.json"))
k = {x["kernel"]: x["ms_per_launch"] for x in d["kernels"]}
print("%s %-28s step %.3f ms  match %.4f ms  kdtree %.4f  parity %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"],
      k["match_knn2_kernel"], k["kdtree_build_kernel"], d["parity_in_bench"]["bit_exact"]))
P
  done
}
run "fp4 8x32 (default)" VSLAM_MATCH_FORM=fp4
run "fp4 8x32 2 tiles/trip" VSLAM_MATCH_FORM=fp4 VSLAM_MATCH_TILES_PER_TRIP=2
run "fp4 4x64" VSLAM_MATCH_FORM=fp4 VSLAM_MATCH_SHAPE=4x64
run "fp4 4x64 2 tiles/trip" VSLAM_MATCH_FORM=fp4 VSLAM_MATCH_SHAPE=4x64 VSLAM_MATCH_TILES_PER_TRIP=2
run "fp4 8x32 no pre-spread" VSLAM_MATCH_FORM=fp4 VSLAM_MATCH_NO_PRESPREAD=1
run "fp4 4x64 no pre-spread" VSLAM_MATCH_FORM=fp4 VSLAM_MATCH_SHAPE=4x64 VSLAM_MATCH_NO_PRESPREAD=1
run "int8 8x32" VSLAM_MATCH_FORM=i8 VSLAM_MATCH_SHAPE=8x32
run "int8 4x64" VSLAM_MATCH_FORM=i8 VSLAM_MATCH_SHAPE=4x64
