# SQ counters of the grid extractor's kernels: tools/_grid_pmc.sh <tag>
set -e
TAG=${1:-r06_grid_pmc}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/sq -o p --output-format csv -- python3 $R/tools/grid_bench.py --frames 64 --steps 2 --no-prof > $O/sq.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace -d $O/sq2 -o p --output-format csv -- python3 $R/tools/grid_bench.py --frames 64 --steps 2 --no-prof > $O/sq2.log 2>&1
cd $R
python3 tools/sq_summary.py $(find $O/sq -name "*counter_collection.csv" | head -1) $O/sq_counters.csv
python3 - <<PY
import csv, collections, sys, glob
sys.path.insert(0, "$R")
from vslam_amd.profnames import kernel_id
f = glob.glob("$O/sq2/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = kernel_id(r["Kernel_Name"])
    if k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$O/sq_lds.csv", "w") as o:
    o.write("kernel,lds_insts_per_wave,salu_per_wave,vmem_rd_per_wave,lds_active_x4_per_wave,lds_wait_x4_per_wave,bank_conflict_cycles_per_wave,lds_idx_active_per_wave\n")
    for k, v in agg.items():
        w = v["SQ_WAVES"] or 1
        o.write(f"{k},{v['SQ_INSTS_LDS']/w:.0f},{v['SQ_INSTS_SALU']/w:.0f},{v['SQ_INSTS_VMEM_RD']/w:.0f},{4*v['SQ_ACTIVE_INST_LDS']/w:.0f},{4*v['SQ_WAIT_INST_LDS']/w:.0f},{v['SQ_LDS_BANK_CONFLICT']/w:.0f},{v['SQ_LDS_IDX_ACTIVE']/w:.0f}\n")
print(open("$O/sq_lds.csv").read())
PY
rm -rf $O/sq $O/sq2
