#!/bin/bash
# One PMC pass with the SQ issue / wait counters over a bench workload (per kernel: instructions by kind, wave cycles, waits).
# usage: bash tools/pmc_sq.sh <tag> [bench.py workload arguments]
TAG=${1:-sq}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/sq -o bench -- python $R/bench.py --skip-extras --graded-probe-only --steps 2 --warmup 0 --probe-repeat 2 "$@" > /dev/null 2> $O/sq.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/sq2 -o bench -- python $R/bench.py --skip-extras --graded-probe-only --steps 2 --warmup 0 --probe-repeat 2 "$@" > /dev/null 2> $O/sq2.log
python - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for d in ("sq", "sq2"):
    for path in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        seen = set()
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (d, row["Dispatch_Id"])
            if d == "sq" and key not in seen:
                seen.add(key); n[k] += 1
                if "Start_Timestamp" in row: dur[k] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6
with open("$R/gpurun_out/${TAG}_sq.txt", "w") as f:
    f.write("kernel launches ms_total waves valu/wave lds/wave salu/wave vmrd/wave vmwr/wave wavecyc/wave(quad) active_valu wait_any wait_inst_any wait_inst_lds lds_conflict\n")
    for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"])[:40]:
        a = acc[k]; w = max(1.0, a["SQ_WAVES"]); wc = max(1.0, a["SQ_WAVE_CYCLES"])
        f.write("%-40s %4d %8.2f %9d %8.0f %7.0f %7.0f %6.0f %6.0f %9.0f %.3f %.3f %.3f %.3f %.3f\n" % (k[:40], n[k], dur[k], w, a["SQ_INSTS_VALU"] / w, a["SQ_INSTS_LDS"] / w,
                a["SQ_INSTS_SALU"] / w, a["SQ_INSTS_VMEM_RD"] / w, a["SQ_INSTS_VMEM_WR"] / w, wc / w, a["SQ_ACTIVE_INST_VALU"] / wc, a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc,
                a["SQ_WAIT_INST_LDS"] / wc, a["SQ_LDS_BANK_CONFLICT"] / max(1.0, a["SQ_LDS_IDX_ACTIVE"])))
print(open("$R/gpurun_out/${TAG}_sq.txt").read())
PY
find $O -name "*.csv" -size +5M -delete
