#!/bin/bash
# kernel + copy timeline of one CLI run from BGZF files to BED (files made by tools/e2e_bench.py in the same call)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05_cli_trace}
mkdir -p $O
D=/tmp/chromap_amd_e2e
[ -f $D/r1.fq.bgz ] || timeout 600 python tools/e2e_bench.py --gz --reps 1 > $O/e2e.json 2> $O/e2e.log
export TMPDIR=/tmp; cd /tmp
rm -f $D/out.bed; sync
# the CLI's own per-batch times, without the profiler (twice: the second run has the files in the page cache for sure)
for i in 1 2; do CM_CLI_TIMES=1 $GRAFT_REPO_ROOT/chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.bgz -2 $D/r2.fq.bgz -o $D/out.bed 2>&1 | grep -v "^Mapped [0-9]* read" > $O/times$i.log; done
cat $O/times2.log
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o t -- $GRAFT_REPO_ROOT/chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.bgz -2 $D/r2.fq.bgz -o $D/out.bed 2> $O/cli.log
tail -3 $O/cli.log
python - <<PY
import csv, glob
ev = []
for f in glob.glob('$O/trace/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48] + "  s" + r.get("Stream_Id", "?") + " q" + r.get("Queue_Id", "?")))
for f in glob.glob('$O/trace/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + str(r.get("Bytes", r.get("Size", "")))))
ev.sort()
# the mapping part: from the first k_bgzf_tokens on
i0 = next((i for i, e in enumerate(ev) if "bgzf" in e[2]), 0)
t0 = ev[i0][0]
with open('$O/timeline.txt', 'w') as f:
    for s, e, n in ev[i0:]:
        if e - s > 200000: f.write("%9.3f ms  %8.3f ms  %s\n" % ((s - t0) / 1e6, (e - s) / 1e6, n))
print(open('$O/timeline.txt').read()[:6000])
PY
find $O/trace -name "*.csv" -size +3M -delete
