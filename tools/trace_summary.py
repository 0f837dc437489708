#!/usr/bin/env python3
"""Summary of a one-lane kernel timeline (tools/gpu_trace.sh: rocprofv3 --kernel-trace of bench.py --lanes 1): the last complete
step's period, the time a kernel is running in it, the idle gaps, the kernels by time.  usage: trace_summary.py <dir of gpu_trace.sh>"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main(d):
    rows = list(csv.DictReader(open(glob.glob(d + "/trace/*kernel_trace.csv")[0])))
    j = json.loads(open(d + "/bench.json").read().strip().splitlines()[-1])
    print("bench line under the profiler: %.1f M pairs/s, %.2f ms per step, one lane" % (j["value"], j["ms_per_step"]))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows)
    s3a = [i for i, e in enumerate(ev) if e[2].startswith("k_s3a_count")]

    def first_of_step(i):
        while i > 0 and not ev[i][2].startswith("void k_prep_mm"):
            i -= 1
        while i > 0 and ev[i - 1][2].startswith(("void k_prep_mm", "void k_probe_range", "k_copy_u64", "void k_pack_reads", "__amd_rocclr")):
            i -= 1
        return i
    a, b = first_of_step(s3a[-2]), first_of_step(s3a[-1])
    seg = ev[a:b]
    t0, t1 = ev[a][0], ev[b][0]
    iv = sorted((e[0], e[1]) for e in seg)
    busy, (cs, ce), gaps = 0, iv[0], []
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append(s - ce)
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print("last complete step: period %.2f ms, a kernel running %.2f ms, idle between kernels %.2f ms (largest gap %.3f), idle before the next step %.2f ms, %d launches"
          % ((t1 - t0) / 1e6, busy / 1e6, sum(gaps) / 1e6, max(gaps or [0]) / 1e6, (t1 - ce) / 1e6, len(seg)))
    tot, cnt = defaultdict(float), defaultdict(int)
    for e in seg:
        tot[e[2]] += (e[1] - e[0]) / 1e6
        cnt[e[2]] += 1
    for k, v in sorted(tot.items(), key=lambda x: -x[1])[:18]:
        print("  %7.3f ms  x%-2d %s" % (v, cnt[k], k))


if __name__ == "__main__":
    main(sys.argv[1])
