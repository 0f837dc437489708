#!/usr/bin/env python3
"""Where k_s3b_coop spends its cycles (measurement aid): shader-clock cycles of lane 0 of every group between the phases of
cm_coop_s3b, summed over the groups of a few batches of the repeat-bearing workload.  python tools/coop_profile.py [repeats]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    torch.zeros(1, device="cuda")
    from chromap_amd import ChromapGPU, Stats
    rep = sys.argv[1] if len(sys.argv) > 1 else "32,600,3000,0.02"
    if not rep.startswith("profile:"):
        f = rep.split(",")
        rep = (int(f[0]), int(f[1]), int(f[2]), float(f[3]))
    g = ChromapGPU(synthetic=(3_100_000_000, 24, 12345, rep), preset="atac")
    g.set_option("lanes", 1)
    if len(sys.argv) > 2:
        g.set_option("coop", int(sys.argv[2], 0))
    g.generate_resident(4_000_000, read_length=50, frag_min=30, frag_max=600, sub_rate=0.01, seed=3000)
    g.map_resident(Stats())
    g.set_option("coop_profile", 1)
    for _ in range(3):
        g.map_resident(Stats())
    v = [g.get_option("coop_profile_%d" % k) for k in range(64)]
    names = ["run table + scan (header loads)", "expand pass 1 (occurrence loads)", "pass 2 (strand partition)", "scan of the pieces' + counts", "merge levels", "cluster sweep"]
    tot = float(sum(v[:6])) or 1.0
    n = max(1, v[8])
    print("groups %d, mean hits %.0f, mean runs %.1f, mean cycles per group %.0f" % (n, v[10] / n, v[9] / n, tot / n))
    for k, nm in enumerate(names):
        print("  %-28s %5.1f %%  %8.0f cycles per group" % (nm, 100.0 * v[k] / tot, v[k] / n))
    print("  inside the sweep: walks %.0f, barrier after them %.0f, chunk sums + scans %.0f, copy-out %.0f cycles per group"
          % (v[11] / n, v[12] / n, v[13] / n, v[14] / n))
    nr = max(1, v[16])
    print("heavy rescue searches (a group of 16 lanes per read and direction): %d searched + %d bailed out; per search: best mate candidates "
          "(windows before merging) %.1f, mate candidates %.1f, minimizers %.1f, occurrences of its minimizers %.0f, hits %.1f"
          % (v[16], v[26], v[17] / nr, v[18] / nr, v[20] / nr, v[19] / nr, v[21] / nr))
    print("  k_s4a_rescue_list, cycles of a wave in its three loops (sum over waves / longest wave): wave-per-read %d / %d (%d reads), lane-per-read %d / %d, "
          "16-lanes-per-read %d / %d" % (v[27], v[28], v[11], v[29], v[30], v[31], v[15]))
    print("  k_s4b_rescue_list likewise: wave-per-read %d / %d, lane-per-read %d / %d, 16-lanes-per-read %d / %d" % (v[32], v[33], v[34], v[35], v[36], v[37]))
    print("  wave kernels: longest single read in k_s4a_rescue_wave %d cycles; waves' cycles sum / longest wave: s4a %d / %d, s4b %d / %d" % (v[22], v[27], v[28], v[32], v[33]))
    print("  wave kernel (a wave per read), larger of the two searches' best mate candidates: < 16: %d, < 32: %d, < 64: %d, < 128: %d, < 200: %d, < 300: %d, more: %d" % tuple(v[40:47]))
    ns = max(1, v[55])
    print("cm_coop_rescue (sampled waves: %d searches, %.1f rounds each, %.0f occurrences in the windows per search), cycles per search: best + windows %.0f, "
          "minimizer tables %.0f, A (bounds) %.0f, eq + B (chain) %.0f, C (scan + emit) %.0f, pool emit %.0f, tail %.0f"
          % (v[55], v[57] / ns, v[56] / ns, v[48] / ns, v[49] / ns, v[50] / ns, v[51] / ns, v[52] / ns, v[53] / ns, v[54] / ns))
    nd = max(1, v[63])
    print("cm_coop_rescue_dir (sampled blocks of k_s4b_coop: %d directions), cycles per direction: hits to shared memory %.0f, runs + merge sort %.0f, "
          "sweep %.0f, own candidates staged + merge into Z %.0f, greedy walk + compaction %.0f" % (v[63], v[58] / nd, v[59] / nd, v[60] / nd, v[61] / nd, v[62] / nd))
    print("k_s5_sort_coop: %d reads, %.0f cycles of a wave per read, longest wave %d cycles; lists left to lane 0: %d not in position order, %d other"
          % (v[6], v[38] / max(1, v[6]), v[39], v[7], v[47]))
    print("timings of the last batch:", g.timings())


if __name__ == "__main__":
    main()
