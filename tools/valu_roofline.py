#!/usr/bin/env python3
"""Which roofline does the path actually run into?  From the SQ counter pass and the kernel trace of ONE profiled round (both written
by tools/profile_round.sh on an MI355X): per kernel the vector instructions issued (SQ_INSTS_VALU, wave instructions summed over the
chip), its duration, and the share of the chip's vector-issue capacity that is -- 256 CUs x 4 SIMDs, one wave instruction per SIMD
every 4 cycles at 2.4 GHz (cdna_hip_programming.md; the microbenchmark of profiles/r02_a_valu_rate_*.txt prices single instructions
between 2.3 and 4.7 cycles, so a kernel of mostly full-rate instructions can exceed 1.0 on this scale).
    python tools/valu_roofline.py profiles/r04_pmc_sq_clean_pass256.csv profiles/r04_clean_kernel_stats.csv 256 > profiles/r05_d_valu_roofline.md"""
import csv
import sys

SIMDS, GHZ, CYCLES_PER_INST = 256 * 4, 2.4, 4.0


def main(pmc_csv, stats_csv, images):
    images = int(images)
    with open(pmc_csv) as f:
        pmc = list(csv.DictReader(f))
    with open(stats_csv) as f:
        dur = {r["kernel"]: float(r["avg_us"]) for r in csv.DictReader(f)}          # average duration of one launch (the trace ran three passes)
    launches = {r["kernel"]: float(r["dispatches"]) for r in pmc}                   # launches per pass: the PMC run was ONE pass
    rows = [(r["kernel"], float(r["SQ_INSTS_VALU_per_dispatch"]) * launches[r["kernel"]], dur[r["kernel"]])
            for r in pmc if r["kernel"].startswith("k_") and r["kernel"] in dur]
    print("# Vector-issue roofline of the hot path (round-4 counters re-read in round 5; no new measurement)\n")
    print("Source: `%s` (one pass of %d clean diagrams) and `%s` (kernel durations, same build).  Capacity: %d SIMDs x %.1f GHz / %.0f cycles per"
          " wave instruction = %.0f G wave instructions/s.\n" % (pmc_csv, images, stats_csv, SIMDS, GHZ, CYCLES_PER_INST, SIMDS * GHZ / CYCLES_PER_INST))
    print("| kernel | launches per pass | us per pass | us per diagram | VALU wave instructions per diagram | share of the vector-issue capacity |")
    print("|---|---|---|---|---|---|")
    tot_v = tot_t = 0.0
    for k, v, d in sorted(rows, key=lambda r: -r[2] * launches[r[0]]):
        t = d * launches[k]
        frac = v * CYCLES_PER_INST / (SIMDS * GHZ * 1e3 * t)
        tot_v += v
        tot_t += t
        print("| `%s` | %d | %.1f | %.2f | %.0f | %.2f |" % (k, launches[k], t, t / images, v / images, frac))
    print("| **whole path (kernels above)** | | %.1f | **%.2f** | %.0f | **%.2f** |" % (tot_t, tot_t / images, tot_v / images,
                                                                                      tot_v * CYCLES_PER_INST / (SIMDS * GHZ * 1e3 * tot_t)))
    print("\nReading: over the whole path the chip issues vector instructions on %.0f %% of the SIMD-cycles it has while a kernel runs -- "
          "the bound this path runs into is INSTRUCTION ISSUE, not HBM (the stage's measured traffic is 0.81 x its algorithmic bytes at "
          "0.65 of the HBM roofline, and the three largest kernels hardly touch HBM at all).  A faster path needs fewer vector instructions per "
          "diagram; `k_radius` (latency) and `k_blur` (its stores) are the two kernels with headroom of another kind." % (
              100.0 * tot_v * CYCLES_PER_INST / (SIMDS * GHZ * 1e3 * tot_t)))


if __name__ == "__main__":
    main(*sys.argv[1:4])
