#!/usr/bin/env python3
"""Summarises tools/roofs.sh: per kernel, the measured rates against the chip's ceilings.

Ceilings (MI355X, measured by tools/gather_probe on this pool: profiles/r02_gather_probe.txt):
  VALU issue          0.415 wave-instructions / clock / SIMD (independent v_fma_f32, 8 waves per SIMD, nominal 2.4 GHz)
  texture addresser   0.88 lane-loads (16 B each) / clock / CU for fully divergent gathers, 4.0 when a wave's lanes share a line
  fabric reads        54-61 G 128-B lines / s for random lines beyond the L2s (6.9-7.8 TB/s); HBM spec 8 TB/s
Bytes on the fabric side: TCC_EA0_RDREQ x 128 B (calibrated: one request per distinct 128-B line for 16-B, 64-B and 112-B
touches alike, profiles/r02_gather_probe_counters.txt), writes: 64-B requests x 64 + the others x 32.
"""
import collections
import csv
import glob
import sys

CLK = 2.4e9
CUS, SIMDS = 256, 1024


def short(name):
    k = name.split("(")[0].replace("void ", "").replace("trtd::", "")
    base = k.split("<")[0]
    counting = "<true" in k.replace(" ", "")
    return base + ("<count>" if counting else "")


def main():
    out, title = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(float)
    dur = collections.defaultdict(float)
    calls = collections.Counter()
    for f in glob.glob(out + "/g*/*/*_counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[(short(row["Kernel_Name"]), row["Counter_Name"])] += float(row["Counter_Value"])
    for f in glob.glob(out + "/g1/*/*_kernel_trace.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e9
                calls[k] += 1
    print("#", title)
    print("# counters: sums over all launches of the kernel in one bench run (1 counting render + warm-up + 1 timed step); <count> = the counting build")
    for k in sorted(dur, key=lambda k: -dur[k]):
        if "rocclr" in k or dur[k] < 1e-4 or "<count>" in k:
            continue
        g = lambda c: agg.get((k, c), 0.0)
        t = dur[k]
        cyc = t * CLK
        print(f"{k:22s} {t * 1e3:9.2f} ms over {calls[k]:4d} launches")
        if g("SQ_INSTS_VALU"):
            print(f"    VALU issue        {g('SQ_INSTS_VALU') / (cyc * SIMDS):6.3f} wave-instr/clk/SIMD   ({g('SQ_INSTS_VALU') / (cyc * SIMDS) / 0.415 * 100:5.1f} % of the 0.415 FMA ceiling)"
                  f"   SALU {g('SQ_INSTS_SALU') / (cyc * CUS):5.2f} instr/clk/CU   SALU:VALU {g('SQ_INSTS_SALU') / max(g('SQ_INSTS_VALU'), 1):4.2f}")
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            print(f"    lanes active per VALU instruction   {g('SQ_THREAD_CYCLES_VALU') / (g('SQ_ACTIVE_INST_VALU') * 64):5.2f}")
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            print(f"    wave time         waiting on memory/barrier {g('SQ_WAIT_ANY') / wc * 100:5.1f} %   issue-stalled {g('SQ_WAIT_INST_ANY') / wc * 100:5.1f} %   resident waves/SIMD {wc * 4 / (cyc * SIMDS):4.1f}")
        if g("TCP_TOTAL_ACCESSES_sum"):
            r = g("TCP_TOTAL_ACCESSES_sum") / (cyc * CUS)
            print(f"    texture addresser {r:6.3f} accesses/clk/CU   (divergent ceiling 0.88, coherent 4.0)   L1->L2 read requests {g('TCP_TCC_READ_REQ_sum') / t / 1e9:7.1f} G/s")
        if g("TCC_REQ_sum"):
            hit = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1)
            lines = g("TCC_EA0_RDREQ_sum")
            print(f"    L2                hit rate {hit * 100:5.1f} %   fabric reads {lines / t / 1e9:6.2f} G lines/s = {lines * 128 / t / 1e12:5.2f} TB/s"
                  f"   ({lines / t / 1e9 / 54 * 100:5.1f} % of the 54 G lines/s random-line ceiling, {lines * 128 / t / 8e12 * 100:5.1f} % of 8 TB/s)")
        if g("TCC_EA0_WRREQ_sum"):
            w64 = g("TCC_EA0_WRREQ_64B_sum")
            wb = w64 * 64 + (g("TCC_EA0_WRREQ_sum") - w64) * 32
            print(f"    fabric writes     {wb / t / 1e12:5.2f} TB/s")


if __name__ == "__main__":
    main()
