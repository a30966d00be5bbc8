#!/bin/bash
# Calibration of the rocprofv3 TCC read counters on a workload with a KNOWN request count:
# tools/gather_probe's divergent per-lane 16-B gathers over tables that sit in L2, in the Infinity Cache and in HBM.
# Each --pmc group is its own run (with --kernel-trace only).  Output: gpurun_out/probe/*.{txt,csv}; the summary that is
# judged is copied to profiles/ by hand.
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/probe
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
$root/tools/gather_probe > $out/probe.txt 2> $out/probe.err || { echo "probe failed"; cat $out/probe.err; exit 1; }
g=0
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  g=$((g+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$g -- $root/tools/gather_probe "gather loads" > $out/g$g.txt 2> $out/g$g.err || echo "group $g failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)   # (dispatch id) -> {counter: value}
names = {}
for f in sorted(glob.glob(out + '/g*/*/*_counter_collection.csv')):
    grp = f.split('/')[-3]
    with open(f) as fh:
        for row in csv.DictReader(fh):
            d = int(row['Dispatch_Id'])
            rows[d][row['Counter_Name']] = rows[d].get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
            names[d] = row['Kernel_Name'].split('(')[0].replace('void ', '')
cases = [l.split('  ')[0].strip() for l in open(out + '/probe.txt') if l.startswith('gather')]
# every case is launched twice (warm-up, timed) in the order probe.txt lists them
print('# dispatch order = probe.txt order, two launches per case; counters of the SECOND launch of each case')
for i, c in enumerate(cases):
    d = 2 * i + 2
    if d in rows:
        print(c, {k: round(v, 1) for k, v in sorted(rows[d].items())})
PY
