#!/bin/bash
# PMC passes over one bench.py workload (each --pmc group in its own run, with --kernel-trace only).
# usage: tools/pmc.sh <tag> "<bench args>" "<counters group 1>" "<counters group 2>" ...
tag=$1; args=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
g=0
for grp in "$@"; do
  g=$((g+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$g -- python3 $root/bench.py $args --steps 1 --warmup 1 --no-cpu-baseline --no-traffic > $out/g$g.json 2> $out/g$g.err || echo "group $g failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(float); dur = collections.defaultdict(float); calls = collections.Counter()
for f in glob.glob(out + '/g*/*/*_counter_collection.csv'):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('trtd::', '')
            k = k.split('<')[0] + ('<count>' if '<true' in row['Kernel_Name'] else '')
            agg[(k, row['Counter_Name'])] += float(row['Counter_Value'])
for f in glob.glob(out + '/g1/*/*_kernel_trace.csv'):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('trtd::', '')
            k = k.split('<')[0] + ('<count>' if '<true' in row['Kernel_Name'] else '')
            dur[k] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6; calls[k] += 1
kern = sorted(dur, key=lambda k: -dur[k])
for k in kern:
    if 'rocclr' in k: continue
    print('%-26s %9.2f ms %5d calls' % (k, dur[k], calls[k]))
    for (kk, c), v in sorted(agg.items()):
        if kk == k: print('      %-40s %.6g' % (c, v))
PY
