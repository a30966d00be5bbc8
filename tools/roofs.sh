#!/bin/bash
# Counter evidence for "which roof binds this kernel": per path kernel of one bench.py workload, the instruction-issue,
# texture-address, L2 and fabric (Infinity Cache + HBM) rates, to be read against the ceilings tools/gather_probe measured
# on the same chip (profiles/r02_gather_probe.txt).  Every --pmc group is its own run with --kernel-trace only.
# usage: tools/roofs.sh <tag> "<bench args>"       -> gpurun_out/roofs_<tag>/summary.txt
tag=$1; args=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/roofs_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
g=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  g=$((g+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$g -- python3 $root/bench.py $args --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-overlap-extra --no-traffic > $out/g$g.json 2> $out/g$g.err || echo "group $g failed"
done
python3 $root/tools/roofs_summary.py $out "$tag: bench.py $args --steps 1 --warmup 1" > $out/summary.txt
cat $out/summary.txt
