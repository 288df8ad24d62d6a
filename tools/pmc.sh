#!/bin/bash
# PMC passes (one counter group per run, --pmc alone: no trace flags) for the bench command given as arguments.
# usage: tools/pmc.sh OUTDIR python bench.py --config lite --steps 1 --warmup 0 --no-cpu-baseline
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for grp in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
  "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  if [ -n "${PMC_GROUPS:-}" ] && ! echo " $PMC_GROUPS " | grep -q " $i "; then continue; fi
  timeout -k 5 ${PMC_TIMEOUT:-120} rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- "$@" > $out/pass$i.stdout 2> $out/pass$i.stderr
  echo "pass $i rc=$? : $grp"
done
python tools/pmc_summary.py $out
# (PMC_GROUPS="1 2" restricts the passes: see the loop above)
