#!/bin/bash
# Counter passes aimed at k_physics (what keeps it at 2.9 TB/s with the VALUs 41 % busy): memory-instruction mix, TA / TCP
# LDS, waits.  One group per run, --pmc alone.  usage: tools/pmc_physics.sh OUTDIR
# (Groups of derived TA_* / TCP_* / TCC_*_sum counters were tried on the 1e8-photon command: each pass ran into its 200 s
#  limit -- they replay every dispatch many times -- so they are not in here.)
set -u
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for grp in \
  "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVES" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" ; do
  i=$((i+1))
  timeout -k 5 ${PMC_TIMEOUT:-200} rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pass$i.stdout 2> $out/pass$i.stderr
  echo "pass $i rc=$? : $grp"
done
python tools/pmc_summary.py $out | grep -A60 "k_physics<false>" | sed -n 1,60p
