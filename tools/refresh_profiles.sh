#!/bin/bash
# One GPU-box pass that refreshes the judged artefacts: GPU tests, the bench lines of every config and the
# rocprofv3 kernel summary of the default bench command.  usage: tools/refresh_profiles.sh OUTDIR
set -u
out=$1
mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -20 $out/gpu_tests.log; exit 1; }
tail -1 $out/gpu_tests.log
python bench.py > $out/bench_c3_default.json 2> $out/bench_c3_default.log || exit 1
for cfg in tiny lite c5 detector; do
  python bench.py --config $cfg > $out/bench_$cfg.json 2> $out/bench_$cfg.log || exit 1
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_c3 -- python3 bench.py --no-cpu-baseline > $out/rocprof_c3.json 2> $out/rocprof_c3.log || exit 1
python tools/prof_summary.py $out/rocprof_c3 $out/rocprof_c3_default_summary.txt bench.py
cat $out/bench_*.json
