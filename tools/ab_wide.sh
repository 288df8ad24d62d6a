#!/bin/bash
# A/B variants of the wide ray cast: tools/ab_wide.sh "<bench args>" name:waves_per_cu ...
# ("base" = the in-tree library; other names are build_variants/lib_<name>.so)
args=$1; shift
for v in "$@"; do
  name=${v%%:*}; waves=${v#*:}
  if [ $name = base ]; then unset CHROMA_HIP_LIBRARY; else export CHROMA_HIP_LIBRARY=$PWD/build_variants/lib_$name.so; fi
  export CHROMA_WIDE_WAVES_PER_CU=$waves
  echo "== $v: $(python bench.py $args --no-cpu-baseline 2>&1 >/dev/null | grep 'timed' )"
done
