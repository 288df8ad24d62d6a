#!/bin/bash
# A/B: tools/ab_env.sh "<bench args>" "name[,ENV=VAL,...]" ...   (name "base" = in-tree library, else build_variants/lib_<name>.so)
args=$1; shift
for v in "$@"; do
  IFS=, read -ra parts <<< "$v"
  name=${parts[0]}
  ( if [ $name != base ]; then export CHROMA_HIP_LIBRARY=$PWD/build_variants/lib_$name.so; fi
    for kv in "${parts[@]:1}"; do export "$kv"; done
    echo "== $v: $(python bench.py $args --no-cpu-baseline 2>&1 >/dev/null | grep 'timed' )" )
done
