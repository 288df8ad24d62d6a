#!/bin/bash
# A/B kernel variants: tools/ab.sh "<bench args>" variant...   (variant "base" = in-tree library)
args=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset CHROMA_HIP_LIBRARY; else export CHROMA_HIP_LIBRARY=$PWD/build_variants/lib_$v.so; fi
  echo "== $v: $(python bench.py $args --no-cpu-baseline 2>&1 >/dev/null | grep 'timed' )"
done
