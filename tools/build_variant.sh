#!/bin/bash
# Build a variant of the engine for an A/B run: tools/build_variant.sh NAME -DFOO=1 ...  ->  build_variants/lib_NAME.so
# (host objects are shared with the in-tree build; tools/ab_env.sh selects the variant through CHROMA_HIP_LIBRARY)
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $here/build_variants
cd $here/chroma_amd/csrc
make -s all >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-value -Wno-unused-result \
  "$@" -c chroma_hip.hip -o $here/build_variants/chroma_hip_$name.o
objs=$(ls *.o | grep -v '^chroma_hip\.o$' | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $here/build_variants/lib_$name.so $here/build_variants/chroma_hip_$name.o $objs -pthread
rm -f $here/build_variants/chroma_hip_$name.o
echo built build_variants/lib_$name.so
