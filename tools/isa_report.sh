#!/bin/bash
# Registers, scratch, LDS and occupancy of every kernel as compiled for gfx950 (no GPU needed):
# tools/isa_report.sh > profiles/rNN/isa_resources.txt
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-value -Wno-unused-result \
  -S --cuda-device-only -o $tmp/k.s $here/chroma_amd/csrc/chroma_hip.hip 2>/dev/null
echo "# kernel, VGPRs, SGPRs, scratch bytes, LDS bytes, waves/SIMD, code bytes   (hipcc -O3, gfx950; flags of chroma_amd/csrc/Makefile)"
awk '
/^[_A-Za-z0-9]+:.*; @/ { name=$1; sub(":","",name) }
/; codeLenInByte/ { code=$4 }
/; TotalNumSgprs:/ { s=$3 }
/; NumVgprs:/ { v=$3 }
/; ScratchSize:/ { sc=$3 }
/; LDSByteSize:/ { l=$3 }
/; Occupancy:/ { printf "%s, %s, %s, %s, %s, %s, %s\n", name, v, s, sc, l, $3, code }
' $tmp/k.s | while IFS= read -r line; do n=${line%%,*}; d=$(echo "$n" | c++filt | sed 's/(.*//'); echo "$d,${line#*,}"; done | sort
rm -rf $tmp
