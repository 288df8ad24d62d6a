#!/bin/bash
# The host-side C++ (chroma_amd/csrc/{bvh_build,wide_build,mesh_utils}.cpp) under AddressSanitizer and UBSan
# (CPU build only: the pool has no GPU sanitizer), then the CPU oracle the same way.  usage: tools/asan_host.sh
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -shared -pthread -ffp-contract=off \
  -I$here/include $here/chroma_amd/csrc/bvh_build.cpp $here/chroma_amd/csrc/wide_build.cpp $here/chroma_amd/csrc/mesh_utils.cpp \
  -o $tmp/libhost_asan.so
CHROMA_ASAN_LIB=$tmp/libhost_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  ASAN_OPTIONS=detect_leaks=0 python $here/tools/asan_host_driver.py
# ... the same builders under ThreadSanitizer (they are multi-threaded)
g++ -O1 -g -fsanitize=thread -fno-omit-frame-pointer -std=c++17 -fPIC -shared -pthread -ffp-contract=off \
  -I$here/include $here/chroma_amd/csrc/bvh_build.cpp $here/chroma_amd/csrc/wide_build.cpp $here/chroma_amd/csrc/mesh_utils.cpp \
  -o $tmp/libhost_tsan.so
CHROMA_ASAN_LIB=$tmp/libhost_tsan.so LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS=halt_on_error=1 \
  python $here/tools/asan_host_driver.py
# ... and the CPU oracle (test infrastructure) through its own tests
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -std=gnu11 -ffp-contract=off -fno-fast-math \
  $(grep -q -m1 ' fma' /proc/cpuinfo && echo -mfma) -Wno-unused-function -o $tmp/liboracle_asan.so $here/oracle/chroma_oracle.c -lm -lpthread
(cd $here && CHROMA_ORACLE_LIBRARY=$tmp/liboracle_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle.py tests/test_wide_tree.py -x -q -p no:cacheprovider)
rm -rf $tmp
