#!/bin/bash
# Diagnostic builds of the B-stationary GEMM: libpasero_hip_<variant>.so next to the real library, selected with
# PASERO_HIP_LIB (ABL_* builds are timing only: their results are wrong).
# usage: tools/gemmbs_ablate.sh "FLAGS1" "FLAGS2" ...    e.g.  tools/gemmbs_ablate.sh STAMPS "PRIO=1" "SLEEP=2"
set -e
cd "$(dirname "$0")/../pasero_amd/csrc"
make -j8 >/dev/null
OBJS=$(ls *.o | grep -v '^gemmbs.o$')
for v in "$@"; do
  name=$(echo "$v" | tr ' =' '__')
  defs=""
  for d in $v; do defs="$defs -DPKBS_$d"; done
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $defs -c gemmbs.hip -o /tmp/gemmbs_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/gemmbs_$name.o -Wl,--version-script=exports.map -o ../libpasero_hip_$name.so 2>/dev/null
done
ls ../libpasero_hip_*.so
