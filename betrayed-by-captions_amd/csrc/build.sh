#!/bin/bash
# Build libcgg_hip.so (gfx950 only) in-tree next to the Python package.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
OBJS=""
pids=""
for f in "$HERE"/*.hip; do
  o="$OUT/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/cgg_common.h" -nt "$o" ] || [ "$HERE/x3.h" -nt "$o" ] || [ "$HERE/msda_common.h" -nt "$o" ] || [ "$HERE/../../include/cgg_hip.h" -nt "$o" ]; then
    # a source may ask for extra compiler flags on a "// build-flags: ..." line (e.g. the MFMA VGPR form)
    extra="$(grep -m1 '^// build-flags:' "$f" | sed 's|^// build-flags:||')"
    $HIPCC $FLAGS $extra -c "$f" -o "$o" &
    pids="$pids $!"
  fi
  OBJS="$OBJS $o"
done
for p in $pids; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libcgg_hip.so" $OBJS
echo "built $OUT/libcgg_hip.so"
