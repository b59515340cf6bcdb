#!/bin/bash
# Builds libdehaze_hip.so (gfx950 code objects) in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../dehaze_hip/libdehaze_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE -Wno-unused-result"
mkdir -p "$HERE/build"
objs=()
for f in "$HERE"/*.hip; do
  o="$HERE/build/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ] || [ "$ROOT/include/dehaze_hip.h" -nt "$o" ]; then
    echo "hipcc -c $(basename "$f")"
    "$HIPCC" $FLAGS -c "$f" -o "$o" &
  fi
  objs+=("$o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
echo "built $OUT"
