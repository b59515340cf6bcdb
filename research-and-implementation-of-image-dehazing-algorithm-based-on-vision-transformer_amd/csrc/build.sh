#!/bin/bash
# Builds libdehaze_hip.so (gfx950 code objects) in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../dehaze_hip/libdehaze_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE -Wno-unused-result"
mkdir -p "$HERE/build"
# build id = content hash of every source the library is made of; dhz_build_id() returns it, the PMC passes stamp it into
# profiles/pmc_traffic.json and bench.py refuses a traffic figure whose stamp is not the loaded library's
ID=$(cat "$HERE"/*.hip "$HERE"/*.h "$ROOT/include/dehaze_hip.h" | sha256sum | cut -c1-16)
LINE="#define DHZ_BUILD_ID \"$ID${DHZ_BUILD_TAG:+-$DHZ_BUILD_TAG}\""
if [ ! -f "$HERE/build/build_id.h" ] || [ "$(cat "$HERE/build/build_id.h")" != "$LINE" ]; then      # the whole line: a changed tag alone re-stamps too
  echo "$LINE" > "$HERE/build/build_id.h"
fi
FLAGS="$FLAGS -I$HERE/build"
objs=()
pids=()
for f in "$HERE"/*.hip; do
  o="$HERE/build/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ] || [ "$HERE/tok_epilogue.h" -nt "$o" ] || [ "$ROOT/include/dehaze_hip.h" -nt "$o" ] \
     || { [ "$(basename "$f")" = api.hip ] && [ "$HERE/build/build_id.h" -nt "$o" ]; }; then
    echo "hipcc -c $(basename "$f")"
    rm -f "$o"                                   # a failed compile must not leave the previous object behind
    "$HIPCC" $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
  objs+=("$o")
done
for p in "${pids[@]}"; do wait "$p" || { echo "build.sh: compile failed" >&2; exit 1; }; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}" -ldl
echo "built $OUT"
