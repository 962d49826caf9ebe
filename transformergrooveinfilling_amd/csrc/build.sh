#!/bin/bash
# Build libgroove_hip.so for gfx950 (cross-compiles without a GPU).  The .so stays in-tree
# (git-ignored, but shipped to the GPU box with the snapshot).  Four translation units, compiled in parallel.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../lib"
obj="$out/obj"
mkdir -p "$obj"
flags="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
pids=()
for tu in groove_hip groove_seq_fwd groove_seq_bwd groove_seq64; do
  /opt/rocm/bin/hipcc $flags -c "$here/$tu.hip" -o "$obj/$tu.o" "$@" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$obj"/groove_hip.o "$obj"/groove_seq_fwd.o "$obj"/groove_seq_bwd.o "$obj"/groove_seq64.o -o "$out/libgroove_hip.so"
echo "built $out/libgroove_hip.so"
