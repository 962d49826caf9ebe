#!/bin/bash
# Build libgroove_hip.so for gfx950 (cross-compiles without a GPU).  The .so stays in-tree
# (git-ignored, but shipped to the GPU box with the snapshot).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared \
    -Wall -Wno-unused-function -Wno-unused-variable \
    "$here/groove_hip.hip" -o "$out/libgroove_hip.so" "$@"
echo "built $out/libgroove_hip.so"
