#!/bin/bash
# TEST INFRASTRUCTURE: build the host fiber-emulator flavour of the kernels (no GPU needed).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
g++ -O2 -g -std=c++17 -DGT_EMU -DGT_EMU_IMPL -x c++ -I"$here" -fPIC -shared \
    -Wall -Wno-unused-function -Wno-unused-variable -Wno-unknown-pragmas -Wno-unused-but-set-variable \
    "$root/transformergrooveinfilling_amd/csrc/groove_hip.hip" -o "${GT_EMU_OUT:-$here/libgroove_emu.so}" "$@"
echo "built ${GT_EMU_OUT:-$here/libgroove_emu.so}"
