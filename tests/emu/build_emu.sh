#!/bin/bash
# TEST INFRASTRUCTURE: build the host fiber-emulator flavour of the kernels (no GPU needed).  The same four translation
# units as the HIP build, compiled as C++ in parallel; the emulator runtime (GT_EMU_IMPL) lives in the first one.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
src="$root/transformergrooveinfilling_amd/csrc"
outlib="${GT_EMU_OUT:-$here/libgroove_emu.so}"
obj="$here/obj_$(basename "$outlib" .so)"
mkdir -p "$obj"
flags="-O2 -g -std=c++17 -DGT_EMU -x c++ -I$here -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-unknown-pragmas -Wno-unused-but-set-variable -Wno-psabi"
pids=()
g++ $flags -DGT_EMU_IMPL -c "$src/groove_hip.hip" -o "$obj/groove_hip.o" "$@" & pids+=($!)
g++ $flags -c "$src/groove_seq_fwd.hip" -o "$obj/groove_seq_fwd.o" "$@" & pids+=($!)
g++ $flags -c "$src/groove_seq_bwd.hip" -o "$obj/groove_seq_bwd.o" "$@" & pids+=($!)
g++ $flags -c "$src/groove_seq64.hip" -o "$obj/groove_seq64.o" "$@" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
g++ -shared -fPIC "$obj"/groove_hip.o "$obj"/groove_seq_fwd.o "$obj"/groove_seq_bwd.o "$obj"/groove_seq64.o -o "$outlib"
echo "built $outlib"
