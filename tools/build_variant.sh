#!/bin/bash
# A/B variants of the library for one gpurun call: tools/build_variant.sh NAME TU[,TU] "-DMACRO=.. ..." rebuilds the named translation
# units (groove_hip | groove_seq_fwd | groove_seq_bwd | groove_seq64) with the extra flags and links transformergrooveinfilling_amd/lib/libgroove_NAME.so
# from them and the stock objects of the others (run csrc/build.sh first).  Select with GT_LIB_PATH.
set -e
name=$1; tus=$2; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/transformergrooveinfilling_amd/csrc"; out="$root/transformergrooveinfilling_amd/lib"; obj="$out/obj"
pids=()
for tu in ${tus//,/ }; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -c "$src/$tu.hip" -o "$obj/${tu}_$name.o" "$@" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=""
for t in groove_hip groove_seq_fwd groove_seq_bwd groove_seq64; do
  if [[ ",$tus," == *",$t,"* ]]; then objs="$objs $obj/${t}_$name.o"; else objs="$objs $obj/$t.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$out/libgroove_$name.so"
echo "built $out/libgroove_$name.so"
