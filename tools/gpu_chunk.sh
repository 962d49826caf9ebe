L=$PWD/transformergrooveinfilling_amd/lib
for lib in libgroove_hip.so libgroove_chunk4096.so libgroove_chunk8192.so; do
  for sh in 11 7; do
    echo "$lib: $(GT_LIB_PATH=$L/$lib python tools/shape_bench.py --only $sh --steps 20 --warmup 5 2>/dev/null | tail -1)"
  done
done
