cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h
mkdir -p $O; rm -f $O/*.txt
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1     # (warm the box)
for c in 512 1024 2048; do
  GT_WGRAD128H_MIN_CHUNK=$c python tools/shape_bench.py --only 9 --steps 100 2>/dev/null | tail -1 | sed "s/^/bf16 chunk>=$c  /" >> $O/shapes.txt
  GT_WGRAD128H_MIN_CHUNK=$c python tools/shape_bench.py --only 12 --steps 100 2>/dev/null | tail -1 | sed "s/^/p2   chunk>=$c  /" >> $O/shapes.txt
  GT_WGRAD128_MIN_CHUNK=$c python tools/shape_bench.py --only 6 --steps 100 2>/dev/null | tail -1 | sed "s/^/fp32 chunk>=$c  /" >> $O/shapes.txt
done
cat $O/shapes.txt
