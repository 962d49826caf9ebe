cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g
mkdir -p $O; rm -f $O/*.txt
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1     # (warm the box)
python tools/shape_bench.py --only 5 --steps 40 2>/dev/null | tail -1 | sed 's/^/row-fused (default)      /' >> $O/shapes.txt
GT_ROW_FUSE_BIG_MAX_D=0 python tools/shape_bench.py --only 5 --steps 40 2>/dev/null | tail -1 | sed 's/^/gemm64 + LN pass         /' >> $O/shapes.txt
GT_ROW_FUSE_BIG_MAX_D=0 GT_LN_XCHG=1 python tools/shape_bench.py --only 5 --steps 40 2>/dev/null | tail -1 | sed 's/^/gemm64 + LN row exchange /' >> $O/shapes.txt
python tools/shape_bench.py --only 5 --steps 40 2>/dev/null | tail -1 | sed 's/^/row-fused (default)      /' >> $O/shapes.txt
for i in 9 12; do python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 >> $O/shapes.txt; done
GT_ROW_FUSE_BIG_MAX_D=0 GT_LN_XCHG=1 python tools/class_profile.py 5 > $O/class_profile_5_xchg.txt 2>&1
cat $O/shapes.txt
