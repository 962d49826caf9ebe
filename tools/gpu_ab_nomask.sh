# round 6 measurement: the FFN2 dgrad (EPI_MASK_NZ) without its mask read -- upper bound of a 1-bit keep mask in place of hact (WRONG RESULTS in the variant)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m
mkdir -p $O
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
bash tools/ab_variants.sh "7 11 13 6" transformergrooveinfilling_amd/lib/libgroove_nomask.so > $O/ab.txt 2>&1
cat $O/ab.txt
python tools/class_profile.py 7 > $O/class_profile_7.txt 2>&1
GT_LIB_PATH=$PWD/transformergrooveinfilling_amd/lib/libgroove_nomask.so python tools/class_profile.py 7 > $O/class_profile_7_nomask.txt 2>&1
python tools/class_profile.py 11 > $O/class_profile_11.txt 2>&1
GT_LIB_PATH=$PWD/transformergrooveinfilling_amd/lib/libgroove_nomask.so python tools/class_profile.py 11 > $O/class_profile_11_nomask.txt 2>&1
grep -h "ffn\|kernel time" $O/class_profile_*.txt
