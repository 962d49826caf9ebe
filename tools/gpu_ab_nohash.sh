# round 6 measurement: the one-kernel-per-op path without its per-element dropout hash (WRONG RESULTS in the variant): the bound of any scheme that saves keep bits
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h
mkdir -p $O
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
bash tools/ab_variants.sh "7 11 13 6 9 4" transformergrooveinfilling_amd/lib/libgroove_nohash.so > $O/ab.txt 2>&1
cat $O/ab.txt
python tools/class_profile.py 11 > $O/class_profile_11.txt 2>&1
GT_LIB_PATH=$PWD/transformergrooveinfilling_amd/lib/libgroove_nohash.so python tools/class_profile.py 11 > $O/class_profile_11_nohash.txt 2>&1
python tools/class_profile.py 7 > $O/class_profile_7.txt 2>&1
GT_LIB_PATH=$PWD/transformergrooveinfilling_amd/lib/libgroove_nohash.so python tools/class_profile.py 7 > $O/class_profile_7_nohash.txt 2>&1
head -14 $O/class_profile_11.txt $O/class_profile_11_nohash.txt $O/class_profile_7.txt $O/class_profile_7_nohash.txt
