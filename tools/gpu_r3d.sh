#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
L=$R/transformergrooveinfilling_amd/lib
GT_LIB_PATH=$L/libgroove_pfall.so timeout 600 python -m pytest tests/test_hip_parity.py -q -x -k "sequence_resident or train_step" --timeout 600 2>&1 | tail -3 | tee gpurun_out/r3d_pytest.log
for round in 1 2 3; do
  for so in hip pfall; do
    echo "$so $(GT_LIB_PATH=$L/libgroove_$so.so python tools/shape_bench.py --only 2 --steps 300 2>/dev/null | tail -1)"
    echo "$so ride=0 $(GT_SEQ_RIDE=0 GT_LIB_PATH=$L/libgroove_$so.so python tools/shape_bench.py --only 2 --steps 300 2>/dev/null | tail -1)"
  done
done | tee gpurun_out/r3d_ab.log
python tools/wg_unit_bench.py 64 2>&1 | grep -v amdgpu | tee gpurun_out/r3d_unit.log
cd /tmp && export TMPDIR=/tmp
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rocprofv3 --pmc $pmc --output-format csv -d $R/gpurun_out/r3d_pmc -- python3 $R/tools/wg_unit_bench.py 64 > /dev/null 2>&1
  f=$(find $R/gpurun_out/r3d_pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:30]
    if "seq_tail" not in k: continue
    key = (k, r.get("Grid_Size"))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(key, r["Counter_Name"])] += 1
for key in acc:
    print(key, {c: round(v / cnt[(key, c)]) for c, v in acc[key].items()})
PY
  rm -rf $R/gpurun_out/r3d_pmc
done 2>&1 | tee $R/gpurun_out/r3d_pmc.log
