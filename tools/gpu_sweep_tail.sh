for pct in 40 50 60 70; do for ks in 2 3 4; do
  r=$(GT_SEQ_RIDE_LAST_PCT=$pct GT_SEQ_TAIL_KS=$ks python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f'%d['ms_per_step'], d['kernel_classes_us_per_step'])")
  echo "pct $pct ks $ks: $r"
done; done
