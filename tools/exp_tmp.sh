for v in "" build_exp/lib_ALIGN.so; do
  M6A_HIP_LIB=${v:+$PWD/$v} python bench.py --workload ragged --no-cpu-baseline --no-live-traffic --steps 20 --warmup 5 | python -c "
import json,sys; m=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', m['kernels'])"
done
