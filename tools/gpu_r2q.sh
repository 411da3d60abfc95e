#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r2q_pytest.log 2>&1
grep -n "passed\|failed" $O/r2q_pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/r2q_smoke.log 2>&1; tail -1 $O/r2q_smoke.log
bash tools/gpu_final.sh r2q
