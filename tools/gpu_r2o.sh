#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host_pointer or infer_end or cli or encoder" > $O/r2o_pytest.log 2>&1
grep -n "passed\|failed" $O/r2o_pytest.log | tail -2
python tools/measure_misc.py > $O/r2o_misc.json 2> $O/r2o_misc.err
python tools/measure_cli.py > $O/r2o_cli.json 2> $O/r2o_cli.err
echo done
