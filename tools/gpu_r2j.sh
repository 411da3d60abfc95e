#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r2j_pytest.log 2>&1
grep -n "passed\|failed" $O/r2j_pytest.log | tail -2
python tools/measure_cli.py > $O/r2j_cli.json 2> $O/r2j_cli.err
python tools/host_path_sweep.py > $O/r2j_host_sweep.json 2> $O/r2j_host_sweep.err
echo done
