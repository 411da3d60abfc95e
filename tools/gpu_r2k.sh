#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
cat /sys/kernel/mm/transparent_hugepage/enabled > $O/r2k_thp.txt 2>&1
python tools/measure_misc.py > $O/r2k_misc.json 2> $O/r2k_misc.err
echo done
