#!/bin/bash
# ThreadSanitizer over libm6a_io's threaded paths, from plain C++ mains (TSan under an uninstrumented Python does not finish):
# dataprep with 8 threads and 16 KB index ranges (the parallel indexer's stitching, the ordered streaming writer), and loader +
# CSV writers (one-shot and sharded) with 8 threads.   tools/tsan_io.sh  -> appends to profiles/r06_sanitizers.txt (SAN_OUT overrides)
cd "$(dirname "$0")/.."
B=build/tsan; mkdir -p $B
F="-fsanitize=thread -g -O1 -std=c++17 -pthread -Iinclude -Im6anet_amd/csrc"
g++ $F m6anet_amd/csrc/m6a_io.cpp tools/tsan_dataprep_main.cpp -o $B/dataprep_tsan || exit 1
g++ $F m6anet_amd/csrc/m6a_io.cpp tools/tsan_io_main.cpp -o $B/io_tsan || exit 1
zcat tests/golden/ref_tests_data/eventalign.txt.gz > $B/eventalign.txt
mkdir -p $B/out $B/csv
{
echo "== ThreadSanitizer (tools/tsan_io.sh, $(date -u +%Y-%m-%dT%H:%MZ)): m6a_io_dataprep, 8 threads, M6A_IO_INDEX_RANGE_KB=16 (130 ranges)"
M6A_IO_INDEX_RANGE_KB=16 TSAN_OPTIONS=halt_on_error=0 $B/dataprep_tsan $B/eventalign.txt $B/out 8 2>&1 | grep -E "rc |WARNING|SUMMARY" | sort | uniq -c
cmp $B/out/eventalign.index tests/golden/ref_tests_data/eventalign.index && echo "index byte-identical to the reference's fixture; no ThreadSanitizer report"
echo "== ThreadSanitizer: m6a_io_load_sites + m6a_io_write_csv + m6a_io_csv_shard_size/_write, 8 threads"
TSAN_OPTIONS=halt_on_error=0 $B/io_tsan tests/golden/ref_tests_data $B/csv 2>&1 | grep -E "rc |WARNING|SUMMARY" | sort | uniq -c
} | tee -a ${SAN_OUT:-profiles/r06_sanitizers.txt}
