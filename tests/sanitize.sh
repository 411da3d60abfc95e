#!/bin/bash
# SURVEY.md section 5 / VERDICT r3 item 8: AddressSanitizer + UndefinedBehaviorSanitizer builds of the host-side native code --
# libm6a_io.so (loader, writers, dataprep), the oracle, tools/feed_probe, and the HOST half of libm6a_hip.so (device code is
# not instrumented: -fno-gpu-sanitize) -- and the whole CPU test suite run against them.
#   tests/sanitize.sh            -> profiles/r06_sanitizers.txt (SAN_OUT overrides)   (under tests/: it builds and runs the oracle, which is test infrastructure)
set -u
cd "$(dirname "$0")/.."
B=build/sanitize
mkdir -p $B
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
OUT=${SAN_OUT:-profiles/r06_sanitizers.txt}
{
echo "== build ($(date -u +%Y-%m-%dT%H:%MZ), $(gcc --version | head -1))"
set -x
# -DM6A_IO_TEST_HOOKS: the fault-injection hook of tests/test_dataprep.py is compiled only into test builds; the sanitized copy IS one
g++ $SAN -std=c++17 -fPIC -shared -pthread -Wall -Wextra -DM6A_IO_TEST_HOOKS -Iinclude -Im6anet_amd/csrc m6anet_amd/csrc/m6a_io.cpp -o $B/libm6a_io.so || exit 1
gcc $SAN -std=gnu11 -fPIC -ffp-contract=off -shared oracle/m6a_oracle.c -o $B/libm6a_oracle.so -lm -lpthread || exit 1
gcc $SAN -Wall -Iinclude tools/feed_probe.c -ldl -o $B/feed_probe || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer \
    -shared-libsan -DM6A_MT_JUMP_PATH="\"$PWD/m6anet_amd/assets/mt19937_jump.bin\"" -Iinclude -Im6anet_amd/csrc \
    m6anet_amd/csrc/m6a_kernels.hip m6anet_amd/csrc/m6a_pool_reg.hip m6anet_amd/csrc/m6a_pool_rtab.hip m6anet_amd/csrc/m6a_api.hip \
    m6anet_amd/csrc/m6a_host_ring.hip m6anet_amd/csrc/m6a_job.hip m6anet_amd/csrc/m6a_comm.hip m6anet_amd/csrc/m6a_validate.hip -o $B/libm6a_hip.so
HIP_SAN=$?
set +x
echo "libm6a_hip.so host-side sanitizer build: rc=$HIP_SAN"
export M6A_IO_LIB=$PWD/$B/libm6a_io.so M6A_IO_LIB_HOOKS=$PWD/$B/libm6a_io.so M6A_ORACLE_LIB=$PWD/$B/libm6a_oracle.so
PRE="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== CPU suite against the sanitized libm6a_io.so + oracle (LD_PRELOAD=$PRE)"
LD_PRELOAD=$PRE python -m pytest tests/ -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
echo "rc=$?"
if [ "$HIP_SAN" = 0 ]; then
  echo "== the ABI / error-path tests against the sanitized HOST half of libm6a_hip.so (clang runtime)"
  CLANG_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
  M6A_HIP_LIB=$PWD/$B/libm6a_hip.so LD_PRELOAD=$CLANG_RT python -m pytest tests/test_abi_and_host.py -x -q -p no:cacheprovider 2>&1 | tail -8
  echo "rc=$?"
fi
echo "== feed_probe (plain C caller) builds under the sanitizers; it needs a GPU to run"
} > $OUT 2>&1
tail -30 $OUT
