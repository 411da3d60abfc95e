# GPU box: the -m gpu suites against the ASan + UBSan build of the HOST half of libm6a_hip.so (tests/sanitize.sh builds it into
# build/sanitize/; device code is not instrumented).  Output: gpurun_out/r06_san_gpu_full.txt
cd $GRAFT_REPO_ROOT
CLANG_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export M6A_HIP_LIB=$PWD/build/sanitize/libm6a_hip.so M6A_IO_LIB=$PWD/build/sanitize/libm6a_io.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$CLANG_RT timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py tests/test_reference_at_scale.py -v -m gpu -p no:cacheprovider -k "not bench and not cli and not full_size and not configs1 and not feed_probe and not plain_c and not full_pipeline" > gpurun_out/r06_san_gpu_full.txt 2>&1
echo "exit code $?" >> gpurun_out/r06_san_gpu_full.txt
grep -E "PASSED|FAILED|ERROR" gpurun_out/r06_san_gpu_full.txt | awk '{print $NF}' | sort | uniq -c
grep -E "^(FAILED|ERROR) " gpurun_out/r06_san_gpu_full.txt | sed "s/ - .*//" | head -80; grep -E "Error in dlopen|AddressSanitizer|runtime error" gpurun_out/r06_san_gpu_full.txt | sort | uniq -c | head; tail -3 gpurun_out/r06_san_gpu_full.txt
