# run encoder parity tests against a variant lib
for v in "$@"; do echo "== $v"; M6A_HIP_LIB=$PWD/tools/ko/libm6a_ab_$v.so python -m pytest tests/test_gpu_parity.py -q -x -k "encoder or read_prob or general16 or bundled" 2>&1 | tail -1; done
