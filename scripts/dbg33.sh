cd $GRAFT_REPO_ROOT
for n in "$@"; do echo "== $n"; SQ_LIB=$PWD/scripts/build/libsqgpu_$n.so timeout 300 python -m pytest "tests/test_gpu_vs_oracle.py::test_uniform_length_kernels_every_alignment[33]" -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2; done
