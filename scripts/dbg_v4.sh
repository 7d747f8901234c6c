cd $GRAFT_REPO_ROOT
for u in 1 3 4 5 15 16 17 27 31 32; do echo -n "v4nw1 U=$u: "; SQ_LIB=$PWD/scripts/build/libsqgpu_v4nw1.so timeout 100 python scripts/dbg33b.py $u 2>&1 | head -1; done
for u in 33 34 37 40 47 48 49 63 64; do echo -n "v4nw2 U=$u: "; SQ_LIB=$PWD/scripts/build/libsqgpu_v4nw2.so timeout 100 python scripts/dbg33b.py $u 2>&1 | head -1; done
SQ_LIB=$PWD/scripts/build/libsqgpu_v4nw2.so timeout 300 python -m pytest "tests/test_gpu_vs_oracle.py::test_uniform_length_kernels_every_alignment[33]" "tests/test_gpu_vs_oracle.py::test_uniform_length_kernels_every_alignment[63]" "tests/test_gpu_vs_oracle.py::test_uniform_length_kernels_every_alignment[64]" -q -m gpu -p no:cacheprovider 2>&1 | tail -2
timeout 500 bash scripts/ab_lib.sh base prod v4nw5
