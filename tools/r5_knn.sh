#!/bin/bash
# Runs ON THE GPU BOX: the kNN graph kernels with the 64-bit-key lists (default build) against the (value, index) lists of
# rounds 2-4 (tools/build_variant.sh list32 "-DPCOPS_KNN_LIST32=1"), then the kNN tests.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== key64 (default)"; python tools/bench_knn.py
echo "== list32"; PCOPS_LIB=$PWD/scanobjectnn_amd/libpcops_list32.so python tools/bench_knn.py
echo "== key64 again"; python tools/bench_knn.py
} > gpurun_out/knn_key64_ab.txt 2>&1
tail -20 gpurun_out/knn_key64_ab.txt
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q 2>&1 | tail -15
