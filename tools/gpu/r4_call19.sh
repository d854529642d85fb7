#!/bin/bash
# round-4 GPU call 19: where should the weight-stationary form of the normalising GEMVs start? (lane-count sweep + parity counts)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
(timeout 300 python tools/batch_bench.py 0.6b 16,32,48,64 56; FQ3_BENCH_NORM_SKINNY_ABOVE=8 timeout 300 python tools/batch_bench.py 0.6b 16,32,48,64 56) > $O/c19_frames_0p6b.txt 2>&1; grep "ms per" $O/c19_frames_0p6b.txt
(timeout 300 python tools/batch_bench.py 1.7b 32,64 56; FQ3_BENCH_NORM_SKINNY_ABOVE=8 timeout 300 python tools/batch_bench.py 1.7b 16,32,64 56) > $O/c19_frames_1p7b.txt 2>&1; grep "ms per" $O/c19_frames_1p7b.txt
FQ3_TEST_NORM_SKINNY_ABOVE=8 timeout 900 python -m pytest tests/test_gpu_batch_fulldepth.py -q -m gpu -s -k bf16_mfma 2>&1 | grep "parity\] batch" | cut -c1-260 > $O/c19_parity_above8.txt; cat $O/c19_parity_above8.txt
