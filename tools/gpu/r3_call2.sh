#!/bin/bash
# round-3 GPU call 2: harness (rows-per-workgroup variants, last-arriver fusion), the whole GPU suite, packed-prefill decision, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
(timeout 300 tools/microbench/kernel_chain batch 10 > $O/kc_batch2.txt 2>&1; echo "rc $?" >> $O/kc_batch2.txt)
(timeout 200 tools/microbench/kernel_chain fuse 10 > $O/kc_fuse.txt 2>&1; echo "rc $?" >> $O/kc_fuse.txt)
grep -v " ok$" $O/kc_batch2.txt | tail -30; cat $O/kc_fuse.txt
(timeout 1500 python -m pytest tests -q -m gpu -x > $O/t2.log 2>&1; echo "rc $?" >> $O/t2.log)
tail -8 $O/t2.log
(timeout 300 python tools/batch_bench.py 0.6b 8,16 48 > $O/bb2_0p6b.txt 2>&1)
(timeout 300 python tools/batch_bench.py 1.7b 8,16 48 > $O/bb2_1p7b.txt 2>&1)
cat $O/bb2_0p6b.txt $O/bb2_1p7b.txt | grep "ms per"
(timeout 500 python tools/packed_prefill_check.py 8,16 > $O/packed_prefill.txt 2> $O/packed_prefill.err)
cat $O/packed_prefill.txt; tail -3 $O/packed_prefill.err
(timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 > $O/bench1.json 2> $O/bench1.err; echo "rc $?" >> $O/bench1.err)
tail -3 $O/bench1.err; head -c 3000 $O/bench1.json
