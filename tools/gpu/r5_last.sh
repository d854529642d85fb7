#!/bin/bash
# round-5 last GPU call: the whole GPU suite + smoke() on the final commit
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/last_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/last_tests.log; tail -4 $O/last_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/last_smoke.txt 2>&1; tail -1 $O/last_smoke.txt
