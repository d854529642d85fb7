#!/bin/bash
# usage: pmc_pass.sh <workload> <out.txt> <counter> [counter ...]   -- one rocprofv3 --pmc pass (no trace domains), summarised
set -e
W=$1; OUT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$W
rocprofv3 --pmc "$@" -d /tmp/pmc_$W -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py $W > /tmp/pmc_$W.log 2>&1 || tail -5 /tmp/pmc_$W.log
DB=$(find /tmp/pmc_$W -name "*.db" | head -1)
(echo "# rocprofv3 --pmc $* -- python tools/pmc_workload.py $W"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $DB) > $GRAFT_REPO_ROOT/$OUT
