#!/bin/bash
# usage: pmc_pass.sh <workload> <out.txt> <counter> [counter ...]   -- one rocprofv3 --pmc pass (no trace domains), summarised
# PMC_FROM=<kernel substring>: summarise only the dispatches from the first such kernel on (tools/pmc_summary.py --from)
set -e
W=$1; OUT=$2; shift 2
TAG=${W// /_}            # "batch 32" -> batch_32 for the scratch paths; the workload words stay separate arguments
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
rocprofv3 --pmc "$@" -d /tmp/pmc_$TAG -o p -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py $W > /tmp/pmc_$TAG.log 2>&1 || tail -5 /tmp/pmc_$TAG.log
DB=$(find /tmp/pmc_$TAG -name "*.db" | head -1)
(echo "# rocprofv3 --pmc $* -- python tools/pmc_workload.py $W"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $DB ${PMC_FROM:+--from $PMC_FROM}) > $GRAFT_REPO_ROOT/$OUT
