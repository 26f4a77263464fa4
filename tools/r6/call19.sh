#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c19; mkdir -p $out; cd $REPO
timeout 3000 python -m pytest tests -m gpu -q > $out/pytest_full.log 2>&1; echo "rc $?" >> $out/pytest_full.log; tail -8 $out/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log; tail -3 $out/smoke.log
