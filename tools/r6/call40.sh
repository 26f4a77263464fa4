#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
t0=$(date +%s); python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; t1=$(date +%s); echo "smoke $((t1-t0)) s"
python bench.py > /tmp/b.json 2>/tmp/b.err; t2=$(date +%s); echo "bench.py default $((t2-t1)) s"; tail -1 /tmp/b.json | cut -c1-400
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
