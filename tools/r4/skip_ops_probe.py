"""TIMING PROBE ONLY (results of the step are wrong): run bench.py with some launches dropped from the plan, to size what
a launch costs inside the replayed hipGraph before building anything that would remove it.
  DL3_PROBE_SKIP=dl3_bn_finalize,dl3_bn_bwd_finalize python tools/r4/skip_ops_probe.py --batch 2 --steps 100 ..."""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dl3_amd import engine as E  # noqa: E402

skip = set(filter(None, os.environ.get("DL3_PROBE_SKIP", "").split(",")))
_op = E.Engine.op


def op(self, lst, name, *args):
    rec = _op(self, lst, name, *args)
    if name in skip:
        lst.pop()
    return rec


E.Engine.op = op
sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
