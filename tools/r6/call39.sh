#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
probe() { # $1 = label, rest = gemm_bench specs
  label=$1; shift
  REPS=3000 python tools/r6/gemm_bench.py "$@" > /tmp/gb.out 2>&1 &
  pid=$!
  sleep 6
  for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | tr '\n' ' ' | sed "s/^/[$label] /"; echo
    sleep 0.7
  done
  wait $pid
  cat /tmp/gb.out | tail -2
}
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ' '; echo " [idle]"
probe fwd1536 fwd:65536x1536x2048
probe fwd160 fwd:524288x160x960
probe wgrad960 wgrad:524288x960x160
probe wgraddy160 wgraddy:524288x160x960
probe dwlike lwgrad:524288x256x21
