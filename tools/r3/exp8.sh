#!/bin/bash
out=gpurun_out/r3h; mkdir -p $out
export PROBE_SHAPES=4096x160x960,4096x64x384,4096x960x160 PROBE_KINDS=fwd
for lib in libdl3_timing.so libdl3_timing_DDL3_DBG_NOSTORE.so libdl3_timing_DDL3_DBG_NOSTAT.so libdl3_timing_DDL3_DBG_NOSTORE_DDL3_DBG_NOSTAT.so; do
echo "== $lib"; PROBE_LIB=$lib python tools/r3/phase_probe.py 128 2>&1 | grep -v amdgpu.ids
done > $out/phase_epi.log
cat $out/phase_epi.log
