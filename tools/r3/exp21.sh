#!/bin/bash
mkdir -p gpurun_out/r3u; ./build_variants/coexec_probe | tee gpurun_out/r3u/coexec.log
