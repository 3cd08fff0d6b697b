#!/bin/bash
mkdir -p gpurun_out; O=$PWD/gpurun_out
hipcc -O2 --offload-arch=gfx950 -o /tmp/prefault_probe tools/experiments/prefault_probe.cpp -lpthread && timeout 300 /tmp/prefault_probe > $O/r04_b_prefault_probe.jsonl 2>&1; cat $O/r04_b_prefault_probe.jsonl
