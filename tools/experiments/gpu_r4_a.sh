#!/bin/bash
# Round 4, run A: the raw-boundary vectors of the reference's PLONK fixtures on the device + the PCIe staging probe.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 600 python -m pytest tests/test_gpu_plonk_vectors.py tests/test_gpu_plonk_honk.py -m gpu -q -p no:cacheprovider > $O/r04_a_pytest_plonk.log 2>&1; tail -2 $O/r04_a_pytest_plonk.log
hipcc -O2 --offload-arch=gfx950 -o /tmp/pcie_probe tools/experiments/pcie_probe.cpp -lpthread && timeout 300 /tmp/pcie_probe > $O/r04_a_pcie_probe.jsonl 2>&1; cat $O/r04_a_pcie_probe.jsonl
