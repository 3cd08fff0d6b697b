import sys, os
sys.path.insert(0, os.getcwd())
import torch
import cosnarks_amd as ca
from cosnarks_amd import groth16 as g
r = g.bench_synthetic(0, 20, iters=3)
print(r)
