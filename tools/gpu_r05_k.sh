#!/bin/bash
# round 5, call k: shared uploads of concurrent host-scalar MSMs -- parity, then the trait path A/B (bench prove lines with sharing on / off)
O=gpurun_out/r05_k; mkdir -p $O
python -m pytest tests/test_gpu_msm.py tests/test_gpu_trait_path.py tests/test_gpu_groth16.py -x -q -m gpu -k "share_one_upload or concurrent or trait_path" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
python - > $O/trait_ab.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import bindings as B, groth16 as g
B._check(hip.lib().csh_init(0))
for rnd in range(3):
    for share in (0, 1):
        B.tune_set("msm_share_uploads", share)
        s0 = B.tune_get("stat_uploads_shared")
        r = g.bench_synthetic(hip.BN254, 20, 11, with_rep3=True)
        print(json.dumps({"round": rnd, "msm_share_uploads": share, "uploads_shared": B.tune_get("stat_uploads_shared") - s0, "prove_ms": r["prove_ms"], "trait_path_ms": r["trait_path_ms"],
                          "trait_phases": r["trait_path_phases_ms"], "rep3_device_resident_3p_ms": r["rep3_three_parties_prove_ms"],
                          "rep3_seeded_alone_ms": r["rep3_trait_path"]["seeded_device_masks"]["one_party_alone_ms"],
                          "rep3_seeded_alone_phases": r["rep3_trait_path"]["seeded_device_masks"]["one_party_alone_phases_ms"],
                          "rep3_seeded_3p_ms": r["rep3_trait_path"]["seeded_device_masks"]["three_parties_one_gpu_ms"],
                          "rep3_host_alone_ms": r["rep3_trait_path"]["host_masks"]["one_party_alone_ms"]}), flush=True)
PY
tail -3 $O/pytest.log; cat $O/trait_ab.log | cut -c1-700
