# Interleaved A/B of csh_msm's handling of concurrent calls over one host slice on the trait path of the synthetic 2^20 prove:
# CSH_MSM_SHARE_UPLOADS=1 (share the upload only, rounds 5-6) against 2 (the uploading call runs the others as one multi-MSM).
for round in 1 2 3; do
  for mode in 1 2; do
    CSH_MSM_SHARE_UPLOADS=$mode python tools/prove_probe.py --modes policy --iters 11 --rep3 | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
r=rows[-1]
print('share=$mode round $round: trait %.2f (min %.2f) phases %s | device-resident %.2f | rep3 host-masks party %.2f seeded %.2f | 3 parties seeded %.1f host %.1f | checks %s %s' % (r['trait_path_ms'], r.get('trait_path_ms_min', 0), r['trait_path_phases_ms'], r['prove_ms'], r['rep3_trait_path']['host_masks']['one_party_alone_ms'], r['rep3_trait_path']['seeded_device_masks']['one_party_alone_ms'], r['rep3_trait_path']['seeded_device_masks']['three_parties_one_gpu_ms'], r['rep3_trait_path']['host_masks']['three_parties_one_gpu_ms'], r['trait_path_closed_form_check'], r['rep3_trait_path']['host_masks']['proofs_equal_plain']))"
  done
done
