#!/bin/bash
# round 5, call b: balanced windows -- parity (new test, plan fuzz, split tests), then interleaved A/B per size
O=gpurun_out/r05_b; mkdir -p $O
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py -x -q -m gpu -k "balanced or fuzz or window or closed_form or edge or small" --durations=5 > $O/pytest_msm.log 2>&1; echo "pytest exit $?" >> $O/pytest_msm.log
for ln in 14 15 16 17 18 19 20; do
  python tools/msm_ab.py --job 0:0:$ln --rounds 8 --reps 10 uniform=msm_balanced=0 balanced=msm_balanced=1 >> $O/ab_balanced.log 2>&1
done
python tools/msm_ab.py --job 1:0:16 --rounds 6 --reps 10 uniform=msm_balanced=0 balanced=msm_balanced=1 >> $O/ab_balanced.log 2>&1
python tools/msm_ab.py --job 0:1:16 --rounds 6 --reps 10 uniform=msm_balanced=0 balanced=msm_balanced=1 >> $O/ab_balanced.log 2>&1
tail -5 $O/pytest_msm.log; grep -h '"variant": "balanced", "tune"' $O/ab_balanced.log | cut -c1-330
