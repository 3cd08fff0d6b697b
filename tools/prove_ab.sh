# Interleaved A/B of two BUILDS on the synthetic 2^20 prove (tools/prove_loop.py): copy the build to compare against to
# co-snarks_amd/lib_old/ first (cp co-snarks_amd/lib/*.so co-snarks_amd/lib_old/; git-ignored, travels with gpurun), rebuild, then
#   gpurun -- 'bash tools/prove_ab.sh > gpurun_out/<tag>_ab.log'
# (A/B of a tune knob inside ONE build: tools/prove_ab_env.sh.)
R=$PWD
OLD="COSNARKS_HIP_LIB=$R/co-snarks_amd/lib_old/libcosnarks_hip.so COSNARKS_GROTH16_LIB=$R/co-snarks_amd/lib_old/libcosnarks_groth16.so"
for round in 1 2 3; do
  for mode in old new; do
    if [ $mode = old ]; then E="$OLD"; else E="X=1"; fi
    env $E python tools/prove_loop.py 20 14 | tail -9 | awk -v m=$mode -v r=$round '/prove/ {gsub(/[\[\],]/,""); w+=$3; s+=$4; f+=$5; n++} END {printf "%s round %s: witness %.3f msm %.3f finish %.3f total %.3f (n=%d)\n", m, r, w/n, s/n, f/n, (w+s+f)/n, n}'
  done
done
