# Interleaved A/B of an environment-set tune knob on the synthetic 2^20 prove (here: the witness map's kept coset table, CSH_H_TABLE_CACHE).
for round in 1 2 3; do
  for mode in off on; do
    if [ $mode = off ]; then E="CSH_H_TABLE_CACHE=0"; else E="X=1"; fi
    env $E python tools/prove_loop.py 20 14 | tail -9 | awk -v m=$mode -v r=$round '/prove/ {gsub(/[\[\],]/,""); w+=$3; s+=$4; f+=$5; n++} END {printf "cache %s round %s: witness %.3f msm %.3f finish %.3f total %.3f (n=%d)\n", m, r, w/n, s/n, f/n, (w+s+f)/n, n}'
  done
done
