for c in 0 1 2 3; do echo "cfg5 wide cfg $c (min rows 256)"; TXO_WIDE_CFG=$c TXO_WIDE_MIN_ROWS=256 python probes/cfg5.py 2>&1 | grep -E "mixed" | tail -1; done
for c in off 0 1 2 3; do
  if [ $c = off ]; then export TXO_WIDE_MIN_ROWS=512; else export TXO_WIDE_MIN_ROWS=128; export TXO_WIDE_CFG=$c; fi
  echo "b256 wide cfg $c"; python probes/b256_sweep.py 256 2>&1 | grep -E "lanes=(-|1) graph=-" | head -2
done
