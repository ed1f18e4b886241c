export TMPDIR=/tmp
OUT=$PWD/gpurun_out/dn; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_$c; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $d -o r -- python $OLDPWD/tools/kbench.py --layers enc1.0s2,up4 --which fwd,dgrad --iters 3 > $OUT/$c.log 2>&1)
  db=$(find $d -name "*.db" | head -1)
  python tools/rocpd_pmc.py $db > $OUT/$c.txt 2>&1
  grep -A1 "down2\|up2" $OUT/$c.txt
done
