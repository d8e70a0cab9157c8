#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known-byte kernels (two separate PMC passes).  usage (GPU box, repo root): scripts/pmc_calibrate.sh <out.txt>
out=${1:-gpurun_out/pmc_calibration.txt}
export TMPDIR=/tmp
cd /tmp
python /root/repo/scripts/pmc_calibrate.py > $OLDPWD/$out 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pcal_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pcal_$c -- python /root/repo/scripts/pmc_calibrate.py > /dev/null 2>&1
  echo "== $c (KiB per launch, mean)" >> $OLDPWD/$out
  python /root/repo/scripts/pmc_generic.py $(find /tmp/pcal_$c -name "*counter_collection.csv" | head -1) "" | cut -c1-200 >> $OLDPWD/$out
done
cd $OLDPWD
cat $out
