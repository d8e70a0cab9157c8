#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known-byte kernels (two separate PMC passes).  usage (GPU box, repo root): scripts/pmc_calibrate.sh <out.txt>
out=${1:-gpurun_out/pmc_calibration.txt}
export TMPDIR=/tmp
cd /tmp
python /root/repo/scripts/pmc_calibrate.py > $OLDPWD/$out 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_DRAM"; do
  i=$((i+1))
  rm -rf /tmp/pcal_$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pcal_$i -- python /root/repo/scripts/pmc_calibrate.py > /dev/null 2>&1
  echo "== $c (FETCH/WRITE_SIZE in KiB, TCC_* in requests; per launch, mean)" >> $OLDPWD/$out
  python /root/repo/scripts/pmc_generic.py $(find /tmp/pcal_$i -name "*counter_collection.csv" | head -1) "" | cut -c1-260 >> $OLDPWD/$out
done
cd $OLDPWD
cat $out
