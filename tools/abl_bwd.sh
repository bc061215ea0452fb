for v in base NOBAR NOSWEEP NOSTATE NOCHAN NOSTORE NOFLUSH; do
  if [ $v = base ]; then unset MEDSCAN_LIBRARY; else export MEDSCAN_LIBRARY=$PWD/build/variants/libmedscan_abl_$v.so; fi
  echo "== $v"; python tools/scan_kernel_bench.py 64 10 0,2,3 2>&1 | grep stage
done
