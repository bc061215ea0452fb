# usage: bash tools/abl_bwd.sh "variant names" "stages"  -- kernel bench of build/variants/libmedscan_<name>.so (base = the in-tree library)
for v in ${1:-base}; do
  if [ $v = base ]; then unset MEDSCAN_LIBRARY; else export MEDSCAN_LIBRARY=$PWD/build/variants/libmedscan_$v.so; fi
  echo "== $v"; python tools/scan_kernel_bench.py 64 10 ${2:-0,2,3} 2>&1 | grep stage
done
