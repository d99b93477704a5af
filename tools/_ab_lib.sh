set -e
R=$(pwd)
cp $R/libstevi_amd/libstevi_hip.so /tmp/keep.so
cd /tmp; export TMPDIR=/tmp
for arm in A B A B; do
  cp $R/_ab/lib$arm.so $R/libstevi_amd/libstevi_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$arm -- python3 $R/tools/sweep_ab.py --rounds 1 --reps 100 census_tiles=1 > /tmp/ab_$arm.log 2>&1
  f=$(ls /tmp/abl_$arm/*/*kernel_stats.csv | head -1)
  echo arm $arm scan $(grep "scan_cols_kernel<true>" $f | awk -F, '{print $(NF-4)}') tile $(grep "tile_finalize" $f | awk -F, '{print $(NF-4)}') $(grep -o "checksum\": [0-9]*" /tmp/ab_$arm.log | head -1)
  rm -rf /tmp/abl_$arm
done
cp /tmp/keep.so $R/libstevi_amd/libstevi_hip.so
