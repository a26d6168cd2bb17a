for v in default exp_NOSEARCH; do
  if [ $v = default ]; then unset S3D_LIB_PATH; else export S3D_LIB_PATH=$PWD/slam3d_amd/lib/$v.so; fi
  echo "== $v"; timeout 300 python tools_dev/r4.py 0 2>&1 | grep -A1 "^flags" | tail -2
done
