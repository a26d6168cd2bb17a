for v in default exp_ab0 default exp_ab0; do
  if [ $v = default ]; then unset S3D_LIB_PATH; else export S3D_LIB_PATH=$PWD/slam3d_amd/lib/$v.so; fi
  echo "== $v"; NPAIRS=256 SINGLE=1 timeout 300 python tools_dev/r4.py 0 2>&1 | grep -A1 "^flags\|single" | tail -3
done
