for v in exp_sf3 exp_sf2; do
  export S3D_LIB_PATH=$PWD/slam3d_amd/lib/$v.so
  echo "== $v"; NPAIRS=256 timeout 300 python tools_dev/r4.py 0 2>&1 | grep -A1 "^flags" | tail -2
done
