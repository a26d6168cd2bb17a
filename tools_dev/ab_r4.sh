for v in 0 16 32 64 128; do
  export S3D_SORT_GROUP=$v
  for f in 0 0x800000; do
  echo "== group $v flags $f"; NPAIRS=256 timeout 300 python tools_dev/r4.py $f 2>&1 | grep "^flags" | tail -1
  done
done
