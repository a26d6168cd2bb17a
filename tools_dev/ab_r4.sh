export S3D_LIB_PATH=$PWD/slam3d_amd/lib/exp_k4w4.so
NPAIRS=256 timeout 300 python tools_dev/r4.py 0x10000000 2>&1 | grep "^flags" | tail -1
