for n in 1 8 16 32 64; do
  echo "== npairs $n"; NPAIRS=$n timeout 300 python tools_dev/r4.py 0 0x100000 2>&1 | grep "^flags" | tail -2
done
