#!/bin/bash
# A/B of library builds on the default batch: tools_dev/ab_libs.sh <out> lib1.so lib2.so ...   (BLOCKS, PAIRS as for ab_accum.py)
cd "$GRAFT_REPO_ROOT" || exit 1
out=$1; shift
mkdir -p "$(dirname "$out")"; : > "$out"
for lib in "$@"; do
  echo "== $lib" >> "$out"
  S3D_LIB_PATH="$PWD/$lib" python3 tools_dev/ab_accum.py >> "$out" 2>&1
done
cat "$out"
