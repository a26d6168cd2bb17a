cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np
for i in range(4):
    np.load('tests/golden/cloud%d.npz' % (i+1))['xyzi'].astype(np.float32).tofile('/tmp/scan%d.bin' % i)
PY
./cpp/example_link_neighbors 0 /tmp/scan0.bin /tmp/scan1.bin /tmp/scan2.bin /tmp/scan3.bin | cut -c1-150 | tail -30
