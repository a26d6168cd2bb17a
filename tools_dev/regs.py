"""dev tool: register / scratch / LDS use per kernel from a `hipcc --cuda-device-only -S` listing.
   usage: python tools_dev/regs.py [listing.s] [name-filter]   (builds the listing of s3d_api.hip when none is given)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".s") else None
flt = [a for a in sys.argv[1:] if not a.endswith(".s")]
if path is None:
    path = "/tmp/s3d_regs.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                           "--cuda-device-only", "-S", "-o", path, "s3d_api.hip"], cwd=os.path.join(ROOT, "slam3d_amd/csrc"),
                          stderr=subprocess.DEVNULL)
text = open(path).read()
for blk in text.split("  - .agpr_count:")[1:]:
    f = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
    name = subprocess.run(["c++filt", f.get("name", "?")], capture_output=True, text=True).stdout.split("(")[0].strip()
    if flt and not any(x in name for x in flt):
        continue
    print("%-48s vgpr %3s agpr %3s sgpr %3s spill %3s scratch %4s lds %6s" % (
        name[-48:], f.get("vgpr_count"), blk.split()[0], f.get("sgpr_count"), f.get("vgpr_spill_count"),
        f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size")))
