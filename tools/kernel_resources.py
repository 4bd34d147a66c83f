#!/usr/bin/env python3
"""Register / LDS / scratch use and occupancy of every kernel of the library as hipcc reports them for gfx950 (no GPU needed):
    python3 tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
Same flags as weiner_slamit_v2_amd/build.py; one line per kernel."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXTRA = {"ba_kernels": ["-ffp-contract=fast"], "pose": ["-ffp-contract=fast"], "hamming": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
KEYS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"]
print("%-12s %-64s %5s %5s %5s %8s %7s %7s %8s %6s" % ("file", "kernel", "vgpr", "agpr", "sgpr", "scratch", "spill_v", "spill_s", "lds_B", "waves"))
for f in sorted(glob.glob(os.path.join(ROOT, "weiner_slamit_v2_amd", "csrc", "*.hip"))):
    b = os.path.basename(f)[:-4]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
           "-Wno-unused-value"] + EXTRA.get(b, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur, rec, rows = None, {}, []
    for line in err.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            if cur:
                rows.append((cur, rec))
            cur, rec = m.group(1), {}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur:
            rec[m.group(1).strip()] = m.group(2)
    if cur:
        rows.append((cur, rec))
    for name, rec in rows:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        dem = re.sub(r"\(.*", "", dem)
        print("%-12s %-64s %5s %5s %5s %8s %7s %7s %8s %6s" % ((b, dem[:64]) + tuple(rec.get(k, "?") for k in KEYS)))
