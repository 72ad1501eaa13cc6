"""Registers, scratch, LDS and occupancy of every kernel of libfdtd_hip, as the compiler
reports them (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed):

    python tools/kernel_resources.py [--unit launch_whole] [substring of the kernel name ...]

Units: launch_half, launch_whole (default), launch_adjoint.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit = "launch_whole"
if "--unit" in sys.argv:
    at = sys.argv.index("--unit")
    unit = sys.argv[at + 1]
    del sys.argv[at:at + 2]
src = os.path.join(ROOT, "fdtd_amd", "csrc", unit + ".hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
text = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
rows, cur = [], None
for line in text.splitlines():
    m = re.search(r"remark:\s+(.+?):\s+(\S+)(?:\s+\[-R.*)?$", line)
    if not m:
        continue
    key, val = m.group(1).strip(), m.group(2)
    if key == "Function Name":
        cur = {"name": val}
        rows.append(cur)
    elif cur is not None:
        cur[key] = val
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                       capture_output=True, text=True).stdout.splitlines()
want = sys.argv[1:]
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'occ':>4}  kernel")
for r, n in zip(rows, names):
    n = re.sub(r"\(fdtdhip::\w+<\w+>\)$", "", n.replace("void fdtdhip::", ""))
    if want and not all(w in n for w in want):
        continue
    print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('TotalSGPRs', r.get('SGPRs', '?')):>5} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>8} {r.get('LDS Size [bytes/block]', '?'):>7} "
          f"{r.get('Occupancy [waves/SIMD]', '?'):>4}  {n}")
