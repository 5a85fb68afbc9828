"""Compact kernel-resource table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kres.py <file.hip> [extra hipcc flags...]     prints: kernel  VGPR AGPR SGPR scratch occupancy LDS"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    print(f"{r['name'][:90]:90s} V{r.get('VGPRs', '?'):>4s} A{r.get('AGPRs', '?'):>4s} S{r.get('TotalSGPRs', '?'):>4s} scr{r.get('ScratchSize [bytes/lane]', '?'):>5s} "
          f"occ{r.get('Occupancy [waves/SIMD]', '?'):>2s} lds{r.get('LDS Size [bytes/block]', '?'):>6s}")
