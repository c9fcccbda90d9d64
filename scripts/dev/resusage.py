"""Resource usage of the kernels of one HIP source: name, VGPRs, spills, LDS, occupancy (one line per kernel).
usage: python scripts/dev/resusage.py sigsvgd_amd/csrc/gram_fast.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o",
                      "/tmp/resusage.o", "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
rows, cur = [], None
for l in out.stderr.splitlines():
    if "error" in l or "warning:" in l:
        print(l)
    m = re.search(r"remark:\s+(.*) \[-Rpass", l)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"sigsvgd::|\(.*\)|void ", "", n)
    g = lambda k: r.get(k, "?")
    print(f"{n:72s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>3s} spill {g('VGPRs Spill'):>4s} "
          f"sspill {g('SGPRs Spill'):>3s} lds {g('LDS Size [bytes/block]'):>6s} occ {g('Occupancy [waves/SIMD]')}")
