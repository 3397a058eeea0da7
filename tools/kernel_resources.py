"""Register / scratch / LDS usage of every kernel of a .hip file as the compiler reports it (no GPU needed):

    python tools/kernel_resources.py munit_amd/csrc/conv_wino.hip [extra hipcc flags]

One line per kernel: VGPRs, AGPRs, spilled VGPRs, scratch bytes per lane, LDS bytes per block, waves per SIMD."""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|Name): (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(2)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
print("%-64s %6s %6s %6s %8s %8s %6s" % ("kernel", "VGPR", "AGPR", "spill", "scratch", "LDS", "waves"))
for r in rows:
    print("%-64s %6d %6d %6d %8d %8d %6d" % (r["name"][:64], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1),
                                             r.get("ScratchSize", -1), r.get("LDS Size", -1), r.get("Occupancy", -1)))
