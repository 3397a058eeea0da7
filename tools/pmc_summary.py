"""Summarise the three rocprofv3 --pmc passes of bench.py into profiles/<tag>_pmc_hbm_mfma.txt.
  python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <mfma_counter_collection.csv> <out.txt>
FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md); KB -> MB per launch."""
import collections, csv, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # lib_fingerprint(): ties this summary to the kernel sources it was collected on (bench.py checks it)


def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = (re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0], r["Grid_Size"], r["Workgroup_Size"])
        a = agg[k]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
m = load(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES")
g = load(sys.argv[3], "GRBM_GUI_ACTIVE")
out = ["# kernel-source fingerprint: " + bench.lib_fingerprint(),
       "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (three separate passes) of",
       "`MUNIT_NO_SIDE_STREAM=1 MUNIT_NO_BRANCH_STREAMS=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline`, MI355X",
       "FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced stream); KB -> MB, per launch.",
       "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCD * 1024 SIMD); clock (MHz) = GRBM_GUI_ACTIVE / 8 / duration.",
       "NOTE: the PMC passes run at pinned profiling clocks; the bare run is throttled under matrix load, see DESIGN.md.", ""]
for k in sorted(f, key=lambda k: -f[k][2])[:int(os.environ.get("PMC_ROWS", "18"))]:
    n, fs, dur = f[k]
    ws, ms, gs = w.get(k, [1, 0, 0]), m.get(k, [1, 0, 0]), g.get(k, [1, 1, 1])
    util = ms[1] / ms[0] / (gs[1] / gs[0] / 8 * 1024) * 100 if gs[1] else 0
    clk = gs[1] / gs[0] / 8 / (gs[2] / gs[0] * 1e3) * 1e3 if gs[2] else 0
    out.append("%-40s blocks=%6d launches=%4d  fetch %8.1f MB  write %7.1f MB  avg %7.1f us  MfmaUtil %5.1f %%  clock %4.0f MHz"
               % (k[0][:40], int(k[1]) // int(k[2]), n, 2 * fs / n / 1024, ws[1] / max(ws[0], 1) / 1024, dur / n, util, clk))
open(sys.argv[4], "w").write("\n".join(out) + "\n")
print("\n".join(out))
