"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid): usage: trace_summary.py <kernel_trace.csv> [steps]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    name = r['Kernel_Name']
    short = name.split('((anon')[0].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    short = short.split('(')[0][:44]
    key = (short, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Grid_Size_Y'], r['Grid_Size_Z'])
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[key][0] += d
    agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print("total kernel time %.2f ms per step (%d dispatches)" % (tot / 1e6 / steps, len(rows)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-44s blocks=%6s,%2s,%4s n/step=%6.1f  ms/step %7.2f  avg %8.1f us  (%4.1f%%)" %
          (k[0], k[1], k[2], k[3], v[1] / steps, v[0] / 1e6 / steps, v[0] / v[1] / 1e3, 100 * v[0] / tot))
