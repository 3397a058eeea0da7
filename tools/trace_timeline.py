"""Timeline statistics of a rocprofv3 --kernel-trace CSV of bench.py (multi-stream run): how busy the GPU is inside a step,
how much kernels overlap, and where the idle gaps are.  usage: trace_timeline.py <kernel_trace.csv> <steps in the trace>"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
print("columns:", list(rows[0].keys()))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows)
n = len(iv)
iv = iv[(steps - 1) * n // steps:]          # the last step
t0, t1 = iv[0][0], max(e for _, e, *_ in iv)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e, k, q, st in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, k[:50]))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, *_ in iv)
print("last step: span %.2f ms  busy(union) %.2f ms  idle %.2f ms  sum of kernel durations %.2f ms  mean concurrency %.2f  kernels %d"
      % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, tot / 1e6, tot / busy, len(iv)))
gaps.sort(reverse=True)
print("largest gaps (us, next kernel):", [(round(g / 1e3, 1), k) for g, k in gaps[:10]])
for thr in (1000, 5000, 20000):
    print("gaps > %d us: %d, total %.2f ms" % (thr // 1000, sum(1 for g, _ in gaps if g > thr), sum(g for g, _ in gaps if g > thr) / 1e6))
byq = collections.defaultdict(lambda: [0, 0])
for s, e, k, q, st in iv:
    byq[(q, st)][0] += e - s
    byq[(q, st)][1] += 1
for k, v in sorted(byq.items(), key=lambda kv: -kv[1][0]):
    print("queue/stream %s: %.2f ms in %d kernels" % (k, v[0] / 1e6, v[1]))
# time during which exactly one / two / three+ kernels run
ev = []
for s, e, *_ in iv:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
lvl, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[min(lvl, 4)] += t - last
    last = t
    lvl += d
print("time at concurrency level: " + "  ".join("%d: %.2f ms" % (k, v / 1e6) for k, v in sorted(hist.items())))
