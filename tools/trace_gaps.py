import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows)
# take the last third of the trace (last step)
n = len(iv); iv = iv[2 * n // 3:]
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
busy = 0; cur_s, cur_e = iv[0][0], iv[0][1]
gaps = []
for s, e, k in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, k)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.2f ms  busy(union) %.2f ms  idle %.2f ms  kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(iv)))
gaps.sort(reverse=True)
print("largest gaps (us, next kernel):", [(round(g / 1e3, 1), k) for g, k in gaps[:12]])
print("gaps > 5us:", sum(1 for g, _ in gaps if g > 5000), "total in them %.2f ms" % (sum(g for g, _ in gaps if g > 5000) / 1e6))
