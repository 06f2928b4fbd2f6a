"""From a rocprofv3 kernel_trace.csv: GPU busy fraction, mean concurrency, and -- for the time when exactly ONE kernel
is running (the exposed, un-overlapped time) -- which kernels that is.  Diagnostic for the multi-stream step."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# window: from the end of the i-th to the end of the j-th launch of the step delimiter (the loss kernel: once per step)
delim, i0, i1 = (sys.argv[2] if len(sys.argv) > 2 else "loss_sample_kernel"), int(sys.argv[3]) if len(sys.argv) > 3 else 6, int(sys.argv[4]) if len(sys.argv) > 4 else 12
marks = sorted(int(r["End_Timestamp"]) for r in rows if delim in r["Kernel_Name"])
a, b = marks[i0], marks[i1]
print("steps in window:", i1 - i0, " ms per step: %.2f" % ((b - a) / 1e6 / (i1 - i0)))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= a or s >= b:
        continue
    name = r["Kernel_Name"].split("(")[0][:48]
    ev.append((max(s, a), 1, name)); ev.append((min(e, b), -1, name))
ev.sort(key=lambda x: (x[0], x[1]))
active = collections.Counter(); n = 0; last = a
busy = 0; conc = 0; solo = collections.Counter(); idle = 0
for t, d, name in ev:
    dt = t - last
    if dt > 0:
        if n == 0: idle += dt
        else:
            busy += dt; conc += n * dt
            if n == 1:
                solo[next(k for k, v in active.items() if v > 0)] += dt
    last = t
    active[name] += d; n += d
tot = b - a
print("window %.1f ms: busy %.1f%%, idle %.1f%%, mean concurrency while busy %.2f" % (tot / 1e6, 100 * busy / tot, 100 * idle / tot, conc / max(busy, 1)))
print("time with exactly one kernel running: %.1f%% of the window; by kernel:" % (100 * sum(solo.values()) / tot))
for k, v in solo.most_common(25):
    print("  %-50s %6.2f%%" % (k, 100 * v / tot))

# the largest idle gaps inside the window, with the kernels that end before / start after them
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows
            if int(r["End_Timestamp"]) > a and int(r["Start_Timestamp"]) < b)
gaps = []
cur_end, cur_name = iv[0][1], iv[0][2]
for s_, e_, nm in iv[1:]:
    if s_ > cur_end:
        gaps.append((s_ - cur_end, cur_name, nm, (cur_end - a) / 1e6))
    if e_ > cur_end:
        cur_end, cur_name = e_, nm
gaps.sort(reverse=True)
print("idle gaps: %d, total %.2f ms; > 20 us: %d (%.2f ms)" % (len(gaps), sum(g[0] for g in gaps) / 1e6,
      sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
by = collections.Counter()
for g in gaps:
    by[(g[1], g[2])] += g[0]
for (p_, n_), v in by.most_common(25):
    print("  %8.1f us total  after %-40s before %-40s" % (v / 1e3, p_, n_))
