"""From a rocprofv3 kernel_trace.csv of bench.py: the kernels around the optimizer launch of one steady-state step (the tail of
backward, the all-reduce if any, Adam, the head of the next step) with start / end relative to Adam's start, per queue -- where
the time between the last backward kernel and the next step's first kernel goes (forced world-1 all-reduce)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
before, after = (int(sys.argv[3]) if len(sys.argv) > 3 else 14), (int(sys.argv[4]) if len(sys.argv) > 4 else 10)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_flat_kernel" in e[2]]
print("steps traced:", len(adam))
# ms per step from Adam to Adam
if len(adam) > step + 1:
    print("ms per step (adam to adam, steps %d..%d): %.3f" % (step, len(adam) - 1, (ev[adam[-1]][0] - ev[adam[step]][0]) / 1e6 / (len(adam) - 1 - step)))
i = adam[step]
t0 = ev[i][0]
print("%10s %10s %8s  q    kernel" % ("start us", "end us", "dur us"))
for e in ev[max(0, i - before):i + after + 1]:
    print("%10.1f %10.1f %8.1f  %-4s %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[3], e[2]))
# the idle time (no kernel running) in the window [-500 us, +300 us] around Adam's start, per step, averaged over the steady steps
import statistics
idles = []
for s in adam[step:-1]:
    a, b = ev[s][0] - 500000, ev[s][0] + 300000
    iv = sorted((max(x[0], a), min(x[1], b)) for x in ev if x[1] > a and x[0] < b)
    cur = a; idle = 0
    for s_, e_ in iv:
        if s_ > cur: idle += s_ - cur
        cur = max(cur, e_)
    idle += max(0, b - cur)
    idles.append(idle / 1e3)
print("idle us inside [-500, +300] us around Adam's start: median %.1f, min %.1f, max %.1f" % (statistics.median(idles), min(idles), max(idles)))
