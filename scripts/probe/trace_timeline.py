"""Timeline of a rocprofv3 --kernel-trace run of bench.py (round 5): how much of a step has 0 / 1 / 2+ kernels executing, per queue idle
gaps, and the kernel-time totals by name -- is the step bound by what its launches need or by gaps between them?"""
import csv, glob, sys, collections
root = sys.argv[1]
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
print(len(rows), "kernel records")
# steps are delimited by prep_input launches (2 per step with two sub-batch streams)
starts = [s for s, e, n, q in rows if "prep_input" in n]
if len(starts) < 8:
    print("no step markers"); sys.exit(0)
# take the middle half of the run
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 2       # prep_input launches per step = sub-batch streams
nsteps = len(starts) // per_step
lo, hi = starts[per_step * (nsteps // 2)], starts[per_step * (nsteps - 2)]
n_mid = (nsteps - 2) - nsteps // 2
sel = [(s, e, n, q) for s, e, n, q in rows if s >= lo and s < hi]
ev = []
for s, e, n, q in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, lo, collections.Counter()
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += d
tot = hi - lo
print(f"{n_mid} steps, {tot / n_mid / 1e6:.3f} ms per step; time with k kernels executing: " + ", ".join(f"k={k}: {hist[k] / n_mid / 1e6:.3f} ms" for k in sorted(hist)))
byname = collections.Counter(); cnt = collections.Counter()
for s, e, n, q in sel:
    import re
    m_ = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?([A-Za-z0-9_:<>, ]+)", n)
    key = (m_.group(1) if m_ else n)[:70]
    byname[key] += e - s; cnt[key] += 1
print(f"sum of kernel durations per step: {sum(byname.values()) / n_mid / 1e6:.3f} ms")
for k, v in byname.most_common(25):
    print(f"  {v / n_mid / 1e6:8.3f} ms  {cnt[k] / n_mid:6.1f} x  {k}")
# per-queue idle gaps
byq = collections.defaultdict(list)
for s, e, n, q in sel:
    byq[q].append((s, e))
for q, lst in byq.items():
    lst.sort()
    gaps = [b[0] - a[1] for a, b in zip(lst, lst[1:]) if b[0] > a[1]]
    busy = sum(e - s for s, e in lst)
    print(f"queue {q}: {len(lst) / n_mid:.0f} launches/step, busy {busy / n_mid / 1e6:.3f} ms/step, gaps {sum(gaps) / n_mid / 1e6:.3f} ms/step (median gap {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us)")
