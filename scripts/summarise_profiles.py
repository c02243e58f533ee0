"""Turn the raw rocprofv3 output of scripts/collect_profiles.sh into the committed summaries (profiles/<tag>_*).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are collected in separate
--pmc passes, are in KB, and FETCH_SIZE is doubled on gfx950 (128-byte requests are counted as 64 B)."""
import csv, glob, json, os, shutil, sys

out, summ, tag = sys.argv[1:4]
STEPS_PROFILED = 3          # --steps 2 --warmup 1
KERNELS = ("tapconv_kernel", "tappatch_kernel", "skinny_kernel", "skinny_group_kernel")     # the contraction launches bench.py brackets with HIP events
LAUNCHES_PER_STEP = 116     # ResNet-50 forward + explanation: 54 forward + 62 input-gradient launches (stem gradient fused; the head gradient is a streaming launch since the end of round 5: 117 before)


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


for name in ("kernel_stats.csv", "domain_stats.csv"):
    f = find(os.path.join(out, "stats"), name)
    if f:
        shutil.copy(f, os.path.join(summ, f"{tag}_bench_{name}"))


def pmc_sum(d, counter):
    f = find(d, "counter_collection.csv")
    tot, launches = 0.0, 0
    if not f:
        return None, 0
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            if not any(k in row.get("Kernel_Name", "") for k in KERNELS):
                continue
            tot += float(row["Counter_Value"]); launches += 1
    return tot, launches


fetch, nf = pmc_sum(os.path.join(out, "fetch"), "FETCH_SIZE")
write, nw = pmc_sum(os.path.join(out, "write"), "WRITE_SIZE")
res = {"source": "BCOS_SUBBATCH_STREAMS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
       "note": "contraction kernels (tapconv_kernel<*>, tappatch_kernel<*>, skinny_kernel); the small calibration launches (8 images) are included "
               "in the sums and contribute < 3 % of the bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts "
               "128-B requests as 64 B), counters in KB",
       "fetch_size_kb_sum": fetch, "write_size_kb_sum": write, "steps": STEPS_PROFILED, "launches_per_step": LAUNCHES_PER_STEP,
       "launches_counted": {"fetch": nf, "write": nw}}
if fetch is not None and write is not None:
    per_step = (2.0 * fetch + write) * 1024.0 / STEPS_PROFILED
    res["hbm_bytes_per_step"] = per_step
    res["hbm_bytes_per_launch"] = per_step / LAUNCHES_PER_STEP
# per-launch duration of the contraction kernels inside the profiled steps (the first launches of the process belong
# to the 8-image calibration pass of bench.py and are dropped: everything before the last 3 * LAUNCHES_PER_STEP launches)
tr = find(os.path.join(out, "stats"), "kernel_trace.csv")
if tr:
    durs = []
    with open(tr) as fh:
        for row in csv.DictReader(fh):
            if any(k in row["Kernel_Name"] for k in KERNELS):
                durs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    durs.sort()
    steps = [d for _, d in durs][-STEPS_PROFILED * LAUNCHES_PER_STEP:]
    res["rocprof_kernel_trace"] = {"launches": len(steps), "avg_launch_us": sum(steps) / len(steps) / 1e3,
                                   "kernel_ms_per_step": sum(steps) / 1e6 / STEPS_PROFILED}
json.dump(res, open(os.path.join(summ, f"{tag}_hbm_traffic.json"), "w"), indent=1)
json.dump({"hbm_bytes_per_launch": int(res.get("hbm_bytes_per_launch", 0)) or None,
           "hbm_bytes_per_step": int(res.get("hbm_bytes_per_step", 0)) or None, "round": tag,
           "source": f"profiles/{tag}_hbm_traffic.json"}, open(os.path.join(summ, "traffic_latest.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
