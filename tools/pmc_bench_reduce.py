#!/usr/bin/env python3
"""Reduce the rocprofv3 passes of tools/pmc_bench.sh: per-frame counter totals of bench.py's TIMED trace_paths_kernel launches."""
import collections, csv, glob, hashlib, json, os, sys

out_dir, command = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "trace_paths_kernel<0,"            # <0, true> / <0, false>: INSTR 0 = no counters; short / general reciprocal forms


def _normalised_source(path):
    """Kernel source without // comments, trailing blanks and empty lines: a comment edit must not make the counter file look stale."""
    out = []
    for line in open(path, "r", encoding="utf-8", errors="replace"):
        code = line.split("//", 1)[0].rstrip()
        if code:
            out.append(code)
    return "\n".join(out).encode()


def source_tag():
    h = hashlib.sha256()
    for f in ("pt_megakernel.hip", "pt_megakernel_loop.inc", "pt_device.h", "pt_kernels.h"):
        h.update(_normalised_source(os.path.join(ROOT, "raytracer-public_amd", "csrc", f)))
    return h.hexdigest()[:16]


per_frame, launches_seen = collections.OrderedDict(), {}
builder_busy_all = None
steps = None
for p in sorted(glob.glob(os.path.join(out_dir, "pass*/"))):
    n = os.path.basename(os.path.dirname(p))[4:]
    log = json.load(open(os.path.join(out_dir, "launch_log_pass%s.json" % n)))
    steps = log["steps"]
    timed = [k for k, (tag, nf) in enumerate(log["launches"]) if tag.startswith("timed")]      # every repetition of the timed region ("timed", "timed-rep1", ...): the same K frames each
    reps = max(1, len({tag for tag, nf in log["launches"] if tag.startswith("timed")}))
    rows = collections.OrderedDict()          # counter -> {dispatch id: value}; dispatch ids grow in submission order
    for f in glob.glob(p + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if KERNEL not in row.get("Kernel_Name", ""):
                continue
            rows.setdefault(row["Counter_Name"], {})[int(row["Dispatch_Id"])] = float(row["Counter_Value"])
    for name, by_id in rows.items():
        vals = [by_id[k] for k in sorted(by_id)]
        if len(vals) != len(log["launches"]):
            print("pass %s: %d dispatches of %s, launch log has %d -- skipped" % (n, len(vals), KERNEL, len(log["launches"])))
            continue
        per_frame[name] = sum(vals[k] for k in timed) / steps / reps             # mean over the repetitions
        launches_seen[name] = len(timed) // reps
stats = {}
for f in glob.glob(os.path.join(out_dir, "trace/**/*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "trace_paths_kernel" in row["Name"] or "resolve_kernel" in row["Name"]:
            stats[row["Name"]] = {k: row[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs") if k in row}
    os.system("cp %s %s" % (f, os.path.join(out_dir, "kernel_stats.csv")))
# kernel busy time per frame in THIS session (the box the counters were taken on): union of the timed trace_paths_kernel dispatches
# of the --kernel-trace run, picked by position from that run's own launch log
builder_busy = None
try:
    log = json.load(open(os.path.join(out_dir, "launch_log_trace.json")))
    spans = []
    for f in glob.glob(os.path.join(out_dir, "trace/**/*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERNEL in row.get("Kernel_Name", ""):
                spans.append((int(row["Dispatch_Id"]), int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    spans.sort()
    if len(spans) == len(log["launches"]):
        def union_ns(iv):
            total, cur_a, cur_b = 0, None, None
            for a, b in sorted(iv):
                if cur_b is None or a > cur_b:
                    if cur_b is not None:
                        total += cur_b - cur_a
                    cur_a, cur_b = a, b
                else:
                    cur_b = max(cur_b, b)
            if cur_b is not None:
                total += cur_b - cur_a
            return total
        per_rep = collections.OrderedDict()            # busy time of every repetition of the timed region; the MEDIAN one is what bench.py reports as well
        for k, (_, a, b) in enumerate(spans):
            tag = log["launches"][k][0]
            if tag.startswith("timed"):
                per_rep.setdefault(tag, []).append((a, b))
        busy = sorted(union_ns(iv) / 1e6 / log["steps"] for iv in per_rep.values())
        builder_busy = busy[(len(busy) - 1) // 2]
        builder_busy_all = busy
    else:
        print("kernel trace: %d dispatches of %s, launch log has %d -- no builder-side busy time" % (len(spans), KERNEL, len(log["launches"])))
except Exception as e:
    print("no builder-side busy time:", e)
res = {"command": command, "builder_kernel_busy_ms_per_frame": builder_busy, "builder_kernel_busy_ms_per_frame_all_reps": builder_busy_all, "source_tag": source_tag(), "kernel": KERNEL, "steps": steps,
       "frames_per_launch": steps / max(launches_seen.get("SQ_INSTS_VALU", 1), 1), "timed_launches": launches_seen.get("SQ_INSTS_VALU"),
       "per_frame": per_frame, "kernel_stats": stats,
       "how": "tools/pmc_bench.sh: one rocprofv3 --pmc pass per counter group over the command above; the timed launches are identified by "
              "position from bench.py's own launch log; SQ_* cycle counters are in units of 4 cycles; FETCH_SIZE / WRITE_SIZE in KB"}
json.dump(res, open(os.path.join(out_dir, "pmc_bench.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
