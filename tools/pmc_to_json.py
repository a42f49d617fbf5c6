#!/usr/bin/env python3
"""Fold the rocprofv3 CSVs of tools/collect_roofline.sh into profiles/<tag>_pmc_<cfg>.json: per kernel of the
step, averaged over the LAST `steps` steps of the run (the state bench.py measures), duration from the
kernel trace and HBM traffic / SQ counters from their own --pmc passes.
  python tools/pmc_to_json.py <dir with ks/ fetch/ write/ sq/> <cfg> <steps> <out.json>"""
import collections, csv, glob, json, os, sys

root, cfg, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def rows(sub, pattern):
    r = []
    for f in glob.glob(os.path.join(root, sub, "**", pattern), recursive=True):
        r += list(csv.DictReader(open(f)))
    return r


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    return n


def per_kernel_last(rs, value_of, key="Kernel_Name", order="Dispatch_Id"):
    """{kernel: {counter: mean over the dispatches that belong to the last `steps` steps}}"""
    by = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in rs:
        by[short(r[key])][int(r[order])].update(value_of(r))
    res = {}
    for k, disp in by.items():
        ids = sorted(disp)
        res[k] = (ids, disp)
    return res


# launches per step of each kernel: from the trace of the last steps (kernels of the single-step path)
trace = rows("ks", "*kernel_trace.csv")
trace.sort(key=lambda r: int(r["Dispatch_Id"]))
names = [short(r["Kernel_Name"]) for r in trace]
# the last step ends with k_build_end (single step); walk back `steps` of them
ends = [i for i, n in enumerate(names) if n == "k_build_end"]
if len(ends) < 3:
    raise SystemExit(f"only {len(ends)} single steps in the trace")
if len(ends) <= steps:
    steps = len(ends) - 1  # (the set-up phase may consist of bulk steps only: average over the single steps there are)
first = ends[-steps - 1] + 1
tail = trace[first:]
per_step = collections.Counter(short(r["Kernel_Name"]) for r in tail)
dur = collections.defaultdict(float)
for r in tail:
    dur[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span_us = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3
kernels = {}
for k, cnt in per_step.items():
    kernels[k] = dict(launches_per_step=cnt / steps, avg_us=dur[k] / cnt, us_per_step=dur[k] / steps)


def fold(sub, counters):
    rs = rows(sub, "*counter_collection.csv")
    if not rs:
        return
    by = collections.defaultdict(lambda: collections.defaultdict(dict))
    meta = {}
    for r in rs:
        k = short(r["Kernel_Name"])
        by[k][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        meta[k] = dict(vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]), workgroup=int(r["Workgroup_Size"]), grid=int(r["Grid_Size"]))
    for k, disp in by.items():
        if k not in kernels:
            continue
        n = int(round(kernels[k]["launches_per_step"] * steps))
        ids = sorted(disp)[-n:] if n else []
        for c in counters:
            vals = [disp[i][c] for i in ids if c in disp[i]]
            if vals:
                kernels[k][c] = sum(vals) / len(vals)
        kernels[k].update(meta[k])


fold("fetch", ["FETCH_SIZE"])
fold("write", ["WRITE_SIZE"])
fold("sq", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"])
for k, v in kernels.items():
    if "FETCH_SIZE" in v:
        f, w = v["FETCH_SIZE"] * 1024, v.get("WRITE_SIZE", 0.0) * 1024
        v["hbm_bytes_raw_per_launch"] = f + w
        v["hbm_bytes_upper_per_launch"] = 2 * f + w  # gfx950: 128-B requests tallied at 64 B (MI355X_MICROARCH.md)
        v["hbm_gbs_raw"] = (f + w) / (v["avg_us"] * 1e-6) / 1e9
        v["hbm_gbs_upper"] = (2 * f + w) / (v["avg_us"] * 1e-6) / 1e9
    if "SQ_WAVE_CYCLES" in v and v.get("SQ_WAVE_CYCLES"):
        v["wait_any_frac"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
        v["issue_stall_frac"] = v.get("SQ_WAIT_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
import hashlib
_h = hashlib.sha256()
_d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "megalania_amd", "csrc")
for _f in sorted(os.listdir(_d)):
    _h.update(_f.encode() + b"\0" + open(os.path.join(_d, _f), "rb").read())
json.dump(dict(config=cfg, steps_averaged=steps, span_us_per_step=span_us / steps, source_hash=_h.hexdigest()[:16],
               note="averages over the last steps of tools/run_state.py (single steps on the evolved slab); FETCH_SIZE / WRITE_SIZE in KB; "
                    "SQ_* cycle counters count quad-cycles (MI355X_MICROARCH.md); *_upper doubles the fetch bytes (gfx950 correction) as an upper bound",
               kernels=dict(sorted(kernels.items(), key=lambda kv: -kv[1]["us_per_step"]))), open(out, "w"), indent=1)
print(open(out).read()[:3000])
