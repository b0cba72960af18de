"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch per kernel.

    python tools/pmc_summarize.py <fetch_dir> <write_dir> <table_bytes> <out.json>

Method (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are reported in KiB; on
gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes and WRITE_SIZE is uncalibrated,
so both are calibrated here on a kernel of known traffic in the same run -- torch's copy of the
item table (table_bytes read, table_bytes written) -- and the factors applied to our kernels.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
    per = defaultdict(list)
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter:
                    per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return per


KEYS = ("neumf_step_kernel", "neumf_mark_owner_kernel", "neumf_mark_multi_kernel", "neumf_unmark_kernel", "neumf_reduce_partials_kernel",
        "bprmf_fwd_bwd_kernel", "plan_rows_kernel", "plan_bucket_kernel", "plan_flags_kernel", "plan_bitmap_kernel", "plan_bucket_hash_kernel", "plan_scatter_kernel", "plan_count_kernel",
        "plan_colscan_kernel", "plan_chunk_kernel", "plan_final_kernel", "seg_update_multi_x2_kernel", "seg_update_kernel", "long_chunk_kernel", "long_final_kernel",
        "long_plan_kernel", "segment_heads_kernel", "make_keys_kernel", "reduce_sum_kernel",
        "radix_sort_onesweep", "onesweep_histograms", "merge_sort", "block_sort")


def short(name):
    for key in KEYS:
        if key in name:
            tmpl = name[name.find(key) + len(key):].split("(")[0] if key.endswith("_kernel") else ""
            return key + tmpl
    return None


def top_mean(vals):
    """mean over the largest launches of a kernel (the item-phase ones: B*C occurrences)"""
    if not vals:
        return None
    m = max(vals)
    sel = [x for x in vals if x >= 0.5 * m]
    return sum(sel) / len(sel)


def calib(per, table_bytes):
    best = None
    for name, vals in per.items():
        if "copy" in name.lower():
            big = [v for v in vals if v * 1024 > table_bytes / 4]
            if big:
                best = (name, sum(big) / len(big))
    return best


def load_rows(dirname, counter):
    """[(dispatch id, kernel name, value)] in dispatch order"""
    rows = []
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter:
                    rows.append((int(row.get("Dispatch_Id") or len(rows)), row["Kernel_Name"], float(row["Counter_Value"])))
    return sorted(rows)


def step_total(dirname, counter, table_bytes, factor, n_steps):
    """bytes per step of EVERYTHING dispatched behind the last calibration-sized copy (the marker of tools/pmc_workload.py's
    whole-step mode), and the same per kernel"""
    rows = load_rows(dirname, counter)
    last = max((i for i, (_, name, v) in enumerate(rows) if "copy" in name.lower() and v * 1024 > table_bytes / 4), default=None)
    if last is None:
        return None, {}
    per = defaultdict(lambda: [0.0, 0])
    for _, name, v in rows[last + 1:]:
        k = name.split("(")[0].replace("void ", "")[:100]
        per[k][0] += v * 1024 * factor
        per[k][1] += 1
    total = sum(b for b, _ in per.values())
    return total / n_steps, {k: {"bytes_per_step": b / n_steps, "launches_per_step": c / n_steps} for k, (b, c) in per.items()}


def main():
    fetch_dir, write_dir, table_bytes, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    n_steps = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    cf, cw = calib(fetch, table_bytes), calib(write, table_bytes)
    kf = table_bytes / (cf[1] * 1024) if cf else 2.0
    kw = table_bytes / (cw[1] * 1024) if cw else 1.0
    res = {"_method": "HBM bytes = raw KiB * 1024 * calibration factor; factors from a streaming copy "
                      "of known size in the same run (gfx950: FETCH_SIZE under-counts wide reads 2x)",
           "_table_bytes": table_bytes,
           "_calibration": {"fetch_factor": kf, "write_factor": kw,
                            "fetch_kernel": cf[0][:80] if cf else None,
                            "write_kernel": cw[0][:80] if cw else None,
                            "fetch_raw_kib": cf[1] if cf else None, "write_raw_kib": cw[1] if cw else None}}
    agg = defaultdict(lambda: {"f": [], "w": []})
    for n in set(fetch) | set(write):
        s = short(n)
        if s is not None:
            agg[s]["f"] += fetch.get(n, [])
            agg[s]["w"] += write.get(n, [])
    for s, v in agg.items():
        fr, wr = top_mean(v["f"]), top_mean(v["w"])
        rb = fr * 1024 * kf if fr is not None else None
        wb = wr * 1024 * kw if wr is not None else None
        res[s] = {"fetch_raw_kib": fr, "write_raw_kib": wr, "hbm_read_bytes": rb, "hbm_write_bytes": wb,
                  "hbm_bytes_per_launch": (rb or 0) + (wb or 0), "launches_seen": len(v["f"])}
    if n_steps > 0:      # whole-step mode: everything behind the marker copy
        rt, rk = step_total(fetch_dir, "FETCH_SIZE", table_bytes, kf, n_steps)
        wt, wk = step_total(write_dir, "WRITE_SIZE", table_bytes, kw, n_steps)
        if rt is not None and wt is not None:
            kern = {}
            for k in set(rk) | set(wk):
                kern[k] = {"hbm_read_bytes_per_step": rk.get(k, {}).get("bytes_per_step", 0.0),
                           "hbm_write_bytes_per_step": wk.get(k, {}).get("bytes_per_step", 0.0),
                           "launches_per_step": max(rk.get(k, {}).get("launches_per_step", 0), wk.get(k, {}).get("launches_per_step", 0))}
            res["_step"] = {"steps": n_steps, "hbm_read_bytes_per_step": rt, "hbm_write_bytes_per_step": wt, "hbm_bytes_per_step": rt + wt,
                            "kernels": kern}
            print("whole step: %.1f MB read + %.1f MB written per step over %d steps" % (rt / 1e6, wt / 1e6, n_steps))
            for k, v in sorted(kern.items(), key=lambda kv: -(kv[1]["hbm_read_bytes_per_step"] + kv[1]["hbm_write_bytes_per_step"]))[:12]:
                print("   %-90s %8.2f MB  x %.1f" % (k[:90], (v["hbm_read_bytes_per_step"] + v["hbm_write_bytes_per_step"]) / 1e6, v["launches_per_step"]))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res["_calibration"]))
    for k in sorted(res):
        if not k.startswith("_"):
            print(k, {a: round(b / 1e6, 1) for a, b in res[k].items() if a.startswith("hbm") and b is not None})


if __name__ == "__main__":
    main()
