"""Per-kernel L2<->fabric traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each collected in its own
run with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes):

    python tools/pmc_traffic_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [steps]

Units and corrections (same guide): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of
wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics.
Infinity-Cache hits are counted, so the figures are an upper bound on HBM bytes.  Output: average bytes per launch for
every kernel (bench.py's GEMM variant names for the GEMM templates), and the whole-step totals."""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"gemm_kernel<(\d+), (\d+), (true|false), (true|false), (true|false)>", name)
    if m:
        bm, bn, ak, bk, bf = m.groups()
        lay = {("true", "true"): "nt", ("false", "false"): "tn", ("true", "false"): "nn", ("false", "true"): "tt"}[(ak, bk)]
        return f"gemm_{'bf16_' if bf == 'true' else ''}{lay}_{bm}x{bn}"
    m = re.search(r"gemm_small_kernel<(true|false), (true|false), \d+>", name)
    if m:
        lay = {("true", "true"): "nt", ("false", "false"): "tn", ("true", "false"): "nn", ("false", "true"): "tt"}[m.groups()]
        return f"gemm_small_{lay}"
    if "gg_tn_bf16_kernel" in name:
        return "gemm_group_tn_bf16"
    if "gg_tn_split" in name:                 # fp32 operands, products on the bf16 matrix pipe (both forms)
        return "gemm_group_tn_bf16x3"
    if "gr_split_kernel" in name:
        return "gemm_rows256_bf16x3"
    if "gr_bf16_kernel" in name:
        return "gemm_rows256_bf16"
    if "gr_kernel" in name:
        return "gemm_rows256"
    m = re.search(r"gemm_wide(_split)?_kernel<sdumc_wide::WideCfg<\d+, \d+, \d+, \d+, \d+, \d+, (true|false), (true|false)", name)
    if m:
        return "gemm_wide_" + ("nt" if m.group(2) == "true" else "tn") + ("_bf16x3" if m.group(1) else "")
    m = re.search(r"gemm_p3_nt_kernel<sdumc_p3::PCfg<\d+, \d+, (true|false)", name)
    if m:                                     # the names of bench.py's event variants (gemm_f32.hip kVariantName)
        return "gemm_p3_nt_masked_bf16x3" if m.group(1) == "true" else "gemm_p3_nt_bf16x3"
    if "gemm_b1_nt_kernel" in name:
        return "gemm_b1_nt_bf16"
    if "gg_tn_kernel" in name:
        return "gemm_group_tn"
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def load(path, counter):
    per = collections.defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return per


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else None
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_sha
    out = {"source_sha": source_sha(),
           "units": "bytes per launch (average over the launches of the profiled run)",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950: 128-byte requests tallied at 64 bytes); WRITE_SIZE as read",
           "caveat": "L2 <-> fabric requests: Infinity-Cache hits are included, so this bounds HBM bytes from above",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(k, []), write.get(k, [])
        fb = 2.0 * sum(fv) / max(1, len(fv))
        wb = sum(wv) / max(1, len(wv))
        out["kernels"][k] = {"launches": max(len(fv), len(wv)), "fetch_bytes": round(fb), "write_bytes": round(wb),
                             "traffic_bytes": round(fb + wb)}
    tot_f = 2.0 * sum(sum(v) for v in fetch.values())
    tot_w = sum(sum(v) for v in write.values())
    out["run_total"] = {"fetch_bytes": round(tot_f), "write_bytes": round(tot_w)}
    # launches of the profiled command that are NOT part of a step: installing the batch (torch copies, the feature-plane split
    # of set_batch -- once per batch, features do not change across epochs) and torch's own fills
    setup = lambda k: k.startswith("__amd_rocclr") or k.startswith("at::native") or k == "sdumc_p3::p3_split_kernel"
    set_f = 2.0 * sum(sum(v) for k, v in fetch.items() if setup(k))
    set_w = sum(sum(v) for k, v in write.items() if setup(k))
    out["setup_total"] = {"fetch_bytes": round(set_f), "write_bytes": round(set_w),
                          "kernels": sorted(k for k in set(fetch) | set(write) if setup(k))}
    if steps:
        out["per_step"] = {"steps_in_run": steps, "fetch_bytes": round((tot_f - set_f) / steps), "write_bytes": round((tot_w - set_w) / steps),
                           "note": "the step's own launches; batch installation (setup_total) excluded"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    top = sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches"])[:8]
    for k, v in top:
        print(f"{k:28s} n={v['launches']:4d} fetch {v['fetch_bytes'] / 1e6:8.1f} MB  write {v['write_bytes'] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main()
