#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, they cost 3 + 2).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pmc-calibrate
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pmc-calibrate
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/bench_pmc.json profiles/r02_pmc_traffic.json

Calibration (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x and WRITE_SIZE is
uncalibrated): bench.py --pmc-calibrate launches a device copy of exactly 2^30 B; the raw counter of that kernel gives the factor
bytes-per-count for 16 B/lane streaming accesses, which is what the row streams of k_eg_tile / k_build are.  Both the raw counter
means and the factors are written out so the correction can be audited."""
import collections, csv, glob, json, re, sys

KERNELS = {"eg_pass": r"k_eg_tile<", "eg_mr2": r"k_eg_tile_mr<2>", "eg_mr3": r"k_eg_tile_mr<3>", "halo_fold": r"k_halo_fold", "pcg_step": r"k_pcg_step3<1|k_pcg_step<1", "pcg_direction": r"k_pcg_dir3<|k_pcg_direction",
           "pcg_step_lad": r"k_pcg_step3_lad<1", "pcg_direction_lad": r"k_pcg_dir3_lad",
           "eg_pass_untiled": r"k_eg_jtjp|k_eg_pass<1>", "eg_pass_gradient": r"k_eg_pass<0>", "eg_pass_diag": r"k_eg_pass<2>", "eg_gradcol": r"k_eg_gradcol", "gather": r"k_gather<false, true>",
           "build": r"k_build<true", "cost": r"k_build<false", "observe": r"k_observe", "copy_1GiB": r"(elementwise|vectorized|copy).*"}
COPY_BYTES = float(1 << 30)


def load(directory, counter):
    rows = collections.defaultdict(list)
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows


def pick(rows, pattern, full_only=True):
    vals = []
    for k, v in rows.items():
        if re.search(pattern, k):
            vals += v
    if not vals:
        return None
    if full_only:       # PCG launches queued behind the convergence flag exit at once; keep the launches that did the work
        m = max(vals); vals = [x for x in vals if x > 0.5 * m]
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_dir, write_dir, bench_json, out = sys.argv[1:5]
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    # the calibration copy: the largest torch elementwise kernel
    def copy_counter(rows):
        best = None
        for k, v in rows.items():
            if "i3d::" in k or "rocprim" in k:
                continue
            m = max(v)
            if best is None or m > best:
                best = m
        return best
    cf, cw = copy_counter(fetch), copy_counter(write)
    f_read, f_write = COPY_BYTES / cf, COPY_BYTES / cw
    bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pmc-calibrate`",
           "calibration": {"copy_bytes": COPY_BYTES, "FETCH_SIZE_raw_of_copy": cf, "WRITE_SIZE_raw_of_copy": cw,
                           "bytes_per_FETCH_SIZE_count": f_read, "bytes_per_WRITE_SIZE_count": f_write,
                           "note": "factor measured on a 16 B/lane streaming copy of 2^30 B; applied to every kernel below"},
           "kernel_tag": bench.get("kernel_tag"), "eg_rows": bench["config"]["rows"]["Eg"], "active_voxels": bench["config"]["active_voxels"], "kernels": {}}
    for name, pat in KERNELS.items():
        if name == "copy_1GiB":
            continue
        a, b = pick(fetch, pat), pick(write, pat)
        if a is None or b is None:
            continue
        res["kernels"][name] = {"FETCH_SIZE_raw_mean": a[0], "WRITE_SIZE_raw_mean": b[0], "launches_counted": [a[1], b[1]],
                                "read_bytes_per_launch": a[0] * f_read, "write_bytes_per_launch": b[0] * f_write,
                                "traffic_bytes_per_launch": a[0] * f_read + b[0] * f_write}
    for k in ("eg_pass", "eg_mr2", "eg_mr3", "build"):
        if k in res["kernels"] and k in bench.get("kernels", {}):
            res["kernels"][k]["algorithmic_bytes_per_launch"] = bench["kernels"][k]["algorithmic_GB"] * 1e9
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
