"""HBM traffic per kernel launch from two rocprofv3 counter passes of tools/net_bench:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir1> --output-format csv -- tools/net_bench <rows> <classes> <epochs>
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir2> --output-format csv -- tools/net_bench <rows> <classes> <epochs>
    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <rows per launch> <out.json>
FETCH_SIZE and WRITE_SIZE are reported in kilobytes; on gfx950 FETCH_SIZE tallies 128-byte read
requests as 64 bytes, so it is doubled; WRITE_SIZE is taken as it is."""
import collections
import csv
import json
import re
import sys

GEMM_TAGS = ["fwd1", "fwd2", "fwd3", "delta2", "delta1", "dw3", "dw2", "dw1"]
OTHER = {"prep_rows_kernel": "prep", "head_mfma_kernel": "head", "head_wide_kernel": "head", "head_kernel": "head",
         "slab_reduce_kernel": "reduce", "transpose_kernel": "w2t", "apply_kernel": "apply", "split_planes_kernel": "w2split", "prep_rows4_kernel": "prep"}


def tag_of(name):
    m = re.search(r"gemm_kernel<\d+, \d+, \d+, (\d+),", name) or re.search(r"gemm_(?:x3|f32)w_kernel<\d+, \d+, \d+, (\d+)>", name)
    if m:
        return GEMM_TAGS[int(m.group(1))]
    for key, tag in OTHER.items():
        if key in name:
            return tag
    return None


def per_kernel(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        t = tag_of(r["Kernel_Name"])
        if t:
            tot[t] += float(r["Counter_Value"])
            n[t] += 1
    return {t: tot[t] / n[t] for t in tot}


def main():
    fetch_csv, write_csv, rows, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for t in f:
        raw = f[t] * 1024.0
        rd, wr = 2.0 * raw, w.get(t, 0.0) * 1024.0
        kernels[t] = {"fetch_size_raw_bytes": raw, "read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                      "hbm_bytes_per_row": (rd + wr) / rows}
    doc = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- tools/net_bench",
           "rows_per_launch": rows,
           "corrections": "FETCH_SIZE and WRITE_SIZE are reported in kilobytes; FETCH_SIZE doubled (gfx950 tallies 128-B read "
                          "requests as 64 B); WRITE_SIZE taken as is",
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    for t, k in kernels.items():
        print("%-8s read %8.1f MB  write %8.1f MB  per row %7.1f B" % (t, k["read_bytes"] / 1e6, k["write_bytes"] / 1e6, k["hbm_bytes_per_row"]))


if __name__ == "__main__":
    main()
