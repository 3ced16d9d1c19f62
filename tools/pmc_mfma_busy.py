"""Matrix-pipe busy fraction per kernel from one rocprofv3 counter pass of tools/net_bench:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d <dir> --output-format csv -- tools/net_bench <rows> <classes> <epochs>
    python tools/pmc_mfma_busy.py <counter_collection.csv> <out.json>
GRBM_GUI_ACTIVE is summed over the 8 XCDs (clock = GRBM_GUI_ACTIVE / 8 / duration);
SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs."""
import collections
import csv
import json
import sys

from pmc_traffic import tag_of


def main():
    path, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        t = tag_of(r["Kernel_Name"])
        if t:
            acc[t][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[t]["_dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    kernels = {}
    for t, c in acc.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8.0
        dur = sum(c["_dur"]) / len(c["_dur"])
        kernels[t] = {"duration_us_under_counters": dur / 1e3, "clock_GHz": gui / dur, "mfma_busy_frac": busy / (gui * 1024.0)}
    json.dump({"note": __doc__, "kernels": kernels}, open(out, "w"), indent=1)
    for t, k in kernels.items():
        print("%-8s %8.1f us  %.3f GHz  MFMA busy %.3f" % (t, k["duration_us_under_counters"], k["clock_GHz"], k["mfma_busy_frac"]))


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
