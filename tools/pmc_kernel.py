"""Per-kernel averages of a rocprofv3 --pmc pass: python tools/pmc_kernel.py <counter_collection.csv> <name substring>
Prints, per distinct grid size of the matching kernels, the mean of every counter, the clock (GRBM_GUI_ACTIVE over
the 8 XCDs / duration) and the matrix pipe's busy fraction (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs)."""
import collections, csv, sys
path, key = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    if key not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][:90], r["Grid_Size"], r["Workgroup_Size"])
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[k]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in acc.items():
    print(k)
    m = {n: sum(v) / len(v) for n, v in c.items()}
    for n, v in sorted(m.items()):
        print("    %-32s %16.0f  (n=%d)" % (n, v, len(c[n])))
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        print("    clock %.3f GHz" % (cyc / m["_dur_ns"]))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            print("    MFMA busy %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)))
