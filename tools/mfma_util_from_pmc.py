"""MFMA utilisation of the hot kernels from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass of tools/bench_kernels.py.
SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the chip's 1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16: 37 748 736 for the
38.65 GFLOP conv = exactly its 1 179 648 MFMAs); GRBM_GUI_ACTIVE = cycles the kernel was resident, summed over the 8 XCDs."""
import csv, glob, json, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    if "conv3x3_patch_kernelIDF16bLi128ELi2ELi2ELi2ELi2ELb1" in k or "conv_wgrad3_kernel<2, 2, 1, 2>" in k:
        m = {n: sum(v) / len(v) for n, v in c.items()}
        d = sum(dur[k]) / len(dur[k])
        e = {"mean_ns_under_pmc": d, **m}
        if "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            e["cycles_per_launch"] = cyc
            e["kernel_clock_GHz"] = cyc / d
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m: e["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        out[k[:70]] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
