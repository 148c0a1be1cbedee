"""Wave-state breakdown of the dominant kernels from two rocprofv3 PMC passes (SQ counters; quad-cycle units except MFMA_BUSY):
WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES.

    python tools/wave_state.py PASS1/.../counter_collection.csv PASS2/.../counter_collection.csv > profiles/r2_wave_state.json
"""
import csv, sys, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3x3_patch_kernelIDF16bLi128ELi2ELi2ELi2ELi2ELb1" in k or "conv_wgrad3_kernel<2, 2, 1, 2" in k or "wgrad_reduce_fast_kernel<9" in k:
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    e = {"counters_mean_per_launch": m}
    if wc > 0:
        e["fraction_of_wave_cycles"] = {n[3:]: round(m[n] / wc, 4) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS",
                                                                               "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA") if n in m}
    if m.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        e["lds_bank_conflict_fraction"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    out[k] = e
print(json.dumps(out, indent=1))
