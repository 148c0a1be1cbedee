"""HBM traffic per launch of the hot kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) of `tools/bench_kernels.py all 3` -> profiles/r1_traffic.json.
    python tools/traffic_from_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> out.json
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B / lane) streaming reads -> doubled."""
import csv, glob, json, sys, collections
def mean_by_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = mean_by_kernel(sys.argv[1], "FETCH_SIZE"), mean_by_kernel(sys.argv[2], "WRITE_SIZE")
B, P, F_ = 32, 64, 128
act = B * P * P * F_ * 2
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python3 tools/bench_kernels.py all 3 "
                 "(MI355X); FETCH_SIZE/WRITE_SIZE are KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide 16-B/lane streaming reads)"}
def entry(match, alg, label):
    ks = [k for k in fe if match in k]
    if not ks: return
    k = ks[0]
    out[label] = {"FETCH_SIZE_KB": fe[k], "WRITE_SIZE_KB": wr.get(k, 0.0), "hbm_bytes_corrected": int(2 * fe[k] * 1024 + wr.get(k, 0.0) * 1024)}
    if alg: out[label]["algorithmic_bytes"] = alg
entry("conv3x3_patch_kernelIDF16bLi128ELi2ELi2ELi2ELi2ELb1", 2 * act + 128 * 128 * 9 * 2, "conv3x3_patch_kernel<bf16,128,2,2,2,2> B=32 64x64 128->128")
entry("conv_wgrad3_kernel<2, 2, 1, 2>", 2 * act + 128 * 128 * 9 * 4, "conv_wgrad3_kernel<2,2,1,2> same shape, mean over splits 40..160")
entry("wgrad_reduce_kernel", None, "wgrad_reduce_kernel same shape")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
