"""Summaries of tools/profile_r6.sh's (earlier rounds: profile_r2.sh ... profile_r5.sh) rocprofv3 output -> small csv / json files for profiles/ (run on the GPU
box, right after).   python tools/profile_summary.py <raw dir> <out dir> [prefix = r3]"""
import collections, csv, glob, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
PRE = sys.argv[3] if len(sys.argv) > 3 else "r3"
os.makedirs(dst, exist_ok=True)


def find(d, suffix):
    fs = glob.glob(os.path.join(src, d, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None


def short(n):
    return n.replace("void pwr::", "").replace("_ZN3pwr", "pwr::")[:150]


# 1. kernel statistics (step, isolated kernels, decoder, dist)
for d in ("step", "roof", "iso", "iso24", "dec", "dist", "serial", "trainsteps"):
    f = find(d, "kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(dst, "%s_rocprofv3_%s_kernel_stats.csv" % (PRE, d)), "w") as o:
            w = csv.writer(o)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        print(d, "stats:", len(rows), "kernels; top:", [(short(r["Name"])[:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1)) for r in rows[:4]])


# 2. PMC passes -> bytes per launch
def mean_by_kernel(d, counter):
    f = find(d, "counter_collection.csv")
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[(r["Kernel_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


B, P, F_ = 32, 64, 128
act = B * P * P * F_ * 2
out = {"source": "tools/profile_r%s.sh on MI355X: rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE; both counters are KB; "
                 "FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide streaming reads at 64 B, MI355X_MICROARCH.md section HBM)" % PRE[1:]}
fe, wr = mean_by_kernel("pmc_fetch", "FETCH_SIZE"), mean_by_kernel("pmc_write", "WRITE_SIZE")
fe.update(mean_by_kernel("pmc_fetch24", "FETCH_SIZE")); wr.update(mean_by_kernel("pmc_write24", "WRITE_SIZE"))


MISSING = []      # named kernels without a counter row although their PMC pass ran: reported and the script exits non-zero


def entry(table_f, table_w, match, grid, alg, label, required=True):
    """One row of <PRE>_traffic.json.  `match` is a PREFIX-free substring of the kernel name that must pin down ONE kernel (give template
    arguments up to a comma, e.g. "conv3x3_wstat_kernel<true, 0," -- a further template parameter then still matches); `grid` selects among
    launches of that kernel at several grid sizes.  Several (kernel, grid) keys matching with grid = None is an ERROR (round 5 averaged
    whichever came first: the 80-split reduce against round 4's 24-split one), so is no match when the pass has data at all."""
    grids = None if grid is None else [str(g_) for g_ in (grid if isinstance(grid, (list, tuple)) else [grid])]
    ks = sorted(k for k in table_f if match in k[0] and (grids is None or k[1] in grids))
    if not ks:
        if table_f and required:
            MISSING.append("%s (match %r, grid %s): no such kernel in the FETCH_SIZE pass" % (label, match, grid))
        return
    if len(ks) > 1:
        MISSING.append("%s (match %r): %d (kernel, grid) keys match -- give a grid: %s" % (label, match, len(ks), [(short(k[0])[:60], k[1]) for k in ks]))
        return
    k = ks[0]
    w = table_w.get(k, 0.0)
    out[label] = {"kernel": short(k[0]), "grid_size": k[1], "FETCH_SIZE_KB": table_f[k], "WRITE_SIZE_KB": w,
                  "hbm_bytes_corrected": int(2 * table_f[k] * 1024 + w * 1024)}
    if alg:
        out[label]["algorithmic_bytes"] = alg
        out[label]["ratio"] = out[label]["hbm_bytes_corrected"] / alg


wts = 128 * 128 * 9 * 2
# (tools/bench_kernels.py launches the patch kernel as the DATA GRADIENT with the norm-backward sums: it reads dy AND the forward activations y
# and writes dx -- three activation tensors, not two.  Round 5's file priced it at two and showed "1.59 x"; round 4's 1.05 x was the forward form.)
entry(fe, wr, "conv3x3_patch_kernelIDF16bLi128ELi2ELi2ELi2ELi2ELb1", None, 3 * act + wts + B * 32 * 2 * 128 * 4,
      "conv3x3_patch_kernel<bf16,128,2,2,2,2> data gradient + norm-backward sums (reads dy and y, writes dx) B=32 64x64 128->128")
entry(fe, wr, "conv3x3_wstat_kernel<true, 0,", None, 2 * act + wts, "conv3x3_wstat_kernel<norm prologue, no statistics> B=32 64x64 128->128")
entry(fe, wr, "conv3x3_wstat_kernel<false, 0,", None, 2 * act + wts, "conv3x3_wstat_kernel<plain> B=32 64x64 128->128")
entry(fe, wr, "conv3x3_wstat_kernel<true, 3,", None, act + B * P * P * 14 * 4 + 32 * 128 * 9 * 2, "conv3x3_wstat_kernel<narrow form 128 -> 14, fp32 NCHW out> B=32 64x64", required=False)
# round 4: the wave-specialised kernel, at the isolated comparison's 80 splits (240 workgroups of 512 threads) and at the engine's 24 (72)
for sp, grid in ((80, 240 * 512), (24, 72 * 512)):
    entry(fe, wr, "conv_wgrad3w_kernel<true, true,", grid, 2 * act + sp * 128 * 128 * 9 * 4, "conv_wgrad3w_kernel<norm> same shape, %d splits (algorithmic bytes incl. its split-K slabs)" % sp, required=False)
    entry(fe, wr, "conv_wgrad3w_kernel<false, false,", grid, 2 * act + sp * 128 * 128 * 9 * 4, "conv_wgrad3w_kernel<no norm> same shape, %d splits (algorithmic bytes incl. its split-K slabs)" % sp, required=False)
entry(fe, wr, "wgrad_reduce_fast_kernel<9", None, None, "wgrad_reduce_fast_kernel<9,3> same shape (the split count of the pass: 80 slabs = 47 MB)", required=False)
fd, wd = mean_by_kernel("pmc_fetch_dec", "FETCH_SIZE"), mean_by_kernel("pmc_write_dec", "WRITE_SIZE")
for (Bd, Jd, Pd, nt) in ((32, 14, 64, 256), (64, 21, 64, 256), (128, 42, 128, 512)):
    grid = Bd * Jd * nt
    # (round 6: with >= 2048 maps of 128 x 128 a forward workgroup owns several consecutive maps of a sample, so its grid is a fraction of
    # B J workgroups: the 512-thread kernels serve the 128 x 128 maps only, which one shape of the run has -- matched by name)
    if Pd == 128:
        entry(fd, wd, "decode_fwd_cached<512", None, 12 * Bd * Jd * Pd * Pd + 8 * Bd * Pd * Pd + 12 * Bd * Jd, "decode_fwd B=%d J=%d P=%d" % (Bd, Jd, Pd))
        entry(fd, wd, "decode_bwd_cached<512", None, 28 * Bd * Jd * Pd * Pd + 8 * Bd * Pd * Pd, "decode_bwd B=%d J=%d P=%d" % (Bd, Jd, Pd))
        continue
    entry(fd, wd, "decode_fwd_cached", grid, 12 * Bd * Jd * Pd * Pd + 8 * Bd * Pd * Pd + 12 * Bd * Jd, "decode_fwd B=%d J=%d P=%d" % (Bd, Jd, Pd))
    entry(fd, wd, "decode_bwd_cached", grid, 28 * Bd * Jd * Pd * Pd + 8 * Bd * Pd * Pd, "decode_bwd B=%d J=%d P=%d" % (Bd, Jd, Pd))
if MISSING:
    out["ERRORS"] = MISSING
json.dump(out, open(os.path.join(dst, PRE + "_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# 3. MFMA busy fraction of the dominant kernels
fs_ = [x for x in (find("pmc_mfma", "counter_collection.csv"), find("pmc_mfma24", "counter_collection.csv")) if x]
f = fs_[0] if fs_ else None
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    import itertools
    for r in itertools.chain(*[csv.DictReader(open(x)) for x in fs_]):
        key = r["Kernel_Name"] + (" grid %s" % r.get("Grid_Size", "") if "wgrad3w" in r["Kernel_Name"] else "")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    mm = {}
    for k, c in acc.items():
        if "conv3x3_wstat_kernel" in k or "conv3x3_patch_kernelIDF16bLi128ELi2ELi2ELi2ELi2ELb1" in k or "conv_wgrad3_kernel<2, 2, 1, 2" in k or "conv_wgrad3d_kernel<64, 128>" in k or "conv_wgrad3w_kernel" in k:
            m = {n: sum(v) / len(v) for n, v in c.items()}
            e = {"mean_ns_under_pmc": sum(dur[k]) / len(dur[k]), **m}
            if m.get("GRBM_GUI_ACTIVE", 0) > 0:
                cyc = m["GRBM_GUI_ACTIVE"] / 8.0
                e["cycles_per_launch"] = cyc
                e["mfma_busy_fraction"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / cyc
            mm[short(k)] = e
    json.dump(mm, open(os.path.join(dst, PRE + "_mfma_util.json"), "w"), indent=1)
    print(json.dumps(mm, indent=1))

# 4. data-parallel path (bench.py --force-dist, ONE rank: all this container's GPU box offers).  RCCL's device kernels are named
# ncclDevKernel_*; a one-rank all-reduce needs none (in place: nothing to do), so the trace can only show that the collective calls sit
# on the communication stream's queue between the segments without stalling the engine's queues -- not an overlap of real reductions.
f = find("dist", "kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rc = [r for r in rows if "nccl" in r["Kernel_Name"].lower()]
    byq = collections.Counter((r["Queue_Id"], r["Stream_Id"]) for r in rows)
    lines = ["kernels in the trace: %d on (queue, stream) -> count %s" % (len(rows), dict(byq.most_common(6))),
             "RCCL device kernels (ncclDevKernel_*): %d%s" % (len(rc), "" if rc else "  -- world size 1: the in-place all-reduce of one rank launches no kernel; "
             "the N > 1 overlap is the driver's to measure (SCALE_rNN.json)")]
    ov = 0
    for r in rc[-9:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        co = [q for q in rows if q is not r and int(q["Start_Timestamp"]) < e and int(q["End_Timestamp"]) > s and q["Queue_Id"] != r["Queue_Id"]]
        ov += bool(co)
        lines.append("RCCL kernel on queue %s: %.1f us, overlapped by %d engine kernels on other queues (e.g. %s)" % (
            r["Queue_Id"], (e - s) / 1e3, len(co), short(co[0]["Kernel_Name"])[:50] if co else "-"))
    if rc:
        lines.append("of the last %d RCCL kernels, %d overlap engine kernels of other queues" % (len(rc[-9:]), ov))
    open(os.path.join(dst, PRE + "_dist_overlap.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
for nm in ("step_bench.json", "trainsteps.json", "iso_bench.jsonl", "iso24_bench.jsonl", "dec_bench.jsonl", "dist_bench.json", "roofline_only.json", "serial_bench.json"):
    p = os.path.join(src, nm)
    if os.path.exists(p):
        open(os.path.join(dst, PRE + "_" + nm), "w").write(open(p).read())
if MISSING:
    sys.exit("profile_summary: %d traffic rows could not be produced:\n  " % len(MISSING) + "\n  ".join(MISSING))
