"""Builds profiles/jfa_dense_traffic.json (what bench.py reports as roofline.traffic / valu_issue_frac) from the PMC summaries
of tools/profile_round.sh:  python tools/traffic_json.py gpurun_out/r02 > profiles/jfa_dense_traffic.json"""
import json, os, re, sys
d = sys.argv[1]

def counters(name, n, kernel_re):
    out = {}
    path = os.path.join(d, "pmc_%s_n%d.summary.txt" % (name, n))
    if not os.path.exists(path):
        return out
    cur = None
    acc = {}
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
        elif cur and re.match(kernel_re, cur):
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
            if m:                                                    # several instantiations match (pair modes 8 / 4 / 2): launch-weighted mean
                a = acc.setdefault(m.group(1), [0.0, 0])
                a[0] += float(m.group(3)) * int(m.group(2)); a[1] += int(m.group(2))
    for k, (tot, cnt) in acc.items():
        out[k] = tot / cnt
    return out

res = {"round": int(re.sub(r"\D", "", os.path.basename(os.path.normpath(d))) or 0), "source": "separate rocprofv3 --pmc passes over tools/run_passes.py <n> 1 (tools/profile_round.sh), per-kernel means",
       "correction": "gfx950: read bytes = 2 x FETCH_SIZE (128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section); FETCH_SIZE / WRITE_SIZE are in KB"}
# the dense-pass instantiations of jfa_pass_dense (template arguments: id format, rows, planes, threads, ...)
for n, prefix, kre in ((512, "jfa_pass_dense<IdU<9>, 8, {8|16}, {256|512}, false", r"jfa_pass_dense<(vp::)?(\(anonymous namespace\)::)?IdU<9>, [48], (8|16), (256|512), false"),
                       (1024, "jfa_pass_dense<IdU<10>, {4|8}, 8, 512, false", r"jfa_pass_dense<(vp::)?(\(anonymous namespace\)::)?IdU<10>, [48], 8, 512, false")):
    c = {}
    for grp in ("fetch", "write", "l2", "sq1", "sq2"):
        c.update(counters(grp, n, kre))
    if "FETCH_SIZE" not in c:
        continue
    hbm = int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
    alg = 2 * 4 * n ** 3
    cycles = c.get("GRBM_GUI_ACTIVE", 0) / 8.0                       # summed over the 8 XCDs
    # VALU issue: a wave64 VALU instruction occupies its SIMD-32 for 2 cycles, v_min_f64 for 4 (tools/ubench/probe.hip);
    # every output voxel takes exactly 27 v_min_f64 steps, i.e. 27 n^3 / 64 wave instructions per launch.  (Pair mode: the
    # DPP forms cost ~1.7 clocks more each, tools/ubench/probe3.hip -- not in this estimate, which is therefore a lower bound.)
    valu = c.get("SQ_INSTS_VALU", 0)
    issue = valu * 2.0 + 2.0 * 27.0 * n ** 3 / 64.0
    entry = {"kernel": prefix + ", ...> (dense passes, bunny x24)", "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
             "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round(hbm / alg, 3),
             "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3) if "TCC_HIT_sum" in c else None,
             "SQ_INSTS_VALU": valu, "valu_per_voxel": round(valu * 64 / n ** 3, 1), "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU"),
             "salu_per_valu": round(c.get("SQ_INSTS_SALU", 0) / valu, 3) if valu else None, "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS"),
             "lds_bank_conflict_frac": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3) if c.get("SQ_LDS_IDX_ACTIVE") else None,
             "kernel_cycles": cycles, "valu_issue_frac": round(issue / (1024 * cycles), 3) if cycles else None}
    res["n%d" % n] = entry
    if n == 512:
        res.update({k: entry[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "valu_issue_frac")})
print(json.dumps(res, indent=1))
