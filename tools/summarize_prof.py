#!/usr/bin/env python3
"""Summarises the rocprofv3 passes of tools/profile.sh into profiles/<tag>_*.md/.csv.

usage: python tools/summarize_prof.py gpurun_out/prof_r01 r01
Kernel-trace rows are grouped by (kernel, grid size) so the finest-grid launches are
separated from the coarse ones. PMC traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE /
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read stream,
so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-byte streaming stores.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(mg::Geom.*", "", name)
    name = name.replace("void mg::(anonymous namespace)::", "").replace("mg::(anonymous namespace)::", "")
    return name.strip()


def newest(pattern):
    """gpurun merges every call's files into the same local directory: take the latest."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:] if files else []


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    os.makedirs(out_dir, exist_ok=True)
    trace = newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        gsz = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        wsz = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        groups[(short(r["Kernel_Name"]), gsz, wsz)].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
             # rocprofv3's VGPR_Count on gfx950 is HALF the allocation (granule 8): checked against hipcc's
             # -Rpass-analysis=kernel-resource-usage for five kernels (165 -> 84, 124 -> 64, 117 -> 60, 82 -> 44, 76 -> 40);
             # the column printed is the allocation = 2 x VGPR_Count (+ accumulation registers: none in this library)
             2 * int(r.get("VGPR_Count") or 0) + int(r.get("Accum_VGPR_Count") or 0), r.get("SGPR_Count"), r.get("LDS_Block_Size")))
    total = sum(sum(d for d, *_ in v) for v in groups.values())
    pmc = {}
    for cname, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        files = newest(os.path.join(src, sub, "*", "*_counter_collection.csv"))
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == cname:
                acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
        pmc[cname] = acc
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(os.path.join(src, "pmc_sq", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            sq[(short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    bench = {}
    bj = os.path.join(src, "bench_trace.json")
    if os.path.exists(bj):
        for line in open(bj):
            if line.startswith("{"):
                bench = json.loads(line)
    lines = [f"# rocprofv3 summary `{tag}` (MI355X, `rocprofv3 --kernel-trace --stats` + separate `--pmc` passes)", ""]
    if bench:
        lines += [f"bench line of the traced run: value={bench['value']:.2f} {bench['unit']}, ms_per_step={bench['ms_per_step']:.3f}, "
                  f"in-region smoother sweep {bench['roofline']['sweep_ms']:.4f} ms = {bench['roofline']['achieved']:.0f} GB/s "
                  f"(frac {bench['roofline']['frac']:.3f})", "", f"workload: {bench['config']['workload']}", ""]
    lines += ["| kernel | grid (threads) | wg | calls | avg µs | min µs | total ms | % | VGPRs allocated | read MB/launch (2×FETCH) | write MB/launch |",
              "|---|---|---|---|---|---|---|---|---|---|---|"]
    rows_csv = [("kernel", "grid_threads", "wg", "calls", "avg_us", "min_us", "total_ms", "pct", "vgpr", "read_MB", "write_MB")]
    for key, v in sorted(groups.items(), key=lambda kv: -sum(d for d, *_ in kv[1])):
        durs = [d for d, *_ in v]
        tot = sum(durs)
        if tot / total < 0.002:
            continue
        rd = pmc.get("FETCH_SIZE", {}).get(key)
        wr = pmc.get("WRITE_SIZE", {}).get(key)
        rd_mb = 2 * 1024 * sum(rd) / len(rd) / 1e6 if rd else None
        wr_mb = 1024 * sum(wr) / len(wr) / 1e6 if wr else None
        f = lambda x: f"{x:.1f}" if x is not None else "-"
        lines.append(f"| `{key[0]}` | {key[1]} | {key[2]} | {len(durs)} | {tot / len(durs) / 1e3:.2f} | {min(durs) / 1e3:.2f} | "
                     f"{tot / 1e6:.3f} | {100 * tot / total:.1f} | {v[0][1]} | {f(rd_mb)} | {f(wr_mb)} |")
        rows_csv.append((key[0], key[1], key[2], len(durs), tot / len(durs) / 1e3, min(durs) / 1e3, tot / 1e6, 100 * tot / total, v[0][1], rd_mb, wr_mb))
    if sq:
        lines += ["", "Wave-cycle breakdown (own `--pmc SQ_*` pass; quad-cycles summed over all waves, per launch): parked = SQ_WAIT_ANY "
                  "(s_waitcnt / barrier), stalled = SQ_WAIT_INST_ANY (issue stall), issuing = SQ_ACTIVE_INST_ANY; VALU / SALU = instructions issued.", "",
                  "| kernel | grid | wave-cycles | parked % | issue-stalled % | issuing % | of which VALU % / scalar % | VALU insts per wave-cycle | SALU:VALU insts |", "|---|---|---|---|---|---|---|---|---|"]
        for key, v in sorted(groups.items(), key=lambda kv: -sum(d for d, *_ in kv[1])):
            c = sq.get(key)
            if not c or sum(d for d, *_ in v) / total < 0.01:
                continue
            m = {k: sum(x) / len(x) for k, x in c.items()}
            wc = m.get("SQ_WAVE_CYCLES", 0) or 1
            lines.append(f"| `{key[0]}` | {key[1]} | {wc:.3g} | {100 * m.get('SQ_WAIT_ANY', 0) / wc:.0f} | {100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} | "
                         f"{100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f} | {100 * m.get('SQ_ACTIVE_INST_VALU', 0) / wc:.0f} / {100 * m.get('SQ_ACTIVE_INST_SCA', 0) / wc:.0f} | "
                         f"{m.get('SQ_INSTS_VALU', 0) / wc:.3f} | {m.get('SQ_INSTS_SALU', 0) / max(m.get('SQ_INSTS_VALU', 1), 1):.2f} |")
    lines += ["", "Traffic columns: PMC counters from their own passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950), "
              "per launch, with the gfx950 correction read = 2 × FETCH_SIZE × 1024 B (MI355X_MICROARCH.md §HBM); "
              "Infinity-Cache hits are counted as traffic by these fabric-side counters.", ""]
    open(os.path.join(out_dir, f"{tag}_kernel_summary.md"), "w").write("\n".join(lines))
    with open(os.path.join(out_dir, f"{tag}_kernel_summary.csv"), "w", newline="") as fcsv:
        csv.writer(fcsv).writerows(rows_csv)
    st = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if st:
        open(os.path.join(out_dir, f"{tag}_rocprofv3_kernel_stats.csv"), "w").write(open(st[0]).read())
    print("\n".join(lines))


if __name__ == "__main__":
    main()
