#!/usr/bin/env python3
"""Parse rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: one pass each, as MI355X_MICROARCH.md §rocprofv3 PMC slots
requires) of `bench.py --no-kernel-timing` into per-launch HBM-side traffic per kernel family.

    python tools/collect_traffic.py gpurun_out/r1 > profiles/r01/traffic.json

Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  gfx950 correction (guide §HBM): FETCH_SIZE reports half of the
bytes of a wide coalesced streaming read, so the streaming kernels (normalize_scan, bin_draws) are doubled.  Random 16-byte
row reads are counted exactly — one 64-byte fabric read per miss, calibrated with tools/gather_probe.hip
(profiles/r03/gather_probe_fetch.txt) — so the drawing k_propagate, whose fetches are its gathers, is taken as it is."""
import collections
import csv
import glob
import json
import sys

# one kernel per family: the kernels of the bench's multinomial step (the single-kernel resampler of the supplementary
# systematic leg is listed on its own)
FAM = {"k_propagate<mp_lgssm1": "propagate", "k_propagate_mt<mp_lgssm1": "propagate", "k_normalize_tiles": "normalize_scan", "k_draw_slots": "bin_draws",
       "k_resample_gather": "resample_gather"}
# (k_draw_slots reads little: Philox in, draws out)
STREAMING = {"normalize_scan", "bin_draws"}


NAMES = {}   # family -> kernel names seen (bench.py checks them against the kernels it times)


def per_kernel(dirname, counter):
    """mean per launch of each kernel, summed over the kernels of a family (one launch of each per resample)"""
    files = glob.glob(f"{dirname}/**/*_counter_collection.csv", recursive=True)
    per = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for key in FAM:
                if key in r["Kernel_Name"]:
                    per[key].append(float(r["Counter_Value"]))
                    NAMES.setdefault(FAM[key], set()).add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
    out, cnt = collections.defaultdict(float), collections.defaultdict(int)
    # two forms of one family's kernel may appear in a run (k_propagate for the first, plain step; k_propagate_mt for the steps of the
    # timed loop): the family is the form with the most launches, not their sum
    best = {}
    for key, v in per.items():
        if FAM[key] not in best or len(v) > len(per[best[FAM[key]]]):
            best[FAM[key]] = key
    for fam, key in best.items():
        v = per[key]
        out[fam] = sum(v) / len(v)
        cnt[fam] = len(v)
        NAMES[fam] = {n for n in NAMES.get(fam, set()) if key in n}
    return dict(out), dict(cnt)


def main(root):
    fetch, nf = per_kernel(root, "FETCH_SIZE")
    write, nw = per_kernel(root, "WRITE_SIZE")
    res = {}
    for fam in sorted(set(FAM.values())):
        f_raw = fetch.get(fam, 0.0) * 1024.0
        w = write.get(fam, 0.0) * 1024.0
        f_corr = f_raw * 2.0 if fam in STREAMING else f_raw
        note = "x2 (wide coalesced stream)" if fam in STREAMING else "none (random 16-B rows: uncalibrated width; true value in [1x, 2x] of raw)"
        if fam == "propagate":
            # calibrated (tools/gather_probe.hip under rocprofv3 --pmc FETCH_SIZE, profiles/r03/gather_probe_fetch.txt): a random
            # 16-byte row read that misses L2 is ONE 64-byte fabric read and FETCH_SIZE counts it in full (63.6 B per request from
            # a 1 GB table, 47.5 B from a 16 MB one of which a quarter hits L2) — the x2 correction applies to wide coalesced
            # streams only.  The drawing k_propagate's fetches are its gathers (guide cells, table rows); what it streams in
            # (2 x 8 B x tiles of tile scalars per workgroup) hits L2 after each XCD's first touch.
            f_corr = f_raw
            note = "x1: the launch's fetches are random guide-cell / table-row reads, which FETCH_SIZE counts exactly (calibrated with tools/gather_probe.hip)"
        res[fam] = {"fetch_bytes_raw": f_raw, "fetch_bytes_corrected": f_corr, "write_bytes": w, "traffic_bytes": f_corr + w,
                    "fetch_correction": note,
                    "launches_sampled": [nf.get(fam, 0), nw.get(fam, 0)]}
    import os
    import subprocess

    root_repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root_repo)
    from modppl_amd import build as B

    try:
        commit = subprocess.run(["git", "-C", root_repo, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = None
    # what this summary was measured with: bench.py refuses it for any other build of the kernels
    res["_measured"] = {"commit": commit or os.environ.get("MP_COMMIT"), "source_hash": B.source_hash(),
                        "kernels": {fam: " + ".join(sorted(v)) for fam, v in NAMES.items()}}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
