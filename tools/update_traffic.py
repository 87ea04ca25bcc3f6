#!/usr/bin/env python3
"""Builds profiles/lk_traffic.json (read by bench.py for roofline.traffic) from the FETCH_SIZE / WRITE_SIZE
summaries that tools/profile_round.sh leaves in gpurun_out/.   usage: update_traffic.py <tag> <streams>"""
import csv, json, os, sys
tag, streams = sys.argv[1], int(sys.argv[2])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def lk(counter):
    for row in csv.DictReader(open(os.path.join(root, "gpurun_out", "prof_%s_pmc_%s.csv" % (tag, counter)))):
        if "lk_fb_kernel" in row["kernel"] and row["counter"] == counter:
            return row["kernel"], float(row["mean_per_launch"])
    raise SystemExit("no lk_fb_kernel row for " + counter)
name, f = lk("FETCH_SIZE")
_, w = lk("WRITE_SIZE")
def sq(counter):
    path = os.path.join(root, "gpurun_out", "prof_%s_pmc_SQ_INSTS_VALU.csv" % tag)
    if not os.path.exists(path):
        return None
    for row in csv.DictReader(open(path)):
        if "lk_fb_kernel" in row["kernel"] and row["counter"] == counter:
            return float(row["mean_per_launch"])
    return None
sys.path.insert(0, root)
from bench import csrc_sha16  # noqa: E402 -- the hash bench.py checks before it trusts this file
P = 752 * 480 + 376 * 240 + 188 * 120 + 94 * 60
out = {
    "kernel": name.replace("void vsf::", ""), "streams": streams, "config_name": "C2",
    "csrc_sha16": csrc_sha16(),
    "config": "C2 752x480, 4 levels, 21x21, ~245 features/stream, 6 % of every frame under moving occluders",
    "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
    "hbm_bytes_per_launch": int((f + w) * 1024),
    "algorithmic_bytes_per_launch": (2 * P + 24 * 300) * streams,
    "valu_insts_per_launch": sq("SQ_INSTS_VALU"), "salu_insts_per_launch": sq("SQ_INSTS_SALU"),
    "lds_insts_per_launch": sq("SQ_INSTS_LDS"), "waves_per_launch": sq("SQ_WAVES"),
    "collection": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `python3 bench.py --steps 3 "
                  "--warmup 2 --no-cpu` (tools/profile_round.sh %s), mean over the launches; summaries in "
                  "profiles/%s_pmc_FETCH_SIZE.csv / _WRITE_SIZE.csv" % (tag, tag),
    "correction": "none applied: the x2 FETCH_SIZE correction of MI355X_MICROARCH.md holds for 16 B/lane streaming "
                  "reads; this kernel issues 4 B/lane dword loads. Calibrated on pyr_down_kernel in the same kind of "
                  "run (byte loads, known traffic): FETCH_SIZE over the 3 levels = 536 MB vs 485 MB source bytes + "
                  "~12 % tile-halo re-reads = 543 MB, i.e. factor 1.0.",
    "note": "XCD-aware block mapping keeps each stream's windows in one L2 (2,911,273 KB fetched per launch before "
            "it, round 1). Since the template cache (round 2) the backward pass writes each tracked feature's "
            "per-level template (2.5 KB a level) and the next launch's forward pass reads it back instead of "
            "rebuilding it: ~2.7 GB written and ~1.3 GB of the fetches per 1024-stream launch are that cache, "
            "traded for a third fewer VALU instructions in a kernel that is VALU-issue bound (1.6 TB/s of traffic, "
            "20 % of HBM peak, not the limit). The cache is not part of the algorithmic bytes.",
}
json.dump(out, open(os.path.join(root, "profiles", "lk_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
