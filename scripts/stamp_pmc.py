"""profiles/pmc_traffic.json from the PMC summaries of one visit (scripts/summarize_pmc.py output for V0 and V2):
bytes per 512^3 launch of the headline kernel and per V2 step (its three sweep launches together), stamped with the hash of
the kernel sources they were measured on — bench.py only reports a traffic figure whose hash matches the code it runs.

    python scripts/stamp_pmc.py <v0_summary.json (single steps)> <v2_summary.json> <tag, e.g. r3j> [<v0 two-step summary.json>]
    python scripts/stamp_pmc.py --visit gpurun_out/TAG profiles/r4/TAG       (summaries of scripts/gpu_visit.sh ... pmc: v0, v0s, v1, va, v2)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

if sys.argv[1] == "--visit":
    # the summaries of one `gpu_visit.sh TAG pmc`: every workload of the bench line.  A workload that runs in step pairs is recorded
    # per PAIR (all launches of the pair together: the two-step sweep + the seam kernel, and for CPML grids the shell's launches)
    src, dst = sys.argv[2], sys.argv[3]
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    S = {w: json.load(open(os.path.join(src, f"pmc_{w}_summary.json"))) for w in ("v0", "v0s", "v1", "va", "v2", "v3")
         if os.path.exists(os.path.join(src, f"pmc_{w}_summary.json"))}
    k0 = [k for k in S["v0s"] if k.startswith("fused_step_kernel<false, 256, 0")]
    rec["fused_step_kernel"] = S["v0s"][k0[0]]["hbm_bytes_per_launch"]
    wl = {}
    for w in ("v0", "v1", "va", "v2", "v3"):
        if w not in S:
            continue
        ks = {k: v for k, v in S[w].items() if "hbm_bytes_per_launch" in v and any(t in k for t in ("fused", "seam_kernel", "strip_step", "shell2_step", "ade2_kernel"))}
        pairs = ks.get("fused2_step_kernel", {}).get("launches_FETCH_SIZE", 0)
        if pairs:
            tot = sum(v["hbm_bytes_per_launch"] * v["launches_FETCH_SIZE"] for v in ks.values())
            wl[w] = {"bytes_per_pair": tot / pairs, "pairs": pairs,
                     "parts": {k: {"bytes_per_launch": v["hbm_bytes_per_launch"], "launches_per_pair": v["launches_FETCH_SIZE"] / pairs,
                                   "read": v["read_bytes_per_launch"], "write": v["write_bytes_per_launch"]} for k, v in ks.items()}}
    rec["workloads"] = wl
    if "v0" in wl:
        rec["fused2_step_kernel"] = wl["v0"]["bytes_per_pair"]
        rec["fused2_parts"] = {k: {"bytes_per_launch": v["bytes_per_launch"], "read": v["read"], "write": v["write"]} for k, v in wl["v0"]["parts"].items()}
    if "v2" in wl:
        rec["v2_step_bytes"] = wl["v2"]["bytes_per_pair"] / 2
        rec["v2_launches"] = wl["v2"]["parts"]
    rec["file"] = ", ".join(f"{dst}_pmc_{w}_summary.json" for w in S)
    rec["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    rec["source_hash"] = bench.source_hash()
    rec["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, placement probe off; reads = 2 * FETCH_SIZE * 1024 (gfx950 correction, "
                    "MI355X_MICROARCH.md), writes = WRITE_SIZE * 1024; the counters sit on the L2 -> fabric side: Infinity-Cache hits are counted.  "
                    "fused_step_kernel: per 512^3 launch (single steps); workloads.*: per step PAIR, all launches of the pair")
    json.dump(rec, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    for w in S:
        json.dump(S[w], open(os.path.join(ROOT, f"{dst}_pmc_{w}_summary.json"), "w"), indent=1)
    print(json.dumps({"fused_step_kernel": rec["fused_step_kernel"], "workloads": {w: v["bytes_per_pair"] for w, v in wl.items()}, "source_hash": rec["source_hash"]}))
    sys.exit(0)
v0 = json.load(open(sys.argv[1]))
v2 = json.load(open(sys.argv[2]))
tag = sys.argv[3]
old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
k0 = [k for k in v0 if k.startswith("fused_step_kernel<false, 256, 0")]
assert len(k0) == 1, list(v0)
fused2 = {k: v for k, v in v2.items() if k.startswith("fused_step_kernel")}
steps = min(v["launches_FETCH_SIZE"] for k, v in fused2.items() if ", 1," in k or ", 9," in k)       # the interior launch: one per step
v2_step = sum(v["hbm_bytes_per_launch"] * v["launches_FETCH_SIZE"] for v in fused2.values()) / steps
two = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
out = {
    "fused_step_kernel": v0[k0[0]]["hbm_bytes_per_launch"],
    "v2_step_bytes": v2_step,
    "v2_launches": {k: {"bytes_per_launch": v["hbm_bytes_per_launch"], "launches_per_step": v["launches_FETCH_SIZE"] / steps} for k, v in fused2.items()},
    "file": f"profiles/{tag}_pmc_v0_summary.json, profiles/{tag}_pmc_v2_summary.json", "commit": commit, "source_hash": bench.source_hash(),
    "_note": "bytes per 512^3 launch of " + k0[0] + " and per V2 step (interior x-CPML launch + the two all-axes edge launches): rocprofv3 --pmc "
             "FETCH_SIZE / WRITE_SIZE in separate passes, placement probe off; reads = 2 * FETCH_SIZE * 1024 (gfx950 correction, "
             "MI355X_MICROARCH.md), writes = WRITE_SIZE * 1024; the counters sit on the L2 -> fabric side: Infinity-Cache hits are counted",
    "h_update_kernel": old.get("h_update_kernel"), "e_update_kernel": old.get("e_update_kernel"),
    "_two_pass_source": old.get("_two_pass_source"),
}
if two is not None:
    # one launch of the two-step sweep = fused2_step_kernel + the seam kernel (two time steps)
    parts = {k: two[k] for k in ("fused2_step_kernel", "seam_kernel") if k in two}
    out["fused2_step_kernel"] = sum(v["hbm_bytes_per_launch"] for v in parts.values())
    out["fused2_parts"] = {k: {"bytes_per_launch": v["hbm_bytes_per_launch"], "read": v["read_bytes_per_launch"],
                               "write": v["write_bytes_per_launch"]} for k, v in parts.items()}
    out["file"] += f", profiles/{tag}_pmc_v0_two_step_summary.json"
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: out.get(k) for k in ("fused_step_kernel", "fused2_step_kernel", "v2_step_bytes", "source_hash", "commit")}))
