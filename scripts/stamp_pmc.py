"""profiles/pmc_traffic.json from the PMC summaries of one visit (scripts/summarize_pmc.py output for V0 and V2):
bytes per 512^3 launch of the headline kernel and per V2 step (its three sweep launches together), stamped with the hash of
the kernel sources they were measured on — bench.py only reports a traffic figure whose hash matches the code it runs.

    python scripts/stamp_pmc.py <v0_summary.json (single steps)> <v2_summary.json> <tag, e.g. r3j> [<v0 two-step summary.json>]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

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
