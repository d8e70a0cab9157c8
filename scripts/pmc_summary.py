"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py per kernel -> JSON.
usage: python scripts/pmc_summary.py <dir_with_pass_subdirs> <out.json> <commit>

Per kernel: KiB per launch (mean over all launches of the pass) and launches per bsc_ingest call; `ingest` marks the
kernels of the memory path (libbscnav's own + the rocPRIM sorts / scans and fills it issues), as opposed to the encoder.
bench.py sums (2 x FETCH + WRITE) x launches_per_call over the ingest kernels for `roofline.traffic`."""
import collections, csv, json, os, sys
root, out_path, commit = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "unknown")
ENCODER = ("k_attention", "k_add_layernorm", "k_bias_layernorm", "k_embed_layernorm", "k_final_layernorm", "k_preprocess", "k_pp_taps", "Cijk", "Custom_Cijk",
           "at::native", "__amd_rocclr_copyBuffer", "k_cosine", "k_cand", "k_block_topk", "k_gather", "k_normalize_q", "k_name", "k_pool")
out, calls = {}, None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(root, c, "pmc_counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
    n_calls = max(1, len(next((v for k, v in agg.items() if "k_points" in k), [0])))
    for k, v in agg.items():
        name = k.replace("void ", "")
        e = out.setdefault(name, {})
        e[c + "_KiB_per_launch"] = sum(v) / len(v)
        e["launches_per_call"] = len(v) / n_calls
        e["ingest"] = not any(name.startswith(p) or p in name[:40] for p in ENCODER)
tot = sum((2 * v.get("FETCH_SIZE_KiB_per_launch", 0) + v.get("WRITE_SIZE_KiB_per_launch", 0)) * v["launches_per_call"]
          for v in out.values() if v["ingest"]) * 1024
json.dump({"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python bench.py --no-cpu-baseline --no-localize "
                      "--no-workloads --repeats 1 (two separate passes; 8 steps x 384 frames, room depth)",
           "commit": commit,
           "units": "KiB per launch, mean over all launches of the pass; FETCH_SIZE is raw (gfx950 reports half of wide coalesced "
                    "reads, MI355X_MICROARCH.md, HBM): traffic = 2 x FETCH + WRITE",
           "ingest_traffic_bytes_per_call": tot, "kernels": out}, open(out_path, "w"), indent=1)
print("ingest traffic per call: %.1f MB" % (tot / 1e6))
for k, v in sorted(out.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE_KiB_per_launch", 0) + kv[1].get("WRITE_SIZE_KiB_per_launch", 0)) * kv[1]["launches_per_call"]):
    if v["ingest"]:
        print(f"  {k[:60]:60s} x{v['launches_per_call']:5.1f}  fetch {2 * v.get('FETCH_SIZE_KiB_per_launch', 0) / 1024:8.1f} MB  write {v.get('WRITE_SIZE_KiB_per_launch', 0) / 1024:8.1f} MB")
