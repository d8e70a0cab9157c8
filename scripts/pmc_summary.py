"""Summarise the two rocprofv3 --pmc passes of bench.py (read-request and write-request counters of the L2's memory side) per
kernel -> JSON.   usage: python scripts/pmc_summary.py <dir_with_pass_subdirs RD WR> <out.json> <commit>

HBM-side bytes per launch, exact by request size (scripts/pmc_calibrate.sh checks them on known-byte kernels: a 4 GiB copy reads
33.55 M x 128 B and writes 67.11 M x 64 B; a random 4-byte gather fetches one 128-byte line per element):
    read  = 32 * TCC_EA0_RDREQ_32B + 64 * TCC_EA0_RDREQ_64B + 128 * TCC_EA0_RDREQ_128B
    write = 64 * TCC_EA0_WRREQ_64B + 32 * (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B)
(FETCH_SIZE = 64 B x TCC_EA0_RDREQ counts every 128-byte request at half.)  Per kernel: mean over all launches of the pass and
launches per bsc_ingest call; `ingest` marks the kernels of the memory path (libbscnav's own + the rocPRIM sorts / scans and
fills it issues), as opposed to the encoder.  bench.py sums (read + write) x launches_per_call over them for `roofline.traffic`."""
import collections, csv, hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENCODER = ("k_gemm_split", "k_layernorm_split", "k_split_weights", "k_split_rows", "k_attention", "k_add_layernorm", "k_bias_layernorm", "k_embed_layernorm", "k_final_layernorm", "k_preprocess", "k_pp_taps", "Cijk", "Custom_Cijk",
           "at::native", "__amd_rocclr_copyBuffer", "k_cosine", "k_cand", "k_block_topk", "k_gather", "k_normalize_q", "k_name", "k_pool")
INGEST_SOURCES = ("ingest.hip", "dense.hip", "radix.hip", "prims.hip", "capi.hip", "geometry_dev.h", "bsc_internal.h")


def ingest_sources_sha():
    """sha256 over the sources the ingest kernels are built from: a PMC file records it, bench.py quotes the file only while it
    still matches the tree (a stale profile can no longer be quoted)."""
    h = hashlib.sha256()
    for f in INGEST_SOURCES:
        with open(os.path.join(ROOT, "bsc-nav_amd", "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def summarise(rd_csv, wr_csv, frames_per_call, commit="unknown", command="", token_bytes=4):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in (rd_csv, wr_csv):
        for r in csv.DictReader(open(path)):
            agg[r["Kernel_Name"].split("(")[0][:70].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    n_calls = max(1, len(next((v["TCC_EA0_WRREQ"] for k, v in agg.items() if "k_points" in k), [0])))
    out = {}
    for name, cs in agg.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        launches = len(next(iter(cs.values())))
        rd = 32 * m.get("TCC_EA0_RDREQ_32B", 0) + 64 * m.get("TCC_EA0_RDREQ_64B", 0) + 128 * m.get("TCC_EA0_RDREQ_128B", 0)
        wr = 64 * m.get("TCC_EA0_WRREQ_64B", 0) + 32 * (m.get("TCC_EA0_WRREQ", 0) - m.get("TCC_EA0_WRREQ_64B", 0))
        out[name] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "launches_per_call": launches / n_calls,
                     "ingest": not any(name.startswith(p) or p in name[:40] for p in ENCODER)}
    tot = sum((v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches_per_call"] for v in out.values() if v["ingest"])
    return {"command": command, "commit": commit, "ingest_sources_sha16": ingest_sources_sha(), "frames_per_call": frames_per_call,
            "token_bytes": token_bytes,
            "units": "bytes per launch (mean over all launches of the pass), by request size: read = 32 RDREQ_32B + 64 RDREQ_64B + 128 RDREQ_128B, "
                     "write = 64 WRREQ_64B + 32 (WRREQ - WRREQ_64B); calibration: profiles/r03_pmc_calibration.txt",
            "ingest_traffic_bytes_per_call": tot, "kernels": out}


def text(d):
    lines = ["ingest traffic per call: %.1f MB" % (d["ingest_traffic_bytes_per_call"] / 1e6)]
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -(kv[1]["read_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches_per_call"]):
        if v["ingest"]:
            lines.append(f"  {k[:60]:60s} x{v['launches_per_call']:5.1f}  read {v['read_bytes_per_launch'] / 1e6:8.1f} MB  write {v['write_bytes_per_launch'] / 1e6:8.1f} MB")
    return "\n".join(lines)


if __name__ == "__main__":
    root, out_path, commit = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "unknown")
    _bench = open(os.path.join(ROOT, "bench.py")).read()
    frames_per_call = int(sys.argv[4]) if len(sys.argv) > 4 else int(re.search(r'"--batch", type=int, default=(\d+)', _bench).group(1))
    d = summarise(os.path.join(root, "RD", "pmc_counter_collection.csv"), os.path.join(root, "WR", "pmc_counter_collection.csv"), frames_per_call, commit,
                  command="rocprofv3 --pmc <TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B | TCC_EA0_WRREQ TCC_EA0_WRREQ_64B> "
                          "--kernel-trace -- python bench.py --no-cpu-baseline --no-localize --no-workloads --no-side-precision --no-host-feed --no-exact --no-pmc "
                          f"--repeats 1 (two separate passes; the default f32 pipeline, f32 tokens; 8 steps x {frames_per_call} frames, room depth), "
                          "or of scripts/ingest_only.py (bsc_ingest alone) when the tag says `only`",
                  token_bytes=int(os.environ.get("BSC_PMC_TOKEN_BYTES", "4")))
    json.dump(d, open(out_path, "w"), indent=1)
    print(text(d))
