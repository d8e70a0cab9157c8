"""Print the headline fields of a bench.py JSON line read from stdin (helper for sweeps)."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    st = d.get("stages", {})
    print(tag, round(d["value"]), round(d["ms_per_step"], 2), "frac", round(d.get("roofline", {}).get("frac", 0), 3),
          "iso", round(d.get("roofline_isolated", {}).get("frac", 0), 3), "enc", round(st.get("encoder_ms_per_step", 0), 2),
          "ing", round(st.get("ingest_ms_per_step", 0), 2))
except Exception as e:
    print(tag, "ERR", e)
