"""One-screen summary of a bench.py line (+ its per-op table): python tools/bench_summary.py <bench.json> [bench_ops.json]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, "seq", d["sequential"]["value"], "frac", d["roofline"]["frac"], "dropin", d["dropin_api"]["value"],
      "tuned", d["dropin_api_tuned"]["value"], "attempts", d.get("supervisor", {}).get("attempts"))
print({k: (round(v["ms"], 4), v["launches"]) for k, v in d["breakdown"].items()})
t = d.get("train", {})
print("train", t.get("ms_per_step"), t.get("mfma_frac"), t.get("wgrad"), "| l6", d.get("l6", {}).get("value"), d.get("l6", {}).get("sequential"),
      "| int8", d.get("int8", {}).get("value"), d.get("int8", {}).get("sequential"), "fp16 same", d.get("int8_fp16_same_model", {}).get("value"),
      "| n_b1", d.get("n_b1", {}).get("value"))
if len(sys.argv) > 2:
    ops = json.load(open(sys.argv[2]))["rows"]
    print(" ".join(f"{o['op']}:{o['kind'][:6]}/{o['variant']}={o['ms'] * 1000:.0f}" for o in ops))
