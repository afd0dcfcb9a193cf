"""Markdown table of the MEASURED errors of every parity comparison of a `pytest -m gpu` session
(gpurun_out/parity_errors.json, written by tests/conftest.py) - one row per test and dtype, worst case over its comparisons.
    python tools/parity_table.py [gpurun_out/parity_errors.json] > profiles/r02_parity_errors.md"""
import collections, json, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_errors.json"
d = json.load(open(src))
by = collections.OrderedDict()
for e in d:
    t = e["test"].split("::")[-1].split("[")[0]
    b = by.setdefault((t, e["dtype"]), dict(n=0, max_err=0.0, mean=0.0, ref=0.0, rel=0.0, bar=e["bar"]))
    b["n"] += 1
    b["max_err"] = max(b["max_err"], e["max_err"])
    b["mean"] = max(b["mean"], e["mean_err"])
    b["ref"] = max(b["ref"], e["max_ref"])
    b["rel"] = max(b["rel"], e["max_err"] / max(1.0, e["max_ref"]))
print("| test | dtype | comparisons | max abs err | max err / max(1, max\\|ref\\|) | mean abs err | bar in the test |")
print("|---|---|---|---|---|---|---|")
for (t, dt), b in by.items():
    print(f"| `{t}` | {dt} | {b['n']} | {b['max_err']:.3g} | {b['rel']:.2e} | {b['mean']:.2e} | {b['bar']} |")
