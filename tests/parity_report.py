"""profiles/rNN_parity_configs.md from the jsonl the config-level parity tests write (gpurun_out/parity_configs_*.jsonl).
usage: python tests/parity_report.py gpurun_out/parity_configs_<stamp>.jsonl [more.jsonl ...] > profiles/r03_parity_configs.md"""
import json
import sys


def short(v, n=10):
    return v if not isinstance(v, list) or len(v) <= 2 * n else v[:n] + ["..."] + v[-n:]


def fmt(x):
    return f"{x:.3g}" if isinstance(x, float) else str(x)


print("# Parity at the benchmarked configurations — measured numbers (round 3)\n")
print("Written by `tests/test_gpu_parity_configs.py` on one MI355X (`pytest -m gpu`); one section per run file.  The\n"
      "kernel-isolating legs share ONE classifier evaluation per step between the two sides of the comparison\n"
      "(`tests/parity_tools.py`); `worst` = maximum over all steps of the trajectory.\n")
runs = [[json.loads(line) for line in open(path)] for path in sys.argv[1:]]
if len(runs) > 1:
    print(f"## Run-to-run spread over {len(runs)} runs (separate processes / boxes of the pool)\n")
    print("Every figure below was under its asserted bound in every run (all runs green); min … max over the runs.\n")
    names = []
    for run in runs:
        names += [j["test"] for j in run if j["test"] not in names]
    for name in names:
        recs = [j for run in runs for j in run if j["test"] == name]
        if recs and "worst" in recs[0]:
            # the contraction columns changed their reference once (fp32 GEMM of the other side -> its fp64 evaluation):
            # only runs measured the final way enter the spread
            recs = [r for r in recs if r["worst"].get("contraction_reference") == "fp64"] or recs
            keys = [k for k in recs[0]["worst"] if k not in ("synth_worst_element", "contraction_reference")]
            print(f"**{name}** (maxima over the steps of a trajectory)\n")
            print("| " + " | ".join(keys) + " |\n|" + "---|" * len(keys))
            print("| " + " | ".join(f"{fmt(float(min(r['worst'][k] for r in recs)))} … {fmt(float(max(r['worst'][k] for r in recs)))}" for k in keys) + " |\n")
        elif name == "asr_parity_structured":
            print("**asr_parity_structured** (ASR through the reference's pipeline on held-out structured images)\n")
            print("| run | held-out images | ASR A (fp32 reference configuration) | ASR C (bf16 product) | oracle fp32 inference with C's dictionary | PRODUCT fp32 inference with C's dictionary | \\|A − C\\| |\n|---|---|---|---|---|---|---|")
            for i, r in enumerate(recs):
                a, c = r["asr_A"], r["asr_C"]
                x = r["asr_oracle_inference_fp32_net_with_the_products_dictionary"]
                nx = r.get("cross_check_images", r.get("samples", 512))
                p32 = r.get("asr_product_inference_fp32_streams_fp32_net_with_the_products_dictionary")
                p32 = f"{100 * p32:.2f} %" if p32 is not None else "—"
                print(f"| {i + 1} | {r.get('samples', 512)} | {100 * a:.2f} % | {100 * c:.2f} % | {100 * x:.2f} % (of {nx}) | {p32} | {100 * abs(a - c):.2f} pp |")
            print()
if len(sys.argv) > 3:
    print(f"(Per-run details below: the two most recent of the {len(sys.argv) - 1} runs; the others enter the spread above.)\n")
for path in sys.argv[1:][-2:]:
    print(f"## run `{path.split('/')[-1]}`\n")
    for line in open(path):
        j = json.loads(line)
        name = j.pop("test")
        print(f"### {name}\n")
        w = j.pop("worst", None)
        if w:
            we = w.pop("synth_worst_element", None)
            w.pop("contraction_reference", None)
            print("| " + " | ".join(w) + " |\n|" + "---|" * len(w))
            print("| " + " | ".join(fmt(v) for v in w.values()) + " |\n")
            if we:
                print("worst synthesised element: " + ", ".join(f"{k} = {fmt(v)}" for k, v in we.items() if k != "index") + "\n")
        for k, v in j.items():
            print(f"* `{k}`: {json.dumps(short(v)) if isinstance(v, (list, dict)) else fmt(v)}")
        print()
