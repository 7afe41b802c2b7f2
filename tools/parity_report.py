"""profiles/rNN_parity_configs.md from the jsonl the config-level parity tests write (gpurun_out/parity_configs_*.jsonl).
usage: python tools/parity_report.py gpurun_out/parity_configs_<stamp>.jsonl [more.jsonl ...] > profiles/r03_parity_configs.md"""
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
for path in sys.argv[1:]:
    print(f"## run `{path.split('/')[-1]}`\n")
    for line in open(path):
        j = json.loads(line)
        name = j.pop("test")
        print(f"### {name}\n")
        w = j.pop("worst", None)
        if w:
            we = w.pop("synth_worst_element", None)
            print("| " + " | ".join(w) + " |\n|" + "---|" * len(w))
            print("| " + " | ".join(fmt(v) for v in w.values()) + " |\n")
            if we:
                print("worst synthesised element: " + ", ".join(f"{k} = {fmt(v)}" for k, v in we.items() if k != "index") + "\n")
        for k, v in j.items():
            print(f"* `{k}`: {json.dumps(short(v)) if isinstance(v, (list, dict)) else fmt(v)}")
        print()
