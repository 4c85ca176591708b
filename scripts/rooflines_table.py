"""Markdown table from profiles/r01_kernel_rooflines.json."""
import json
import sys
from pathlib import Path

path = Path(sys.argv[1] if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "profiles" / "r01_kernel_rooflines.json")
d = json.loads(path.read_text())
print("| entry point(s) | configuration | device µs | achieved (algorithmic) | of roof | |\n|---|---|---|---|---|---|")
for k in d["kernels"]:
    if "error" in k:
        continue
    cs = ", ".join(f"{a}={json.dumps(b).replace(' ', '')}" for a, b in k["config"].items())
    extra = ""
    for key, lab, div in (("transitions_per_s", "M transitions/s", 1e6), ("updates_per_s", "k updates/s", 1e3), ("gae_elements_per_s", "M GAE elements/s", 1e6),
                          ("state_steps_per_s", "M state-steps/s", 1e6), ("elements_per_s", "G elements/s", 1e9), ("rows_per_s", "M rows/s", 1e6)):
        if key in k:
            extra = f"{k[key] / div:.1f} {lab}"
    if "achieved" not in k:      # a multi-launch host loop: time only
        print(f"| `{k['entry']}` | {cs} | {k['device_us']:.1f} | — | {k['bound']}-bound | {k.get('note', '')} |")
        continue
    ach = f"{k['achieved']:.0f} GB/s" if k["unit"] == "GB/s" else f"{k['achieved']:.1f} TFLOP/s"
    print(f"| `{k['entry']}` | {cs} | {k['device_us']:.1f} | {ach} | {k['frac'] * 100:.1f} % {k['bound']} | {extra} |")
