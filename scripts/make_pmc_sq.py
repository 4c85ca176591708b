"""SQ counters per kernel from scripts/collect_pmc_sq.sh -> profiles/<tag>_pmc_sq.json.

    python scripts/make_pmc_sq.py gpurun_out/pmc_sq profiles/r02_pmc_sq.json

Averages per dispatch.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
the kernel's waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES per shader engine.  Derived:
  wait_any / wait_inst / active = fractions of the wave cycles (parked at s_waitcnt or a barrier / issue-stalled / issuing);
  mfma_busy_of_used_simds = MFMA busy cycles / (4 cycles x wave quad-cycles / waves per SIMD of the launch) is NOT formed here —
  the JSON keeps the raw averages and the per-wave instruction counts; DESIGN.md section 4 does the arithmetic with the launch shapes.
"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


def main(src: str, out: str, what: str = "python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph") -> None:
    f = max(Path(src).rglob("*counter_collection.csv"), key=lambda q: q.stat().st_mtime)
    acc = defaultdict(lambda: defaultdict(list))
    grid = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            grid[k] = (int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0))
    kernels = {}
    for k, v in acc.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        g, wg = grid.get(k, (0, 0))
        waves = g // 64 if g else 0
        e = {"dispatches": len(next(iter(v.values()))), "grid_threads": g, "workgroup_threads": wg, "waves": waves,
             **{c: round(x, 1) for c, x in m.items()}}
        if wc:
            e["frac_wait_any"] = round(m.get("SQ_WAIT_ANY", 0) / wc, 3)
            e["frac_wait_inst"] = round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
            e["frac_active"] = round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
        if waves:
            e["valu_insts_per_wave"] = round(m.get("SQ_INSTS_VALU", 0) / waves, 1)
            e["salu_insts_per_wave"] = round(m.get("SQ_INSTS_SALU", 0) / waves, 1)
            e["cycles_per_wave"] = round(4 * wc / waves, 1)
        kernels[k] = e
    Path(out).write_text(json.dumps({
        "source": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES "
                  "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU (scripts/collect_pmc_sq.sh, scripts/collect_round_extras.sh) of `" + what + "`; averages per dispatch",
        "file": f.name, "kernels": kernels}, indent=1))
    for k in kernels:
        if any(s in k for s in ("sac_fwd", "sac_lean", "reduce_apply", "rollout64", "rollout_lean", "ppo_lean", "ppo_values", "bptt_actor", "critic_fwd")):
            print(k, json.dumps(kernels[k]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
