"""Per-kernel HBM traffic from the two rocprofv3 --pmc passes of scripts/collect_pmc.sh -> profiles/<tag>_pmc_traffic.json.

    python scripts/make_pmc_traffic.py gpurun_out/pmc_fetch_size gpurun_out/pmc_write_size profiles/r02_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are averaged per dispatch of each kernel (unit KB).  gfx950 correction (MI355X_MICROARCH.md, "HBM"):
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it is a lower bound up to 2x for kernels that mix
dword / x2 / x4 loads; WRITE_SIZE is exact for 16-byte stores.  `source_sha16` is the hash of the kernel sources the counters were
collected on: bench.py reports `roofline.traffic` only while the tree still hashes to it (a stale file is refused, not quoted).
"""
import csv
import hashlib
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KERNEL_SOURCES = ["sac.hip", "sac_lean.hip", "sac_lean.hpp", "sac_shared.hpp", "lean_blocks.hpp", "chain_run.hpp", "wave_mlp.hpp", "common.hpp", "p2p.hpp"]


def source_sha16() -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update((ROOT / "model-based-policy-optimizers_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def averages(d: Path, counter: str):
    f = max(d.rglob("*counter_collection.csv"), key=lambda q: q.stat().st_mtime)      # the newest pass (gpurun_out/ accumulates)
    acc, cnt = defaultdict(float), defaultdict(int)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].replace("void ", "")
            name = name.split("(")[0]
            acc[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}, f


if __name__ == "__main__":
    fetch_dir, write_dir, out = Path(sys.argv[1]), Path(sys.argv[2]), Path(sys.argv[3])
    fetch, f1 = averages(fetch_dir, "FETCH_SIZE")
    write, f2 = averages(write_dir, "WRITE_SIZE")
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (scripts/collect_pmc.sh) of "
                     "`python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph`; averages per dispatch, unit KB",
           "note": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads (lower bound up to 2x for mixed-width loads); "
                   "WRITE_SIZE is exact for the slab / row stores",
           "source_sha16": source_sha16(), "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        res["kernels"][k] = {"fetch_kb": round(fetch.get(k, (0, 0))[0], 1), "write_kb": round(write.get(k, (0, 0))[0], 1),
                             "dispatches": fetch.get(k, write.get(k))[1]}
    out.write_text(json.dumps(res, indent=1))
    print(json.dumps(res["kernels"], indent=1))
