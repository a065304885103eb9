#!/usr/bin/env python3
"""Turn the raw per-kernel sums of tools/pmc_pass.py (counter sets tools/pmc_sets/traffic.txt) into the summary
bench.py reads for roofline.traffic:   python3 tools/pmc_traffic.py RAW.json OUT.json --spp 64 --steps 8
HBM-side bytes per launch are computed from the L2's fabric request counters, the way rocprofv3 derives
FETCH_SIZE / WRITE_SIZE, but with the request sizes counted exactly:
  read  = 32 B x RDREQ_32B + 128 B x RDREQ_128B + 64 B x (RDREQ - RDREQ_32B - RDREQ_128B)
  write = 64 B x WRREQ_64B + 32 B x (WRREQ - WRREQ_64B)
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950 and has to be doubled
for wide loads; counting the 128-byte requests separately makes the correction exact instead).  Infinity-Cache hits
are included in these counters, so this is an upper bound on DRAM traffic."""
import argparse, json
ap = argparse.ArgumentParser()
ap.add_argument("raw"); ap.add_argument("out"); ap.add_argument("--spp", type=int, default=256); ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
raw = json.load(open(a.raw))
res = {"command": "python3 tools/pmc_pass.py RAW tools/pmc_sets/traffic.txt -- " + raw["command"] + "   (separate --pmc passes, --kernel-trace only)",
       "spp_per_pass": a.spp, "n_gpus": 1, "steps": a.steps, "kernels": {}}
for name, c in raw["kernels"].items():
    if "launches" not in c or "TCC_EA0_RDREQ_sum" not in c:
        continue
    k = dict(c)
    n = c["launches"]
    rd, rd32, rd128 = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    wr, wr64 = c["TCC_EA0_WRREQ_sum"], c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    k["read_bytes_per_launch"] = (32.0 * rd32 + 128.0 * rd128 + 64.0 * (rd - rd32 - rd128)) / n
    k["write_bytes_per_launch"] = (64.0 * wr64 + 32.0 * (wr - wr64)) / n
    k["hbm_bytes_per_launch"] = k["read_bytes_per_launch"] + k["write_bytes_per_launch"]
    k["fetch_size_equivalent_bytes_per_launch"] = (32.0 * rd32 + 64.0 * (rd - rd32)) / n      # what FETCH_SIZE would report
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        k["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c.get("SQ_ACTIVE_INST_VALU"):
        k["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        k["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
        k["wave_cycles_waiting_on_memory_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    res["kernels"][name] = k
    if "wf_trace" in name:
        res["hbm_bytes_per_launch"] = k["hbm_bytes_per_launch"]
res["note"] = "valu_busy_frac = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); sums are over every launch of the command (warm-up render included)"
json.dump(res, open(a.out, "w"), indent=1)
for name, k in res["kernels"].items():
    print(name, {x: (round(v, 4) if isinstance(v, float) and v < 10 else v) for x, v in k.items() if x in ("launches", "hbm_bytes_per_launch", "l2_hit_rate", "valu_lane_utilisation", "valu_busy_frac", "wave_cycles_waiting_on_memory_frac")})
