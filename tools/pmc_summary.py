#!/usr/bin/env python3
"""Turn the raw per-kernel sums of tools/pmc_pass.py (counter sets tools/pmc_sets/traffic.txt) into the per-SAMPLE summary
bench.py reads for its roofline:   python3 tools/pmc_summary.py RAW.json OUT.json --samples N [--note TEXT]
N = samples (pixels x spp x passes, warm-up render included) of the profiled command; per-sample figures apply to any
--steps because a sample's work does not depend on how many passes are in flight.

  valu_lane_ops_per_sample = SQ_THREAD_CYCLES_VALU / N        (active lanes summed over every VALU instruction)
  valu_insts_per_sample    = SQ_INSTS_VALU / N                (wave instructions)
  hbm_bytes_per_sample     = fabric-side bytes of the L2 / N: read = 32 B x RDREQ_32B + 128 B x RDREQ_128B + 64 B x (the rest),
                             write = 64 B x WRREQ_64B + 32 B x (the rest)  — the way rocprofv3 derives FETCH_SIZE / WRITE_SIZE
                             but with 128-byte requests at their own width (MI355X_MICROARCH.md, HBM: FETCH_SIZE tallies them
                             at 64 B on gfx950).  Infinity-Cache hits are included: an upper bound on DRAM traffic.
"""
import argparse, json, re
ap = argparse.ArgumentParser()
ap.add_argument("raw"); ap.add_argument("out"); ap.add_argument("--samples", type=float, required=True); ap.add_argument("--note", default="")
a = ap.parse_args()
raw = json.load(open(a.raw))
res = {"collected_with": "tools/pmc_pass.py RAW tools/pmc_sets/traffic.txt -- " + raw["command"] + "  (separate rocprofv3 --pmc passes, --kernel-trace only)",
       "samples_profiled": a.samples, "note": a.note, "kernels": {}}
agg = {}
for name, c in raw["kernels"].items():
    short = re.sub(r"^void ", "", name); short = re.sub(r"^ptd::", "", short); short = re.sub(r"<.*$", "", short) if short.startswith("wf_trace") else short
    d = agg.setdefault(short, {})
    for k, v in c.items():
        d[k] = d.get(k, 0.0) + v if k != "launches" else max(d.get(k, 0), v) if short != "wf_trace" else d.get(k, 0) + v
for name, c in agg.items():
    if "TCC_EA0_RDREQ_sum" not in c:
        continue
    k = {"launches": c.get("launches")}
    rd, rd32, rd128 = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    wr, wr64 = c.get("TCC_EA0_WRREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    k["hbm_read_bytes_per_sample"] = (32.0 * rd32 + 128.0 * rd128 + 64.0 * (rd - rd32 - rd128)) / a.samples
    k["hbm_write_bytes_per_sample"] = (64.0 * wr64 + 32.0 * (wr - wr64)) / a.samples
    k["hbm_bytes_per_sample"] = k["hbm_read_bytes_per_sample"] + k["hbm_write_bytes_per_sample"]
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        k["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c.get("SQ_INSTS_VALU"):
        k["valu_lane_ops_per_sample"] = c["SQ_THREAD_CYCLES_VALU"] / a.samples
        k["valu_insts_per_sample"] = c["SQ_INSTS_VALU"] / a.samples
        k["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        k["valu_busy_frac_in_profiled_run"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
        k["wave_cycles_waiting_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        k["salu_insts_per_sample"] = c.get("SQ_INSTS_SALU", 0.0) / a.samples
    k["raw"] = {x: v for x, v in c.items()}
    res["kernels"][name] = k
json.dump(res, open(a.out, "w"), indent=1)
for name, k in res["kernels"].items():
    print(name, {x: (round(v, 4) if isinstance(v, float) else v) for x, v in k.items() if x != "raw"})
