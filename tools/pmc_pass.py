#!/usr/bin/env python3
"""Collect PMC counters per kernel, one rocprofv3 pass per counter set, and print/write the sums.

Runs on the GPU box:   python3 tools/pmc_pass.py OUT.json SETFILE -- python3 bench.py --no-cpu-baseline ...
SETFILE holds one counter set per line (space separated).  Counters are collected in their own
passes with --kernel-trace only (never combined with sys/hip traces).  This launcher does not touch
the GPU itself; the profiled program is started as a child by rocprofv3.
"""
import csv, glob, json, os, subprocess, sys, tempfile, time


def main():
    out, setfile = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    sets = [l.split() for l in open(setfile) if l.strip() and not l.startswith("#")]
    agg = {}
    per_set_timeout = int(os.environ.get("PMC_SET_TIMEOUT", "240"))
    env = dict(os.environ, TMPDIR="/tmp")
    for i, cs in enumerate(sets):
        d = tempfile.mkdtemp(prefix="pmc%d_" % i, dir="/tmp")
        t0 = time.time()
        print("set %d %s ..." % (i, cs), flush=True)
        logf = open(os.path.join(d, "stdout.log"), "w")
        proc = subprocess.Popen(["rocprofv3", "--kernel-trace", "--pmc"] + cs + ["--output-format", "csv", "-d", d, "--"] + cmd,
                                env=env, cwd=os.getcwd(), stdout=logf, stderr=subprocess.STDOUT, text=True)
        timed_out = False
        while proc.poll() is None:
            time.sleep(5)
            el = time.time() - t0
            if int(el) % 60 < 5:
                print("   ... %d s" % el, flush=True)          # heartbeat: a silent GPU command is taken to be hung
            if el > per_set_timeout:
                proc.kill(); proc.wait(); timed_out = True
                break
        logf.close()
        if timed_out:
            print("set %d TIMED OUT after %d s: stopping (no further GPU step after a timeout)" % (i, per_set_timeout), flush=True)
            break

        class R: pass
        r = R(); r.returncode = proc.returncode; r.stdout = open(os.path.join(d, "stdout.log")).read()
        print("   %.1f s" % (time.time() - t0), flush=True)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        print("set %d %s rc=%d files=%d" % (i, cs, r.returncode, len(files)), flush=True)
        if r.returncode != 0 or not files:
            print(r.stdout[-2000:], flush=True)
            continue
        for f in files:
            seen = {}
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0]
                a = agg.setdefault(k, {})
                a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                key = (k, row["Counter_Name"])
                if key not in seen:
                    seen[key] = set()
                seen[key].add(row["Dispatch_Id"])
            for (k, c), ids in seen.items():
                agg[k]["launches"] = max(agg[k].get("launches", 0), len(ids))
    json.dump({"command": " ".join(cmd), "sets": sets, "kernels": agg}, open(out, "w"), indent=1)
    for k, a in agg.items():
        print(k)
        for c, v in sorted(a.items()):
            print("   %-44s %.6g" % (c, v))


if __name__ == "__main__":
    main()
