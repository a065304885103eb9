#!/usr/bin/env python3
"""Collect PMC counters per kernel, one rocprofv3 pass per counter set, and print/write the sums.

Runs on the GPU box:   python3 tools/pmc_pass.py OUT.json SETFILE -- python3 bench.py --no-cpu-baseline ...
SETFILE holds one counter set per line (space separated).  Counters are collected in their own
passes with --kernel-trace only (never combined with sys/hip traces).  This launcher does not touch
the GPU itself; the profiled program is started as a child by rocprofv3.

rocprofv3 runs in its own session: on a timeout the WHOLE process group is killed (the profiled child holds the GPU, not
rocprofv3), the child's output stays under gpurun_out/pmc_logs/ (so a timeout leaves evidence), no further pass is started and
the exit status is non-zero.
"""
import csv, glob, json, os, signal, subprocess, sys, tempfile, time


def main():
    out, setfile = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    sets = [l.split() for l in open(setfile) if l.strip() and not l.startswith("#")]
    agg = {}
    per_set_timeout = int(os.environ.get("PMC_SET_TIMEOUT", "240"))
    env = dict(os.environ, TMPDIR="/tmp")
    logdir = os.path.join(os.getcwd(), "gpurun_out", "pmc_logs")
    os.makedirs(logdir, exist_ok=True)
    tag = os.path.splitext(os.path.basename(out))[0]
    failed = False
    for i, cs in enumerate(sets):
        d = tempfile.mkdtemp(prefix="pmc%d_" % i, dir="/tmp")
        t0 = time.time()
        print("set %d %s ..." % (i, cs), flush=True)
        logpath = os.path.join(logdir, "%s_set%d.log" % (tag, i))
        logf = open(logpath, "w")
        logf.write("# counters: %s\n# command: %s\n" % (" ".join(cs), " ".join(cmd))); logf.flush()
        proc = subprocess.Popen(["rocprofv3", "--kernel-trace", "--pmc"] + cs + ["--output-format", "csv", "-d", d, "--"] + cmd,
                                env=env, cwd=os.getcwd(), stdout=logf, stderr=subprocess.STDOUT, text=True, start_new_session=True)
        timed_out = False
        while proc.poll() is None:
            time.sleep(5)
            el = time.time() - t0
            if int(el) % 60 < 5:
                print("   ... %d s" % el, flush=True)          # heartbeat: a silent GPU command is taken to be hung
            if el > per_set_timeout:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)      # rocprofv3 AND the GPU-holding program it started
                except ProcessLookupError:
                    pass
                proc.wait(); timed_out = True
                break
        logf.close()
        if timed_out:
            print("set %d %s TIMED OUT after %d s (child output: %s): stopping, no further GPU step after a timeout" % (i, cs, per_set_timeout, logpath), flush=True)
            failed = True
            break

        class R: pass
        r = R(); r.returncode = proc.returncode; r.stdout = open(logpath).read()
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
    json.dump({"command": " ".join(cmd), "sets": sets, "kernels": agg, "timed_out": failed}, open(out, "w"), indent=1)
    for k, a in agg.items():
        print(k)
        for c, v in sorted(a.items()):
            print("   %-44s %.6g" % (c, v))
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
