"""HBM traffic of the sweep kernels from the PMC counters, the way MI355X_MICROARCH.md prescribes: one rocprofv3 pass per
counter (FETCH_SIZE, WRITE_SIZE), --kernel-trace only, calibrated on kernels with known byte counts in the same passes.
Writes profiles/<tag>_pmc_traffic.json (read by bench.py, which drops the figure when the kernel sources changed since)
and profiles/<tag>_pmc_traffic.md.
usage (on the GPU box): python tools/pmc_traffic.py TAG MESH [k]     e.g.  python tools/pmc_traffic.py r03_box box:216 2"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "openfoam-2.2.x_amd", "csrc", f) for f in
       ("ldu_kernels.hip", "ldu_cluster.hip", "ldu_internal.hpp", "ldu_plan.cpp")]


def source_hash():
    h = hashlib.sha256()
    for f in SRC:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def run_pass(counter, mesh, k, out):
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", counter.lower(),
           "--", sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py"), mesh, str(k)]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode:
        raise SystemExit("rocprofv3 failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    info = [l for l in r.stdout.splitlines() if l.startswith("pmc_workload")]
    files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    rows = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            rows.setdefault(name, []).append(float(row["Counter_Value"]))
    return rows, (info[-1] if info else "")


def main():
    tag, mesh = sys.argv[1], sys.argv[2]
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    fetch, info = run_pass("FETCH_SIZE", mesh, k, "/tmp/pmc_%s_f" % tag)
    write, _ = run_pass("WRITE_SIZE", mesh, k, "/tmp/pmc_%s_w" % tag)
    nC = int(info.split("nCells")[1].split()[0]); nF = int(info.split("nFaces")[1].split()[0])
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        f = fetch.get(name, [0.0]); w = write.get(name, [0.0])
        # the workload repeats three times: the last third of a kernel's dispatches is one steady-state round
        fl, wl = f[-max(1, len(f) // 3):], w[-max(1, len(w) // 3):]
        kernels[name] = dict(dispatches=len(f), fetch_kib=max(fl), write_kib=max(wl),
                             fetch_kib_all=sorted(set(round(x) for x in fl))[-4:], write_kib_all=sorted(set(round(x) for x in wl))[-4:])
    dom = [n for n in kernels if "gs_multi" in n]
    dom = max(dom, key=lambda n: kernels[n]["fetch_kib"]) if dom else None
    cal = {n: kernels[n] for n in kernels if n.startswith("reduce_partial_kernel") or n.startswith("reciprocal_kernel")}
    out = dict(tag=tag, workload="%s, %d pipelined GaussSeidel sweeps per launch" % (mesh, k), nCells=nC, nFaces=nF,
               source_hash=source_hash(), kernel=dom.split("<")[0].replace("void ", "").strip() if dom else None, kernel_full=dom,
               fetch_kib=kernels[dom]["fetch_kib"] if dom else None, write_kib=kernels[dom]["write_kib"] if dom else None,
               bytes_per_launch=int((kernels[dom]["fetch_kib"] + kernels[dom]["write_kib"]) * 1024) if dom else None,
               # the guide's gfx950 correction made explicit: FETCH_SIZE tallies a 16-byte-per-lane access at half its bytes.
               # The sweep kernels mix 4- and 8-byte streams (counted in full) with 16-byte granule polls (counted at half):
               # the true traffic lies between the raw figure and 2 x FETCH_SIZE + WRITE_SIZE; `corrected` adds the missing
               # half of the granule reads only, estimated as one 16-byte poll per off-diagonal entry of the k sweeps
               # (an upper estimate of the polls that reach memory: neighbours inside a cluster are read from LDS)
               bytes_per_launch_upper=int((2 * kernels[dom]["fetch_kib"] + kernels[dom]["write_kib"]) * 1024) if dom else None,
               bytes_per_launch_corrected=int((kernels[dom]["fetch_kib"] + kernels[dom]["write_kib"]) * 1024
                                              + min(kernels[dom]["fetch_kib"] * 1024, 0.5 * 16.0 * k * 2 * nF)) if dom else None,
               algorithmic_bytes_per_launch=k * (60 * nC + 12 * nF),
               calibration={n: dict(fetch_kib=c["fetch_kib"], write_kib=c["write_kib"], known_read_kib=8.0 * nC / 1024)
                            for n, c in cal.items()},
               kernels=kernels,
               source="tools/pmc_traffic.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (--kernel-trace only) of "
                      "tools/pmc_workload.py; raw (FETCH_SIZE + WRITE_SIZE) * 1024 of the largest steady-state dispatch; 8-B/lane streams "
                      "count 1.00x, 16-B/lane accesses 0.5x on gfx950 (MI355X_MICROARCH.md) - see the calibration kernels",
               note=info)
    pdir = os.path.join(ROOT, "profiles")
    json.dump(out, open(os.path.join(pdir, tag + "_pmc_traffic.json"), "w"), indent=1)
    with open(os.path.join(pdir, tag + "_pmc_traffic.md"), "w") as md:
        md.write("# %s - HBM traffic from PMC counters (rocprofv3, one pass per counter)\n\n%s\n\n" % (tag, info))
        md.write("source hash of the kernel files: %s\n\n| kernel | dispatches | FETCH_SIZE KiB | WRITE_SIZE KiB |\n|---|---|---|---|\n" % out["source_hash"])
        for n, c in kernels.items():
            md.write("| `%s` | %d | %.0f | %.0f |\n" % (n[:90], c["dispatches"], c["fetch_kib"], c["write_kib"]))
        if dom:
            md.write("\ndominant: `%s`: %.3f GB per launch raw against %.3f GB algorithmic (%.2fx); with the guide's 16-byte "
                     "correction applied to the granule polls %.3f GB (%.2fx); upper bound 2 x FETCH + WRITE %.3f GB (%.2fx)\n"
                     % (dom[:80], out["bytes_per_launch"] / 1e9, out["algorithmic_bytes_per_launch"] / 1e9,
                        out["bytes_per_launch"] / out["algorithmic_bytes_per_launch"], out["bytes_per_launch_corrected"] / 1e9,
                        out["bytes_per_launch_corrected"] / out["algorithmic_bytes_per_launch"],
                        out["bytes_per_launch_upper"] / 1e9, out["bytes_per_launch_upper"] / out["algorithmic_bytes_per_launch"]))
    print(json.dumps({k_: out[k_] for k_ in ("tag", "kernel", "bytes_per_launch", "algorithmic_bytes_per_launch", "source_hash")}))


if __name__ == "__main__":
    main()
