"""Summarise two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md §HBM) into
profiles/<name>.json: per-kernel average beyond-L2 traffic per launch.

usage (on the GPU box, after
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-stage1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o p --output-format csv -- python3 bench.py ... (same)
):  python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic_v2.json
traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts 128-byte requests of wide coalesced reads as 64 B
on gfx950; both counters sit on the fabric side of the L2, so Infinity-Cache hits are included (upper bound on HBM)."""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"][:120]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    res = {"note": __doc__.split("traffic_bytes")[1].strip(), "kernels": {}}
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        n, f = fe[k]
        w = wr.get(k, [1, 0.0])
        favg, wavg = f / max(n, 1), w[1] / max(w[0], 1)
        res["kernels"][k] = {"launches": n, "FETCH_SIZE_KB_avg": round(favg, 1), "WRITE_SIZE_KB_avg": round(wavg, 1),
                             "traffic_bytes_per_launch": int((2 * favg + wavg) * 1024)}
    res["note"] = "traffic_bytes " + res["note"]
    json.dump(res, open(out, "w"), indent=1)
    for k, v in list(res["kernels"].items())[:8]:
        print(k[:90], v)


if __name__ == "__main__":
    main()
