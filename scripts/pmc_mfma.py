"""Summarise a rocprofv3 SQ counter pass into profiles/<name>.json: per kernel, matrix-core busy fraction and where the
waves' cycles go.

usage (on the GPU box, after
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
            SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d gpurun_out/pmc_sq -o p --output-format csv
            -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-stage1
):  python scripts/pmc_mfma.py gpurun_out/pmc_sq profiles/r01_pmc_mfma.json
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): the counter adds the busy cycles of the matrix
pipes of all SIMDs (MI355X_MICROARCH.md: = 32 x N_mfma for 32x32x16 bf16), SQ_BUSY_CU_CYCLES the cycles a CU had a wave,
summed over CUs.  wait_any / wait_inst_any / active_inst_any are fractions of SQ_WAVE_CYCLES (quad-cycle units, summed
over waves): parked at s_waitcnt or a barrier / stalled at issue / issuing."""
import collections, csv, glob, json, sys

NAMES = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU_MFMA_MOPS_BF16"]


def main():
    d, out = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:170]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    res = {"note": __doc__.split("mfma_busy_frac")[1].strip(), "kernels": {}}
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CU_CYCLES", 0.0)):
        a = acc[k]
        wc, bc = a.get("SQ_WAVE_CYCLES", 0.0), a.get("SQ_BUSY_CU_CYCLES", 0.0)
        if not wc or not bc:
            continue
        res["kernels"][k] = {
            "launches": len(cnt[k]),
            "mfma_busy_frac": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * bc), 4),
            "wait_any": round(a.get("SQ_WAIT_ANY", 0.0) / wc, 4),
            "wait_inst_any": round(a.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
            "wait_inst_lds": round(a.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4),
            "active_inst_any": round(a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
            "raw": {n: a.get(n) for n in NAMES if n in a}}
    res["note"] = "mfma_busy_frac " + res["note"]
    json.dump(res, open(out, "w"), indent=1)
    for k, v in list(res["kernels"].items())[:8]:
        print(k[:80], {x: y for x, y in v.items() if x != "raw"})


if __name__ == "__main__":
    main()
