"""Turn the FETCH_SIZE / WRITE_SIZE passes (tools/hbm_traffic.sh) into profiles/hbm_traffic.json.

Counter units and gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide (16 B per lane) streaming read, so the read side is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def per_kernel(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}
fetch, write = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
res = {"note": "KiB counters -> bytes; FETCH_SIZE doubled (gfx950 wide-read correction); per launch, 25,000,000 elements",
       "kernels": {}}
tot = 0.0
for k in sorted(set(fetch) | set(write)):
    if "gq::" not in k:
        continue
    rd, wr = fetch.get(k, 0.0) * 1024 * 2, write.get(k, 0.0) * 1024
    res["kernels"][k] = {"fetch_size_raw_kib": fetch.get(k), "write_size_raw_kib": write.get(k),
                         "hbm_read_bytes": rd, "hbm_write_bytes": wr}
    if "hsq_encode" in k:
        tot += rd + wr
res["hsq_encode_hbm_bytes_per_launch"] = tot
res["algorithmic_bytes_per_launch"] = 4.125 * 25_000_000
json.dump(res, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
