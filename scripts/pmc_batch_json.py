"""Build profiles/r01d_pmc_pyramid_batch.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
scripts/pmc_probe_batch.py:  python scripts/pmc_batch_json.py DIR_FETCH DIR_WRITE S OUT.json"""
import csv, glob, json, sys, collections

def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[f'{r["Kernel_Name"].split("(")[0].replace("void ", "")}[grid={r["Grid_Size"]}]'].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in sorted(acc.items())}

fd, wd, S, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
commit = sys.argv[5] if len(sys.argv) > 5 else "unrecorded"
ingest = sys.argv[6] if len(sys.argv) > 6 else "f64"
F, Wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
FC = 2.0     # profiles/r01_pmc_pyramid.json: k_copy8 calibration, 8 B/lane coalesced reads report half in FETCH_SIZE
tot = lambda name: sum((F.get(k, 0) * FC + Wr.get(k, 0)) * 1024 for k in set(F) | set(Wr) if k.startswith(name))
rows = tot("k_iir_rows")
n_rows = len([k for k in F if k.startswith("k_iir_rows")])
allk = sum((F.get(k, 0) * FC + Wr.get(k, 0)) * 1024 for k in set(F) | set(Wr) if k.startswith("k_"))
json.dump({"streams": S, "commit": commit, "ingest": ingest, "FETCH_SIZE": {"per_kernel_avg_KB": F}, "WRITE_SIZE": {"per_kernel_avg_KB": Wr},
           "summary": {"fetch_correction": FC, "write_correction": 1.0,
                       "k_iir_rows_bytes_per_batch_build": rows, "k_iir_rows_bytes_per_launch": rows / max(n_rows, 1),
                       "all_pyramid_kernels_bytes_per_batch_build": allk, "all_pyramid_kernels_bytes_per_frame": allk / S,
                       "note": f"{S} images of 370x1226 ({ingest} ingest) per launch (slam_pyr_update_batch[_u8]_dev, update! semantics, serial launches SLAMHIP_NO_GRAPH=1); "
                               "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only; FETCH_SIZE x2 "
                               "(k_copy8 calibration in r01_pmc_pyramid.json), WRITE_SIZE exact; per-kernel values are averages over the launches, KB of 1024 B"}},
          open(outp, "w"), indent=1)
print(json.dumps(json.load(open(outp))["summary"], indent=1))
