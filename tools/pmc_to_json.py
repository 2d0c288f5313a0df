#!/usr/bin/env python3
"""HBM bytes per launch of the step's GEMM kernels from the rocprofv3 PMC passes of the bench command.

usage: pmc_to_json.py <dir with w/ and f/ counter dirs> <bench json log of the same configuration> > profiles/rNN/pmc_hbm.json
WRITE_SIZE and FETCH_SIZE are in KB, collected in SEPARATE passes (TCC slot budget); per MI355X_MICROARCH.md (HBM) FETCH_SIZE
reports half of a wide coalesced read stream on gfx950 (doubled here), WRITE_SIZE is exact for 16-byte streaming stores."""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha():
    """Fingerprint of the kernel sources the counters were collected on (bench.py refuses a pmc_hbm.json taken on other sources)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gdrf_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]

# timing slot -> substrings of the kernel names that can fill it (the first kernel found wins; one of them runs per configuration)
SLOT = {"bwd_wbar": ("bwd_wbar_f16_k64_kernel", "bwd_wbar_split"), "fwd_t": ("fwd_t_split",), "tn_sym": ("tn_topics_w2_kernel", "tn_topics_w1_kernel", "tn_topics_f16_kernel"),
        "tn_gt": ("gemm_tn_split_kernel",), "fwd_w": ("FwdWProb",), "bwd_knm": ("BwdKnmProb",),
        "k_nm": ("knm_rbf_f64_kernel", "knm_kernel<double"), "k_nm_f32": ("knm_kernel<float",), "elbo_rows": ("elbo_rows_mfma_kernel", "elbo_rows_kernel"),
        "loc": ("loc_rows_kernel",), "predict": ("predict_mfma_kernel",)}


def means(root, counter):
    val = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if row["Counter_Name"] == counter:
                val[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in val.items()}


def main():
    root, bench_log = sys.argv[1], sys.argv[2]
    d = json.loads([l for l in open(bench_log) if l.startswith("{")][-1])
    w, f = means(os.path.join(root, "w"), "WRITE_SIZE"), means(os.path.join(root, "f"), "FETCH_SIZE")
    out = {"N": d["config"]["N"], "mfma_mode": d["config"]["mfma_mode"], "csrc_sha": csrc_sha(), "kernels": {},
           "note": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes of `bench.py --steps 1`, mean per dispatch; "
                   "traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE counts half of a wide read stream on gfx950)"}
    for slot, pats in SLOT.items():
        for pat in pats:
            kw = [(k, v) for k, v in w.items() if pat in k]
            kf = [v for k, v in f.items() if pat in k]
            if kw and kf:
                # a slot can be several launches per step (A_k: tn_topics_w2_kernel<0> and <4>): their bytes add
                wsum, fsum = sum(v for _, v in kw), sum(kf)
                out["kernels"][slot] = {"kernel": " + ".join(sorted(k.split("(")[0][:96] for k, _ in kw)), "write_bytes": wsum * 1024,
                                        "fetch_bytes_corrected": 2 * fsum * 1024, "traffic_bytes": (2 * fsum + wsum) * 1024}
                break
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
