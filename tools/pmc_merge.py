"""Merge the per-pass summaries of tools/pmc_collect.sh (OUTDIR/<pass>.json) into the two files the
repository keeps per round:

    python tools/pmc_merge.py OUTDIR profiles/rN_pmc_hbm_bytes.json profiles/rN_pmc_sq.json [frames]

hbm file: per sweep kernel FETCH_SIZE / WRITE_SIZE (KiB per launch as rocprofv3 reports them), the
bytes per launch that follow (on gfx950 FETCH_SIZE counts a 128-byte read request as 64 bytes, which
the tcc2 pass shows: TCC_EA0_RDREQ x 64 B = FETCH_SIZE; so bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE *
1024), and the L2 request counters.  sq file: the SQ counters of the same kernels."""
import json
import os
import sys


def load(d, name):
    with open(os.path.join(d, name + ".json")) as fh:
        return json.load(fh)


def main(outdir, hbm_path, sq_path, frames=30):
    fetch, write = load(outdir, "fetch"), load(outdir, "write")
    tcc = [load(outdir, n) for n in ("tcc1", "tcc2", "tcc3")]
    sq = [load(outdir, n) for n in ("sq1", "sq2", "sq3")]
    def strip(d):       # newer rocprofv3 versions print the return type in front of templated kernel names
        return {(k[5:] if k.startswith("void ") else k): v for k, v in d.items()}
    fetch, write, tcc, sq = strip(fetch), strip(write), [strip(t) for t in tcc], [strip(t) for t in sq]
    kernels = sorted(k for k in fetch if k.startswith("k_primal") or k.startswith("k_dual"))
    hbm = {"_comment": "rocprofv3 --pmc passes (one counter set per run, tools/pmc_collect.sh, merged by tools/pmc_merge.py) over "
                       "tools/pmc_target.py: the bench workload (HR frames 1080x1920, IC model, inputs rendered on the host). "
                       "FETCH_SIZE / WRITE_SIZE are KiB per launch as reported; on gfx950 FETCH_SIZE counts 128-byte read requests "
                       "as 64 bytes, so HBM bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.  Infinity-Cache hits are "
                       "included (the counters sit on the L2's fabric side).",
           "workload": {"frames": int(frames), "hr": [1080, 1920], "model": "ic"}, "kernels": {}}
    sqo = {"_comment": "rocprofv3 --pmc SQ passes sq1/sq2/sq3 of tools/pmc_collect.sh over tools/pmc_target.py (the bench workload); "
                       "mean per launch, summed over all shader engines", "kernels": {}}
    for k in kernels:
        f, w = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
        e = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "launches": fetch[k]["launches"],
             "hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0, "tcc": {}}
        for t in tcc:
            e["tcc"].update({c: v for c, v in t.get(k, {}).items() if c != "launches"})
        hbm["kernels"][k] = e
        s = {}
        for t in sq:
            s.update(t.get(k, {}))
        sqo["kernels"][k] = s
    for path, obj in ((hbm_path, hbm), (sq_path, sqo)):
        with open(path, "w") as fh:
            json.dump(obj, fh, indent=1, sort_keys=True)
            fh.write("\n")


if __name__ == "__main__":
    main(*sys.argv[1:])
