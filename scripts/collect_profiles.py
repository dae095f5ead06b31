#!/usr/bin/env python
"""Turns the files a scripts/gpu_final.sh run merged into gpurun_out/ into the tracked summaries under profiles/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import rocpd_stats  # noqa: E402


def main(tag, version):
    g = os.path.join(ROOT, "gpurun_out")
    p = os.path.join(ROOT, "profiles")
    rocpd_stats.main(os.path.join(g, "%s_prof" % tag, "scr_results.db"), os.path.join(p, "r1_scr_kernel_stats_%s.csv" % version))
    db1 = os.path.join(g, "%s_prof1" % tag, "scr_results.db")
    if os.path.isfile(db1):
        rocpd_stats.main(db1, os.path.join(p, "r1_scr_kernel_stats_%s_single_stream.csv" % version))
    for w in ("scr", "aser", "er", "mir"):
        src = os.path.join(g, "%s_bench_%s.log" % (tag, w))
        if os.path.isfile(src):
            line = [l for l in open(src) if l.startswith("{")][-1]
            open(os.path.join(p, "r1_bench_%s_%s.json" % (version, w)), "w").write(line)
    d = json.load(open(os.path.join(g, "%s_pmc_summary.json" % tag)))

    def agg(tag_, prefix):
        s = n = 0
        for k, v in d.get(tag_, {}).items():
            if prefix in k.split("|")[0]:
                s += v["sum"]; n += v["n"]
        return s, n
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --steps 10 --warmup 3 "
                     "--no-cpu-baseline --no-roofline (MI355X, scripts/gpu_final.sh)",
           "units": "FETCH_SIZE / WRITE_SIZE are KiB; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B "
                    "request for 16 B/lane streaming reads -> doubled; WRITE_SIZE uncalibrated, taken as is",
           "kernels": {}}
    for prefix, label in (("conv_gemm_kernel", "conv_gemm_kernel"), ("conv_wgrad_kernel", "conv_wgrad_kernel"), ("bn_fwd_kernel", "bn_fwd_kernel"),
                          ("bn_bwd_reduce", "bn_bwd_reduce_kernel"), ("bn_bwd_apply", "bn_bwd_apply_kernel"),
                          ("rows_copy16<false>", "rows_copy16_gather")):
        fs, fn = agg("FETCH_SIZE", prefix)
        ws, wn = agg("WRITE_SIZE", prefix)
        if fn and wn:
            out["kernels"][label] = dict(launches_profiled=fn, fetch_kib_per_launch=fs / fn, write_kib_per_launch=ws / wn,
                                         hbm_bytes_per_launch=(2 * fs / fn + ws / wn) * 1024)
    json.dump(out, open(os.path.join(p, "r1_scr_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in out["kernels"].items()}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
