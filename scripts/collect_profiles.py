#!/usr/bin/env python
"""Turns the files a scripts/gpu_final.sh run merged into gpurun_out/ into the tracked summaries under profiles/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import rocpd_stats  # noqa: E402


def _db(d, name):
    """rocprofv3 nests its output under the host name: find <name>_results.db anywhere below d."""
    for dp, _, fs in os.walk(d):
        for f in fs:
            if f == name + "_results.db":
                return os.path.join(dp, f)
    return None


def main(tag, version, rnd="r3"):
    g = os.path.join(ROOT, "gpurun_out")
    p = os.path.join(ROOT, "profiles")
    for sub, name, suffix in (("prof", "scr", ""), ("prof1", "scr", "_single_stream"), ("prof2", "aser", "_single_stream"), ("prof3", "er", "_single_stream")):
        db = _db(os.path.join(g, "%s_%s" % (tag, sub)), name)
        if db:
            rocpd_stats.main(db, os.path.join(p, "%s_%s_kernel_stats_%s%s.csv" % (rnd, name, version, suffix)))
    # (round 6, scripts/gpu_r6_final.sh: the kernel statistics are written by the call itself; copy them under the round's names)
    for name in ("scr", "scr_single_stream", "aser_single_stream", "er_single_stream"):
        src = os.path.join(g, "%s_%s_kernel_stats.csv" % (tag, name))
        if os.path.isfile(src):
            w_, _, suffix = name.partition("_")
            open(os.path.join(p, "%s_%s_kernel_stats_%s%s.csv" % (rnd, w_, version, "_" + suffix if suffix else "")), "w").write(open(src).read())
    # (the MFMA-pipe counters: scripts/pmc_mfma.py's table, re-expressed per SIMD / per XCD as in profiles/r5_scr_pmc_mfma.txt, by hand into
    #  profiles/<round>_scr_pmc_mfma.txt)
    for w in ("scr", "aser", "er", "mir", "default"):
        src = os.path.join(g, "%s_bench_%s.log" % (tag, w))
        if os.path.isfile(src):
            line = [l for l in open(src) if l.startswith("{")][-1]
            open(os.path.join(p, "%s_bench_%s_%s.json" % (rnd, version, w)), "w").write(line)
    if not os.path.isfile(os.path.join(g, "%s_pmc_summary.json" % tag)):
        print("no PMC summary for", tag)
        return
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
    for prefix, label in (("conv_t_kernel", "conv_t_kernel"), ("conv_q_kernel", "conv_q_kernel"), ("conv_s_kernel", "conv_s_kernel"), ("conv_wx_kernel", "conv_wx_kernel"),
                          ("conv_wgrad_kernel", "conv_wgrad_kernel"), ("conv_wgrad_multi_kernel", "conv_wgrad_multi_kernel"), ("bn_fwd_kernel", "bn_fwd_kernel"), ("bn_bwd_fused", "bn_bwd_fused_kernel"), ("wgrad_reduce", "wgrad_reduce_kernel"),
                          ("bn_bwd_reduce", "bn_bwd_reduce_kernel"), ("bn_bwd_apply_kernel", "bn_bwd_apply_kernel"), ("bn_bwd_apply_e", "bn_bwd_apply_e_kernel"),
                          ("rows_copy16<false>", "rows_copy16_gather")):
        fs, fn = agg("FETCH_SIZE", prefix)
        ws, wn = agg("WRITE_SIZE", prefix)
        if fn and wn:
            out["kernels"][label] = dict(launches_profiled=fn, fetch_kib_per_launch=fs / fn, write_kib_per_launch=ws / wn,
                                         hbm_bytes_per_launch=(2 * fs / fn + ws / wn) * 1024)
    # the convolution forward / data-gradient class of the bench's roofline (conv_t_kernel + conv_q_kernel + conv_s_kernel launches together)
    ks = [out["kernels"][k] for k in ("conv_t_kernel", "conv_q_kernel", "conv_s_kernel", "conv_wx_kernel") if k in out["kernels"]]
    if ks:
        nl = sum(k["launches_profiled"] for k in ks)
        out["kernels"]["conv_fwd_dgrad"] = dict(launches_profiled=nl, hbm_bytes_per_launch=sum(k["hbm_bytes_per_launch"] * k["launches_profiled"] for k in ks) / nl)
    json.dump(out, open(os.path.join(p, "%s_scr_pmc_traffic.json" % rnd), "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in out["kernels"].items()}))


if __name__ == "__main__":
    main(*sys.argv[1:4])
