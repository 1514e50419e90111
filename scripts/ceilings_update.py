"""Writes the `CEILINGS {...}` line of a scripts/bin/hbm_ceilings run into profiles/traffic.json (`_measured_peaks_gbps`):
the recorded fallback bench.py uses when it cannot measure the ceilings on its own box (N > 1, helper library missing).
Usage: python scripts/ceilings_update.py gpurun_out/.../hbm_ceilings_1gib.txt [box label]"""
import datetime
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(sys.argv[1]).read()
line = [l for l in src.splitlines() if l.startswith("CEILINGS ")][-1]
c = json.loads(line[len("CEILINGS "):])
g = c["gbps"]


def best(key):  # the better of the run's two repetitions, settled figure
    return max(g[key]["settled"], g[key + "_again"]["settled"])


def by_workload(k):
    """The envelope bench.py computes on its own box: trivial mixes and the serial sum of the best pure streams."""
    n = c["gib"] * (1 << 30)
    w = g[k + "_plain_nt"]["w_per_r"] * n
    r = {"w_per_r": g[k + "_plain_nt"]["w_per_r"], "same_mix_plain_nt": best(k + "_plain_nt"), "same_mix_nt_nt": best(k + "_nt_nt"),
         "same_mix_deferred_nt_nt": best(k + "_deferred_nt_nt"), "write_only_nt": best(k.replace("mix", "write") + "_nt")}
    r["write_only_peak"] = max(r["write_only_nt"], best("write4_plain"))  # the pure write rate of a long launch (4 N)
    t_sum = n / max(best("read_plain"), best("read_nt")) + w / r["write_only_peak"]
    r["serial_sum_of_pure_streams"] = round((n + w) / t_sum, 1)
    r["same_mix_best"] = max(r["same_mix_plain_nt"], r["same_mix_nt_nt"], r["same_mix_deferred_nt_nt"], r["serial_sum_of_pure_streams"])
    return r


mix = {"minified": "mix078", "utf8": "mix041", "pretty4": "mix041", "pretty2": "mix041", "pretty8": "mix041",
       "pretty_tab_crlf": "mix041"}
rec = {
    "date": datetime.date.today().isoformat(),
    "box": sys.argv[2] if len(sys.argv) > 2 else c.get("device"),
    "source": "scripts/ubench/hbm_ceilings.hip sha256:" +
              hashlib.sha256(open(os.path.join(ROOT, "scripts", "ubench", "hbm_ceilings.hip"), "rb").read()).hexdigest()[:12],
    "gib": c["gib"], "grid": c["grid"],
    "_note": "trivial kernels on the product's persistent grid, tile walk and store shape (whole-line non-temporal stores), "
             "400 back-to-back launches behind 400 untimed ones; GB/s of (bytes read + bytes written); the better of two "
             "repetitions; the serial sum takes the best pure read and the pure write rate of a 4 N launch.  bench.py "
             "measures the same on its own box when it can.",
    "read_plain": best("read_plain"), "read_nt": best("read_nt"),
    "by_workload": {w: by_workload(k) for w, k in mix.items()},
    "table": {k: v["settled"] for k, v in g.items()},
}
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path))
tj["_measured_peaks_gbps"] = rec
json.dump(tj, open(path, "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("date", "box", "read_plain", "read_nt", "by_workload")}, indent=1))
