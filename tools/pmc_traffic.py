"""Parses rocprofv3 --pmc counter CSVs (FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes, as
MI355X_MICROARCH.md prescribes) into per-launch HBM traffic for the dominant kernel class.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

gfx950 corrections applied: counters are in KiB (x1024); FETCH_SIZE reports exactly half the bytes of a wide coalesced
stream (16 B/lane loads and LDS-DMA alike) so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirpath, counter):
    per = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r.get("Kernel_Name", "")
            per[name][0] += 1
            per[name][1] += float(r["Counter_Value"])
    return per


def klass(name):
    if "kconv_kernel" in name or "dit_panel_kernel" in name:    # resident-tile convs and the fused row-panel DiT kernel are
        return "kgemm_f16"                                      # fp16 MFMA GEMM work (same class as in bench.py's timing)
    if "kgemm_kernel" in name:
        return "kgemm_f16" if ("DF16_" in name or "_Float16" in name) else "kgemm_f32"
    if "attn_kernel" in name:
        return "attention"
    return None


def calib(dirpath, counter, known_bytes=float(1 << 30)):
    """known bytes / (counter x 1024) for the largest dispatch of a tools/pmc_calib.py run (None when not collected)."""
    if not dirpath or not os.path.isdir(dirpath):
        return None
    best = 0.0
    for f in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                best = max(best, float(r["Counter_Value"]))
    return round(known_bytes / (best * 1024), 4) if best > 0 else None


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "tiny-b64"
    commit = sys.argv[5] if len(sys.argv) > 5 else None
    cal_f = sys.argv[6] if len(sys.argv) > 6 else None
    cal_w = sys.argv[7] if len(sys.argv) > 7 else None
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    # per-kernel aggregate kept next to the JSON (the raw per-dispatch CSVs are too large to commit)
    with open(os.path.splitext(out)[0] + "_per_kernel.csv", "w") as fcsv:
        fcsv.write("kernel,class,dispatches,FETCH_SIZE_KiB_sum,WRITE_SIZE_KiB_sum\n")
        for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
            fcsv.write(f'"{k[:160]}",{klass(k)},{max(fe.get(k, [0, 0])[0], wr.get(k, [0, 0])[0])},{fe.get(k, [0, 0])[1]:.0f},{wr.get(k, [0, 0])[1]:.0f}\n')
    res = {}
    for cls in ("kgemm_f16", "kgemm_f32", "attention"):
        nf = sum(v[0] for k, v in fe.items() if klass(k) == cls)
        bf = sum(v[1] for k, v in fe.items() if klass(k) == cls) * 1024 * 2      # KiB, x2 gfx950 correction
        nw = sum(v[0] for k, v in wr.items() if klass(k) == cls)
        bw = sum(v[1] for k, v in wr.items() if klass(k) == cls) * 1024
        if nf and nw:
            res[cls] = {"launches": nf, "fetch_bytes_per_launch": bf / nf, "write_bytes_per_launch": bw / nw,
                        "hbm_bytes_per_launch": bf / nf + bw / nw}
    res["workload"] = workload
    res["commit"] = commit
    # FETCH_SIZE x 2 / WRITE_SIZE x 1 are MI355X_MICROARCH.md's gfx950 corrections; the calibration run (1 GiB elementwise
    # pass, tools/pmc_calib.py, same counters) measures them on this box: known bytes / counter
    res["calibration"] = {"fetch_known_over_counter": calib(cal_f, "FETCH_SIZE"), "write_known_over_counter": calib(cal_w, "WRITE_SIZE"),
                          "applied": {"fetch": 2.0, "write": 1.0}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
