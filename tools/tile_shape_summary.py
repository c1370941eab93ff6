#!/usr/bin/env python3
"""Summarise tools/tile_shape_probe runs taken under tools/power_sample.sh: adds the socket power (median of the samples above 80 % of the
run's maximum: the kernel loop, not the image generation) and rocm-smi's shader clock to each run's JSON line and prints one row per run.
  tile_shape_summary.py <run.json> [...]      (the samples are expected beside each as <run>.smi.jsonl)"""
import json
import re
import statistics
import sys


def smi_of(path):
    pw, ck = [], []
    try:
        for line in open(path):
            m = re.search(r'Current Socket Graphics Package Power \(W\)": "([0-9.]+)"', line)
            c = re.search(r'sclk clock level": "[^"]*\((\d+)Mhz\)', line)
            if m and c:
                pw.append(float(m.group(1)))
                ck.append(float(c.group(1)))
    except OSError:
        pass
    hot = [i for i, p in enumerate(pw) if p > 0.8 * max(pw)] if pw else []
    return (statistics.median(pw[i] for i in hot), statistics.median(ck[i] for i in hot), len(hot)) if hot else (None, None, 0)


def main():
    for path in sys.argv[1:]:
        try:
            d = json.loads(open(path).read().strip().splitlines()[-1])
        except Exception as ex:
            print(path, "??", ex)
            continue
        w, c, n = smi_of(path[:-5] + ".smi.jsonl")
        d["socket_power_w_median"], d["smi_sclk_mhz_median"], d["smi_samples"] = w, c, n
        open(path, "w").write(json.dumps(d) + "\n")
        print("%-6s %d waves x %-6s tile %-7s vgprs %3d scratch %4d B | %7.3f ms | %.3f PFLOP/s = %.4f of the FP4 peak | %5.2f ns per MFMA and CU | %s W, %s MHz (smi), %.0f MHz (workgroup 0) | "
              "L2 -> LDS %.1f TB/s | checksum %s" % (d["shape"], d["waves"], d["products_per_wave"], d["tile"], d["vgprs"], d["scratch_bytes"], d["ms_per_launch"], d["pflops"], d["of_fp4_peak"],
                                                     d["ns_per_mfma_per_cu"], w, c, d["shader_clock_mhz_wg0"], d["l2_to_lds_tb_s"], "ok" if d["checksum_ok"] else "WRONG"))


if __name__ == "__main__":
    main()
