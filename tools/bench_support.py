"""tools/bench_support.py -- the parts of bench.py that are not the timed step: the rocm-smi sampler, the measured end-to-end runs of
both binaries on a chr22-sized fileset (fixed-width, the reference's default variable-width .pgen, `plink2-hip --gpus N`), and the
in-run PMC passes behind `roofline.traffic`.  bench.py imports this; nothing here runs inside a timed region, and nothing under
oracle/ is touched except oracle/_ref/plink2 -- the reference binary -- as the CPU baseline (`cpu_baseline.kind` "reference")."""
import csv
import glob
import json
import os
import re
import shutil
import signal
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(REPO, "oracle", "_ref", "plink2")
CLI_BIN = os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")


class SmiSampler:
    """Socket power (W) and shader clock (MHz) as `rocm-smi -P -g --json` reports them, polled on a thread as fast as it answers (a few
    samples per second) while a tagged window is open: tells a kernel at the socket's power cap (clock pulled below 2.4 GHz) from an
    issue- or latency-bound one.  Reported, never used for a decision."""

    def __init__(self):
        self.samples, self.window, self._stop = [], None, False
        self.cap_w = None
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-M", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20).stdout
            card = next(iter(json.loads(out).values()))
            caps = [float(v) for k, v in card.items() if "(W)" in k]
            self.cap_w = caps[0] if caps else None
        except Exception:
            pass
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def _run(self):
        while not self._stop:
            if self.window is None:
                time.sleep(0.02)
                continue
            try:
                out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-P", "-g", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10).stdout
                card = next(iter(json.loads(out).values()))
                watts = [float(v) for k, v in card.items() if "ower" in k and "(W)" in k]
                mhz = [float(mt.group(1)) for k, v in card.items() if "sclk" in k.lower() for mt in [re.search(r"(\d+)\s*Mhz", str(v))] if mt]
                if self.window is not None:
                    self.samples.append((self.window, watts[0] if watts else None, mhz[0] if mhz else None))
            except Exception:
                time.sleep(0.05)

    def stop(self):
        self._stop = True

    def summary(self, tag):
        rows = [r for r in self.samples if r[0] == tag]
        w = sorted(r[1] for r in rows if r[1] is not None)
        c = sorted(r[2] for r in rows if r[2] is not None)
        med = lambda v: v[len(v) // 2] if v else None
        return {"source": "rocm-smi -P -g polled during the timed steps (%d samples)" % len(rows), "socket_power_w_median": med(w), "socket_power_w_max": w[-1] if w else None,
                "socket_power_cap_w": self.cap_w, "shader_clock_mhz_median": med(c), "shader_clock_mhz_min": c[0] if c else None, "shader_clock_mhz_max": c[-1] if c else None}


def _scratch_dir(need_bytes):
    for d in ("/dev/shm", tempfile.gettempdir()):
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize > need_bytes:
                return d
        except OSError:
            continue
    return None


def write_fixed_width_fileset(pkg, torch, directory, samples, variants, seed, chr_idx, bps, missing_rate=0.0, name="g"):
    """`variants` x `samples` of the device generator as a fixed-width .pgen (storage mode 0x02, pgenlib_read.cc:881-911) + .pvar + .psam
    under `directory`; returns the .pgen's size in bytes."""
    stride = (samples + 3) // 4
    rows_per = max(1, (1 << 30) // stride)
    dev = torch.empty((rows_per, stride), dtype=torch.uint8, device="cuda")
    pin = [torch.empty((rows_per, stride), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    with open(os.path.join(directory, name + ".pgen"), "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x02]) + np.uint32(variants).tobytes() + np.uint32(samples).tobytes() + bytes([0x40]))
        k = 0
        for r0 in range(0, variants, rows_per):
            cnt = min(rows_per, variants - r0)
            pkg.synth_genotypes_device(seed, r0, cnt, samples, missing_rate, dev.data_ptr(), stride)
            torch.cuda.synchronize()
            pin[k & 1][:cnt].copy_(dev[:cnt])
            torch.cuda.synchronize()
            f.write(memoryview(pin[k & 1].numpy()[:cnt]))
            k += 1
    del dev, pin
    torch.cuda.empty_cache()
    with open(os.path.join(directory, name + ".pvar"), "w") as f:
        f.write("#CHROM\tPOS\tID\tREF\tALT\n" + "".join("%d\t%d\tsnp%d\tA\tC\n" % (chr_idx[i] + 1, bps[i], i) for i in range(variants)))
    with open(os.path.join(directory, name + ".psam"), "w") as f:
        f.write("#IID\tSEX\n" + "".join("s%d\t2\n" % q for q in range(samples)))
    return 12 + variants * stride


_PHASES = re.compile(r"setup\+parse ([0-9.]+) s \| genotype load[^|]*?([0-9.]+) s \| run ([0-9.]+) s \(pair kernel ([0-9.]+) ms, replay ([0-9.]+) ms; (\d+) candidate pairs\)")


def run_plink2_hip(directory, pfile, kb, r2, out, gpus=1, extra=(), runs=2, timeout_s=900):
    """plink2-hip end to end, `runs` times (page cache warm, as the reference's own run had it); returns the walls, the return code and the
    --timing phase split and raw [timing] lines of the FASTEST run (the one whose wall is reported as wall_s)."""
    walls, rc, txt, best = [], None, "", None
    for _ in range(runs):
        t1 = time.perf_counter()
        cc = subprocess.run([CLI_BIN, "--pfile", pfile, "--indep-pairwise", kb, repr(r2), "--timing", "--out", out] + (["--gpus", str(gpus)] if gpus > 1 else []) + list(extra),
                            cwd=directory, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
        walls.append(time.perf_counter() - t1)
        rc = cc.returncode
        if rc != 0:
            txt = cc.stdout
            break
        if best is None or walls[-1] < best:   # (the phase split reported is that of the run whose wall is reported)
            best, txt = walls[-1], cc.stdout
    ph = _PHASES.search(txt)
    tot = re.search(r"\[timing\] total ([0-9.]+) s", txt)
    phases = None
    if ph:
        phases = {"setup_and_table_parse_s": float(ph.group(1)), "file_to_hbm_s": float(ph.group(2)), "run_s": float(ph.group(3)), "pair_kernels_ms": float(ph.group(4)),
                  "host_replay_ms": float(ph.group(5)), "candidate_pairs": int(ph.group(6)), "main_total_s": float(tot.group(1)) if tot else None}
    return {"wall_s": min(walls) if walls else None, "wall_s_runs": walls, "rc": rc, "phases": phases,
            "timing_lines": [ln for ln in txt.splitlines() if ln.startswith("[timing]")][:40], "tail": txt[-300:] if rc else ""}


class E2EChr22:
    """BASELINE.json's second metric -- `--indep-pairwise` WALL-CLOCK -- measured, not extrapolated, at the metric's sample count on the
    largest fileset SURVEY 8(d) allows to be materialised: a chr22-sized share of the metric's genome (176,765 of 10,000,000 variants x
    500,000 samples, 22 chromosomes at the metric's density), written by the device generator as a fixed-width .pgen (22 GB, page cache
    or tmpfs).

    start() materialises it and starts the reference (all host threads it can use) in the background, so that its minutes run beside
    the GPU legs of the caller: once on the fixed-width file, and -- `variable_width` -- a second process that first converts the
    fileset to the reference's DEFAULT variable-width .pgen (`--make-pgen`: what a user's files look like) and then prunes that.
    finish() joins them, runs plink2-hip on the same files with the GPU idle (`--timing`: its own phase split; variable-width records
    are decoded on the device, so fewer bytes cross PCIe), and compares every pair of output files byte for byte.
    `gpus` > 1 (bench.py --gpus N, rank 0): plink2-hip --gpus N on the fixed-width file, no reference run."""

    def __init__(self, pkg, torch, cfg, variants, seed, layout, ref_timeout_s=420, missing_rate=0.0):
        self.pkg, self.torch, self.cfg, self.m, self.seed, self.layout, self.ref_timeout_s = pkg, torch, cfg, variants, seed, layout, ref_timeout_s
        self.missing_rate = missing_rate   # (tools/e2e_chr22_missing.py: the same fileset with missing calls in every variant)
        self.tmp, self.ref_proc, self.var_proc, self.res = None, None, None, {}

    def start(self, reference=True, variable_width=True):
        pkg, torch, cfg, m = self.pkg, self.torch, self.cfg, self.m
        n = cfg["samples"]
        stride = (n + 3) // 4
        need = m * stride * (2.1 if variable_width else 1.15) + 2e9
        if reference and not (os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)):
            self.res = {"skipped": "oracle/_ref/plink2 not built"}
            return self
        where = _scratch_dir(need)
        if where is None:
            self.res = {"skipped": "no %.0f GB of scratch space for the fileset" % (need / 1e9)}
            return self
        self.tmp = tempfile.mkdtemp(prefix="ldbench_e2e_", dir=where)
        chr_idx, bps = self.layout(m, 1, cfg["spacing"])
        t0 = time.perf_counter()
        self.file_bytes = write_fixed_width_fileset(pkg, torch, self.tmp, n, m, self.seed, chr_idx, bps, missing_rate=self.missing_rate)
        self.res = {"variants": m, "samples": n, "fileset": "fixed-width .pgen + .pvar + .psam under %s (%.1f GB, written by the device generator in %.1f s)" %
                    (where, self.file_bytes / 1e9, time.perf_counter() - t0)}
        self.cores = os.cpu_count() or 1
        self.kb = "%gkb" % cfg["window_kb"]
        # (every background run is one shell that stamps its own end -- `date +%s.%N` -- so that its wall does not depend on when finish() comes to look)
        if reference:
            prune = "%s --pfile %%s --indep-pairwise %s %r --threads %d --out %%s" % (REF_BIN, self.kb, cfg["r2"], self.cores)
            self.ref_t0 = time.time()
            self.ref_proc = subprocess.Popen(["bash", "-c", (prune % ("g", "ref")) + " > ref.log 2>&1; echo $? > ref.rc; date +%s.%N > ref.end"], cwd=self.tmp, start_new_session=True)
            if variable_width:
                sh = ("%s --pfile g --make-pgen --threads %d --out v > mk.log 2>&1 && date +%%s.%%N > v.stamp && " % (REF_BIN, min(self.cores, 64)) + (prune % ("v", "vref")) +
                      " > vref.log 2>&1; echo $? > v.rc; date +%s.%N > v.end")
                self.var_t0 = time.time()
                self.var_proc = subprocess.Popen(["bash", "-c", sh], cwd=self.tmp, start_new_session=True)
        return self

    def _join(self, proc, deadline):
        """wait for a background shell (until `deadline`, time.time()); False: it was killed"""
        try:
            proc.wait(timeout=max(1.0, deadline - time.time()))
            return True
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)   # (the shell and its child: each background shell leads a process group of its own)
            except OSError:
                proc.kill()
            try:
                proc.wait(timeout=10)
            except Exception:
                pass
            return False

    def _read(self, name, conv=float):
        try:
            return conv(open(os.path.join(self.tmp, name)).read().strip())
        except (OSError, ValueError):
            return None

    def _same(self, a, b):
        try:
            return all(open(os.path.join(self.tmp, a + e), "rb").read() == open(os.path.join(self.tmp, b + e), "rb").read() for e in (".prune.in", ".prune.out"))
        except OSError:
            return False

    def finish(self, gpus=1, compare_with=None):
        """compare_with: prefix of another run's output files in the fileset's directory (bench.py --gpus N: the one-GPU run of plink2-hip made there before)"""
        if not self.tmp:
            return self.res
        try:
            res = self.res
            ref_wall = ref_rc = None
            if self.ref_proc is not None:
                deadline = self.ref_t0 + self.ref_timeout_s
                alive = self._join(self.ref_proc, deadline)
                ref_rc = self._read("ref.rc", int) if alive else -9
                end = self._read("ref.end")
                ref_wall = (end - self.ref_t0) if (end and ref_rc == 0) else None
                log = ""
                try:
                    log = open(os.path.join(self.tmp, "ref.log")).read()
                except OSError:
                    pass
                mt = re.search(r"\((\d+) compute thread", log)
                res["reference_plink2"] = {"wall_s": ref_wall, "rc": ref_rc, "threads_requested": self.cores, "compute_threads": int(mt.group(1)) if mt else None,
                                           "note": "started right after the fileset was written, timed from there to its own end stamp; it ran BESIDE this script's GPU legs (it uses a "
                                                   "dozen host threads, the legs one) and beside the variable-width conversion + run, so its wall is if anything pessimistic"}
            self.torch.cuda.synchronize()
            extra = ["--debug-alias-devices"] if (gpus > 1 and os.environ.get("LDP_BENCH_ALIAS_DEVICES")) else []
            hip = run_plink2_hip(self.tmp, "g", self.kb, self.cfg["r2"], "hip", gpus=gpus, extra=extra)
            if hip["phases"] and hip["phases"]["file_to_hbm_s"] > 0:
                hip["phases"]["file_to_hbm_gbs"] = self.file_bytes / hip["phases"]["file_to_hbm_s"] / 1e9
                hip["phases"]["note"] = ("plink2-hip --timing, the faster of its two runs; file_to_hbm covers pread() of the .pgen rows into the pinned ring, H2D and the count pass (they overlap); "
                                         "run = pair kernels + replay behind the load; the rest of the wall is process start-up, HIP context, list writing and exit")
            res["plink2_hip"] = hip
            res["gpus"] = gpus
            if compare_with:
                res["identical_to_one_gpu"] = bool(hip["rc"] == 0 and self._same("hip", compare_with))
            res["files_identical"] = bool(hip["rc"] == 0 and ref_rc == 0 and self._same("hip", "ref")) if self.ref_proc is not None else None
            res["speedup"] = (ref_wall / hip["wall_s"]) if (ref_wall and hip["wall_s"] and hip["rc"] == 0) else None
            if hip["phases"] and ref_wall:
                res["reference_candidate_pairs_per_s"] = hip["phases"]["candidate_pairs"] / ref_wall
            res["what"] = ("MEASURED end-to-end walls, process start to exit, same command line (--indep-pairwise %s %g) on the same fileset: %d variants (a chr22-sized share "
                           "of the metric's 10M-variant genome: 22 chromosomes at %d bp) x %d samples" % (self.kb, self.cfg["r2"], self.m, self.cfg["spacing"], self.cfg["samples"]))
            if self.var_proc is not None:
                # the same genotypes as the reference's default variable-width .pgen
                v = {}
                alive = self._join(self.var_proc, self.ref_t0 + self.ref_timeout_s + 120)
                var_rc = self._read("v.rc", int) if alive else -9
                stamp, end = self._read("v.stamp"), self._read("v.end")
                v_pgen = os.path.join(self.tmp, "v.pgen")
                if os.path.exists(v_pgen) and stamp:
                    v["pgen_bytes"] = os.path.getsize(v_pgen)
                    v["bytes_vs_fixed_width"] = v["pgen_bytes"] / float(self.file_bytes)
                    v["make_pgen_s"] = stamp - self.var_t0
                v["reference_rc"] = var_rc
                v["reference_plink2_wall_s"] = (end - stamp) if (stamp and end and var_rc == 0) else None
                if os.path.exists(v_pgen) and stamp:
                    vh = run_plink2_hip(self.tmp, "v", self.kb, self.cfg["r2"], "vhip")
                    v["plink2_hip"] = vh
                    if vh["phases"] and vh["phases"]["file_to_hbm_s"] > 0:
                        vh["phases"]["record_bytes_gbs"] = v["pgen_bytes"] / vh["phases"]["file_to_hbm_s"] / 1e9
                        vh["phases"]["rows_decoded_gbs"] = self.file_bytes / vh["phases"]["file_to_hbm_s"] / 1e9
                    v["files_identical_to_reference_on_the_same_file"] = bool(vh["rc"] == 0 and var_rc == 0 and self._same("vhip", "vref"))
                    v["files_identical_to_the_fixed_width_run"] = bool(vh["rc"] == 0 and hip["rc"] == 0 and self._same("vhip", "hip"))
                    if v["reference_plink2_wall_s"] and vh["wall_s"] and vh["rc"] == 0:
                        v["speedup"] = v["reference_plink2_wall_s"] / vh["wall_s"]
                v["what"] = ("the same fileset converted by the reference (`--make-pgen`, its default variable-width storage: LD-compressed, difflist and 1-bit records), both binaries "
                             "end to end on it; plink2-hip copies the records' bytes to the device and decodes them there (DESIGN 4.5b), so `pgen_bytes` is what crosses PCIe")
                res["variable_width"] = v
            return res
        finally:
            subprocess.call(["rm", "-rf", self.tmp])
            self.tmp = None


def pmc_traffic_in_run(bench_py, argv_workload, timeout_s=150):
    """HBM bytes of ONE step of the named workload measured NOW, on this box: two rocprofv3 passes over a one-step child run of bench.py
    (`--kernel-trace --pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`: separate passes, as MI355X_MICROARCH.md's HBM section prescribes), summed over
    the pair kernels' dispatches; FETCH_SIZE / WRITE_SIZE are in KiB and gfx950's FETCH_SIZE reports half of a wide streaming read (x 2).
    Returns (bytes, per-kernel dict, note) or (None, None, why)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="ldbench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    per_kernel = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(bench_py), "--steps", "1", "--warmup", "0",
                   "--no-legs", "--no-cpu-baseline"] + argv_workload
            cp = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
            if cp.returncode != 0:
                return None, None, "rocprofv3 --pmc %s failed: %s" % (ctr, cp.stdout[-200:])
            found = False
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    if r["Counter_Name"] == ctr and (("pair_mfma" in k) or ("pair_tiles_kernel" in k)):
                        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ldp::", "")
                        d = per_kernel.setdefault(short, {"FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0, "dispatches": 0})
                        d[ctr + "_KiB"] += float(r["Counter_Value"])
                        d["dispatches"] += 1 if ctr == "FETCH_SIZE" else 0
                        found = True
            if not found:
                return None, None, "no %s rows for the pair kernels in rocprofv3's output" % ctr
        total = sum(d["FETCH_SIZE_KiB"] * 1024.0 * 2.0 + d["WRITE_SIZE_KiB"] * 1024.0 for d in per_kernel.values())
        return total, per_kernel, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (two passes) around one step of the same workload in a "
                                   "child process; FETCH_SIZE x 1024 x 2 (gfx950 half-count of wide streaming reads, MI355X_MICROARCH.md HBM) + WRITE_SIZE x 1024, "
                                   "summed over the pair kernels' dispatches")
    except Exception as ex:  # pragma: no cover
        return None, None, "in-run PMC failed: %s" % str(ex)[:200]
    finally:
        subprocess.call(["rm", "-rf", tmp])
