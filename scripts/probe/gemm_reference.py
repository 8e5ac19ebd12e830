#!/usr/bin/env python3
"""gemm_reference.py - a KNOWN-GOOD GEMM measured on the same box, in the same call, as the coarse kernel.

VERDICT r4 item 1 / cdna_hip_programming.md section 5.4 rule 10: "never infer a platform ceiling from your own failed
attempts". The shipped coarse pass (the scoring inside MilvusClient.search on the FLAT/IP index,
services/milvus_service.py:280-285) runs at 0.40 of the nominal 2.5 PFLOP/s fp16 MFMA peak; this probe puts next to it, on
ONE box and on the same random data kind:

  (a) the vendor fp16 GEMM (hipBLASLt / rocBLAS through torch.matmul) at the coarse pass's own shape
      (10 240 x 37 120 x 768, fp16 in, fp16 out: 760 MB of output the coarse kernel never writes), at an output-light shape
      of the same K (8 192 x 8 192 x 768), and at the guide's 8 192^3 (K long enough to amortise prologue and epilogue);
  (b) the bare MFMA stream (scripts/probe/bare_mfma: operands in registers, nothing else);
  (c) the shipped coarse kernel with and without its select (ablation build csrc/abc, ICD_FLAT_VAR);
  (d) for each of them, over >= 2 s of back-to-back launches: the clock and power the box reports (sysfs pp_dpm_sclk /
      hwmon power1_average, amd-smi as a cross-check) and - where the kernel is ours - the in-kernel clock
      (s_memtime / s_memrealtime).

Everything is printed as TFLOP/s and as a fraction of 2 500; the last lines give the ratios bench.py's
`roofline.frac_of_reference_gemm` is defined by. Run on the GPU box: `python3 scripts/probe/gemm_reference.py`
(scripts/gpu_gemm_reference.sh wraps it and adds the rocprofv3 kernel trace that names the vendor kernel).
"""
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PEAK = 2500.0


class Sampler:
    """clock / power of device 0 every 50 ms from sysfs (no HIP call, no subprocess in the loop)"""

    def __init__(self):
        self.sclk = None
        self.power = None
        for card in sorted(glob.glob('/sys/class/drm/card*/device')):
            if os.path.exists(card + '/pp_dpm_sclk'):
                self.sclk = card + '/pp_dpm_sclk'
                hw = sorted(glob.glob(card + '/hwmon/hwmon*/power1_average')) + sorted(glob.glob(card + '/hwmon/hwmon*/power1_input'))
                self.power = hw[0] if hw else None
                break
        self.samples = []
        self._stop = threading.Event()
        self._t = None

    def _read(self):
        mhz = watts = None
        try:
            for line in open(self.sclk):
                if '*' in line:
                    mhz = float(re.search(r'(\d+)\s*[Mm][Hh]z', line).group(1))
        except Exception:
            pass
        try:
            watts = float(open(self.power).read()) / 1e6
        except Exception:
            pass
        return mhz, watts

    def start(self):
        self.samples = []
        self._stop.clear()

        def run():
            while not self._stop.is_set():
                self.samples.append(self._read())
                time.sleep(0.05)
        self._t = threading.Thread(target=run, daemon=True)
        self._t.start()

    def stop(self):
        self._stop.set()
        self._t.join()
        # the samples taken under load: a binary arm also generates data and checks a sample on the host, so keep the
        # samples whose power is within 15 % of the arm's highest (all of them when no power sensor is readable)
        pmax = max((x[1] for x in self.samples if x[1] is not None), default=None)
        s = [x for x in self.samples if pmax is None or (x[1] is not None and x[1] >= 0.85 * pmax)]
        mhz = sorted(x[0] for x in s if x[0] is not None)
        w = sorted(x[1] for x in s if x[1] is not None)
        med = lambda v: v[len(v) // 2] if v else None
        return {"sclk_mhz_median": med(mhz), "sclk_mhz_min": mhz[0] if mhz else None, "power_w_median": med(w),
                "power_w_max": w[-1] if w else None, "busy_samples": len(s), "samples": len(self.samples)}


def amd_smi_once():
    try:
        out = subprocess.run(['amd-smi', 'metric', '-g', '0', '--clock', '--power', '--json'], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out)
        j = j[0] if isinstance(j, list) else j
        clk = j.get('clock', {}).get('gfx_0', {}).get('clk', {})
        pw = j.get('power', {}).get('socket_power', {})
        return {"gfx_0_clk": clk.get('value') if isinstance(clk, dict) else clk, "socket_power": pw.get('value') if isinstance(pw, dict) else pw}
    except Exception as e:   # the tool may be refused to an ordinary user: the sysfs samples stand alone then
        return {"error": str(e)[:120]}


def fmt(d):
    return ' '.join(f'{k}={v}' for k, v in d.items())


def torch_arms(sampler):
    import torch
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1234)
    out = {}
    print(f'# torch {torch.__version__}, preferred BLAS backend: {torch.backends.cuda.preferred_blas_library()}')

    def unit_rows(m, k, scale):   # the coarse pass's data kind: unit rows x one power of two
        x = torch.randn(m, k, device=dev, generator=g, dtype=torch.float32)
        x = x / x.norm(dim=1, keepdim=True) * scale
        return x.half()
    for name, (m, n, k) in {"coarse_shape_10240x37120x768": (10240, 37120, 768), "output_light_8192x8192x768": (8192, 8192, 768),
                             "long_k_8192x8192x8192": (8192, 8192, 8192)}.items():
        a = unit_rows(m, k, 16.0)
        b = unit_rows(n, k, 16.0)
        c = torch.empty(m, n, device=dev, dtype=torch.float16)
        flop = 2.0 * m * n * k
        for _ in range(5):
            torch.matmul(a, b.t(), out=c)
        torch.cuda.synchronize()
        # >= 2 s of back-to-back launches under the sampler, then the timed window inside the same load
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        per = None
        sampler.start()
        t0 = time.time()
        smi = None
        while time.time() - t0 < 2.5:
            e0.record()
            for _ in range(20):
                torch.matmul(a, b.t(), out=c)
            e1.record()
            torch.cuda.synchronize()
            per = e0.elapsed_time(e1) / 20
            if smi is None and time.time() - t0 > 1.2:
                smi = amd_smi_once()
        clk = sampler.stop()
        tf = flop / (per * 1e-3) / 1e12
        out[name] = {"ms": per, "tflops": tf, "frac": tf / PEAK, "clock": clk, "amd_smi": smi}
        print(f'torch.matmul fp16 {name}: {per:.4f} ms, {tf:.0f} TFLOP/s ({tf / PEAK:.3f} of 2500) | {fmt(clk)} | amd-smi {smi}')
        del a, b, c
    return out


def run_binary(sampler, label, cmd, env=None, pattern=None):
    e = dict(os.environ)
    e.update(env or {})
    sampler.start()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e)
    time.sleep(1.5)
    smi = amd_smi_once()
    try:
        txt, _ = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        p.kill()
        txt, _ = p.communicate()
        txt += '\n(killed after 240 s)'
    clk = sampler.stop()
    lines = [l for l in txt.splitlines() if (re.search(pattern, l) if pattern else True) and 'amdgpu.ids' not in l]
    print(f'### {label} | {fmt(clk)} | amd-smi {smi}')
    for l in lines:
        print('   ' + l.strip())
    return {"clock": clk, "amd_smi": smi, "lines": lines}


def trace_only():
    """the three vendor GEMMs alone, a few launches each: run under `rocprofv3 --kernel-trace --stats` to NAME the vendor kernels"""
    import torch
    dev = torch.device('cuda:0')
    for m, n, k in ((10240, 37120, 768), (8192, 8192, 768), (8192, 8192, 8192)):
        a = (torch.randn(m, k, device=dev) / 4).half()
        b = (torch.randn(n, k, device=dev) / 4).half()
        c = torch.empty(m, n, device=dev, dtype=torch.float16)
        for _ in range(12):
            torch.matmul(a, b.t(), out=c)
        torch.cuda.synchronize()
        del a, b, c


def main():
    if '--trace-only' in sys.argv:
        return trace_only()
    sampler = Sampler()
    print(f'# sysfs: {sampler.sclk} {sampler.power}; idle: {sampler._read()}; amd-smi idle: {amd_smi_once()}')
    res = {"torch": torch_arms(sampler)}
    import torch
    torch.cuda.synchronize()
    # (b) bare MFMA stream, 2.5 s of settling per shape
    bare = os.path.join(ROOT, 'scripts/probe/bare_mfma')
    if os.path.exists(bare):
        for shape in (16, 32):
            res[f'bare_{shape}'] = run_binary(sampler, f'bare MFMA stream, shape {shape}', [bare, '90', '2.5', str(shape)])
    # (c) the shipped coarse kernel with and without its select, each with an in-kernel-clock twin (bit 33554432)
    ab = os.path.join(ROOT, 'rag_project_icd10_amd/csrc/abc')
    st = os.path.join(ab, 'icd_selftest')
    oracle = os.path.join(ROOT, 'oracle/libicd_oracle.so')
    if os.path.exists(st):
        for label, var in (("shipped coarse kernel (CF_CACHED_VAR)", 6326427), ("... without its select (+8192)", 6334619),
                           ("shipped coarse kernel, in-kernel clock stamps (+33554432)", 39880859),
                           ("... without its select, in-kernel clock stamps", 39889051)):
            res[f'coarse_{var}'] = run_binary(sampler, f'{label} ICD_FLAT_VAR={var}',
                                              [st, '--oracle', oracle, '--skip-cases', '--bench', '--auto-only', '--iters', '3500'],
                                              env={"ICD_FLAT_VAR": str(var)}, pattern=r'mode=auto|in-kernel|parity|FAIL')
    # summary: coarse-kernel times against the vendor GEMM at the same shape (flop of the bench shape / time)
    def coarse_ms(key):
        for l in res.get(key, {}).get("lines", []):
            m = re.search(r'coarse=([0-9.]+)', l)
            if m:
                return float(m.group(1))
        return None
    flop = 2.0 * 10000 * 37000 * 768
    ref = res["torch"]["coarse_shape_10240x37120x768"]
    light = res["torch"]["output_light_8192x8192x768"]
    print('## summary (fractions of 2 500 TFLOP/s)')
    print(f'vendor GEMM at the coarse shape: {ref["frac"]:.3f}; output-light same K: {light["frac"]:.3f}; 8192^3: {res["torch"]["long_k_8192x8192x8192"]["frac"]:.3f}')
    for key, name in (("coarse_6326427", "shipped coarse kernel"), ("coarse_6334619", "its loop without the select")):
        ms = coarse_ms(key)
        if ms:
            tf = flop / (ms * 1e-3) / 1e12
            print(f'{name}: {ms:.4f} ms = {tf:.0f} TFLOP/s = {tf / PEAK:.3f}; x {tf / ref["tflops"]:.3f} of the vendor GEMM at the coarse shape, '
                  f'x {tf / light["tflops"]:.3f} of the output-light one')
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out/gemm_reference.json'), 'w'), indent=1, default=str)


if __name__ == '__main__':
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    main()
