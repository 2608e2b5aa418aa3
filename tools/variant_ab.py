#!/usr/bin/env python3
"""Same-box A/B of library builds (csrc `make variant NAME=x EXTRA="-D..."` -> build/variants/libvof2d_x.so).

    python3 tools/variant_ab.py [--n 4096] [--steps 600] [--reps 2] [--strip] base pf2 ...

Every (variant, repetition) runs in its own process (one process loads one build of the library),
alternating A B A B so clock / thermal drift hits all alike.  Prints per variant: kernel averages of
the built-in profiler over steps 11-60 (before the tiny-value front) and 301-350 (inside it), and
ms/step of a `--steps`-step run in blocks of 100.  --strip: the interior strip of 8 of an 8192^2 grid
(1024 + 2 x 16 rows, dt 1e-6) instead of the full domain.  "base" = the product library."""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))


def lib_path(name):
    b = os.path.join(ROOT, "taichi-2d-vof_amd", "csrc", "build")
    return os.path.join(b, "libvof2d_hip.so") if name == "base" else os.path.join(b, "variants", "libvof2d_%s.so" % name)


def child(a):
    from vof2d import _abi
    from vof2d.engine import Engine, make_desc
    api = _abi.bind(ctypes.CDLL(lib_path(a.child), mode=ctypes.RTLD_GLOBAL), "vof_")
    kw = {}
    n = a.n
    if a.strip:
        n = 8192
        kw = dict(rows=(3 * 1024 + 1 - 16, 4 * 1024 + 16), own=(3 * 1024 + 1, 4 * 1024), dt=1e-6)
    if a.dt > 0:
        kw["dt"] = a.dt
    e = Engine(api, make_desc(api, n, n, a.dtype, "f32", device=0, **kw))
    for k, v in (kv.split("=") for kv in a.param):
        e.set_param(k, float(v))
    e.set_init_F(a.ic)
    out = {"variant": a.child, "blocks": [], "prof": {}}
    e.step(10)
    e.sync()
    out["prof"]["11-60"] = e.profile_steps(50)
    done = 60
    blocks = []
    while done < a.steps:
        if done == 300:
            out["prof"]["301-350"] = e.profile_steps(50)
            done += 50
            continue
        k = min(100, a.steps - done, (300 - done) if done < 300 else 100)
        e.sync()
        t0 = time.perf_counter()
        e.step(k)
        e.sync()
        blocks.append((done + k, 1e3 * (time.perf_counter() - t0) / k))
        done += k
    out["blocks"] = blocks
    import hashlib
    h = hashlib.sha256()
    for f in ("F", "u", "v", "p"):
        h.update((e.get(f) + 0.0).tobytes())
    out["state_sha256"] = h.hexdigest()[:16]       # all variants ran the same steps: equal digests = equal values
    out["istep"] = e.istep
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*", default=["base"])
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=650)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("-ic", type=int, default=1)
    ap.add_argument("--strip", action="store_true")
    ap.add_argument("--dt", type=float, default=0.0, help="time step (0: the library's default; 8192^2 needs 1e-6)")
    ap.add_argument("--param", action="append", default=[], help="knob=value set on every engine")
    ap.add_argument("--env", action="append", default=[], help="NAME=value in the environment of every run")
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.child:
        return child(a)
    res = {v: [] for v in a.variants}
    for rep in range(a.reps):
        for v in a.variants:
            if not os.path.exists(lib_path(v)):
                print("missing", lib_path(v))
                continue
            cmd = [sys.executable, os.path.abspath(__file__), "--child", v, "--n", str(a.n), "--steps", str(a.steps),
                   "--dtype", a.dtype, "-ic", str(a.ic), "--dt", str(a.dt)] + (["--strip"] if a.strip else []) + sum((["--param", p] for p in a.param), [])
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, **dict(kv.split("=", 1) for kv in a.env)))
            if r.returncode != 0:
                print(v, "FAILED", r.stderr[-800:])
                continue
            res[v].append(json.loads(r.stdout.strip().splitlines()[-1]))
    print("workload: %s %s ic%d%s %s" % ("strip 1056x8192 of 8192^2" if a.strip else "%d^2" % a.n, a.dtype, a.ic,
                                        "", " ".join(a.param + a.env)))
    for v, runs in res.items():
        for i, r in enumerate(runs):
            ks = []
            for win in ("11-60", "301-350"):
                p = r["prof"].get(win, {})
                ks.append(win + ": " + " ".join("%s %.1f" % (k.replace("k_", ""), us) for k, (us, cnt) in sorted(p.items())))
            bl = " ".join("%.3f" % ms for _, ms in r["blocks"])
            mean = sum(ms for _, ms in r["blocks"]) / max(1, len(r["blocks"]))
            print("%-10s run %d | %s | %s | ms/step blocks: %s | mean %.4f | state %s @%d" % (
                v, i, ks[0], ks[1], bl, mean, r["state_sha256"], r["istep"]))


if __name__ == "__main__":
    main()
