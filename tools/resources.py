"""Per-kernel register / spill / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage.
    python tools/resources.py [remarks.txt] [name filter ...]
Without a file the product library's sources are compiled (device code only, nothing is written)."""
import os
import re
import subprocess
import sys

KEYS = ("VGPRs", "AGPRs", "TotalSGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]",
        "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")
CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "taichi-2d-vof_amd", "csrc")


def remarks(extra=()):
    srcs = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]   # one translation unit (vof2d_api.hip) that includes kernels/ and runtime/
    out = ""
    for s in srcs:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
               "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", s, *extra]
        out += subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    return out


def parse(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        body = line.split("remark: ")[-1].strip()
        for k in KEYS:
            if body.startswith(k + ":"):
                cur[k] = int(re.search(r": (\d+)", body).group(1))
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True,
                           text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["name"] = re.sub(r"\(.*", "", n).replace("void vof::", "")
    return rows


def main():
    args = sys.argv[1:]
    text = open(args.pop(0)).read() if args and os.path.exists(args[0]) else remarks()
    print("%-64s %5s %5s %6s %6s %7s %4s %6s" % ("kernel", "VGPR", "SGPR", "sSpill", "vSpill", "scratch", "occ", "LDS"))
    for r in parse(text):
        if args and not any(a in r["name"] for a in args):
            continue
        print("%-64s %5s %5s %6s %6s %7s %4s %6s" % (r["name"][:64], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("SGPRs Spill"),
                                                   r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
                                                   r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()
