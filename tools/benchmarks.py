#!/usr/bin/env python3
"""Benchmark runner for vpcli with the interface and CSV layout of the reference's scripts/benchmarks.py
(/root/reference/scripts/benchmarks.py:15-23,45-122): for every mesh in --folder, every type and every
power-of-two size it runs `vpcli <mesh> -n<size> -t<type> -m<niter> -p1 [-s]`, collects the
"[Label]: <ms> ms" timer lines, and writes <output>/<mesh>/<mesh>_<variant>.csv with one row per
iteration and snake_case label columns (e.g. tiled_vox, tiled_vox__memory, tiled_vox__processing)."""
import argparse
import collections
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TIMER = re.compile(r"\[(.*)\]: ([\d.]+) ms")


def snake(label: str) -> str:
    s = label.replace("::", "__")
    s = re.sub(r"(?<=[a-z0-9])([A-Z])", r"_\1", s)
    s = re.sub(r"([A-Z]+)([A-Z][a-z])", r"\1_\2", s)
    return re.sub(r"__+", "__", s.lower())


def parse_records(stdout: str):
    """Yield (variant, {column: ms}) per completed outer scope; inner lines print before their outer line."""
    pending = collections.OrderedDict()
    for line in stdout.splitlines():
        m = TIMER.search(line)
        if not m:
            continue
        label = re.sub(r"\s*\(.*?\)", "", m.group(1))           # drop "(mesh name)"
        col = snake(label)
        pending[col] = pending.get(col, 0.0) + float(m.group(2))
        if "__" not in col:                                      # the outer scope closes a record
            yield snake(label.split("::")[0]), dict(pending)
            pending.clear()


def main():
    ap = argparse.ArgumentParser(description="Benchmark runner")
    ap.add_argument("--niter", type=int, default=10)
    ap.add_argument("--folder", default=os.path.join(ROOT, "assets"))
    ap.add_argument("--maxsize", type=int, default=128)
    ap.add_argument("--minsize", type=int, default=32)
    ap.add_argument("--output", default="benchmarks")
    ap.add_argument("--no-sdf", action="store_true")
    ap.add_argument("--types", nargs="+", default=["3", "1", "2"])
    ap.add_argument("--exe", default=os.path.join(ROOT, "cuda_mesh_voxelization_amd", "vpcli"))
    ap.add_argument("--only", nargs="*", help="restrict to these mesh file names")
    a = ap.parse_args()

    sizes = []
    n = a.minsize
    while n <= a.maxsize:
        sizes.append(n)
        n *= 2
    os.makedirs(a.output, exist_ok=True)
    for fname in sorted(os.listdir(a.folder)):
        path = os.path.join(a.folder, fname)
        if not os.path.isfile(path) or not fname.lower().endswith(".obj") or (a.only and fname not in a.only):
            continue
        rows = collections.defaultdict(lambda: collections.defaultdict(list))     # variant -> size -> [record]
        for t in a.types:
            for size in sizes:
                cmd = [a.exe, path, "-n%d" % size, "-t%s" % t, "-m%d" % a.niter, "-p1"]
                if not a.no_sdf and size <= 512:
                    cmd.append("-s")
                print("Running:", " ".join(cmd))
                p = subprocess.run(cmd, capture_output=True, text=True)
                if p.returncode != 0:
                    sys.exit("command failed (%d):\n%s\n%s" % (p.returncode, p.stdout, p.stderr))
                for variant, rec in parse_records(p.stdout):
                    rows[variant][size].append(rec)
        stem = os.path.splitext(fname)[0]
        os.makedirs(os.path.join(a.output, stem), exist_ok=True)
        for variant, by_size in rows.items():
            cols = sorted({c for recs in by_size.values() for r in recs for c in r})
            with open(os.path.join(a.output, stem, "%s_%s.csv" % (stem, variant)), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["size"] + cols)
                for size in sorted(by_size):
                    for r in by_size[size]:
                        w.writerow([size] + [r.get(c, "") for c in cols])


if __name__ == "__main__":
    main()
