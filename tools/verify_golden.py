#!/usr/bin/env python3
"""Re-pin the oracle's fixtures without touching them: run every capture script into a scratch directory (they import
the reference from /root/reference, so this runs in the build container only) and compare array by array with the
tracked tests/golden/*.npz.

    python tools/verify_golden.py [--keep DIR]

Expected differences: `gv10_pretrain:log` (the reference's log lines carry wall-clock times).  Anything else is listed
and makes the exit code 1.  `git status` is clean afterwards: nothing under tests/golden is written."""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
SCRIPTS = ["capture_golden.py", "capture_golden_oneshot.py", "capture_golden_baggan.py", "capture_golden_kmeans.py",
           "capture_golden_native.py", "capture_golden_heads.py"]
EXPECTED = {("gv10_pretrain", "log")}


def compare(new_dir):
    bad, seen = [], 0
    for name in sorted(os.listdir(GOLDEN)):
        if not name.endswith(".npz"):
            continue
        stem = name[:-4]
        path = os.path.join(new_dir, name)
        if not os.path.exists(path):
            bad.append(f"{stem}: not regenerated")
            continue
        old, new = np.load(os.path.join(GOLDEN, name), allow_pickle=False), np.load(path, allow_pickle=False)
        for key in sorted(set(old.files) | set(new.files)):
            seen += 1
            if key not in old.files or key not in new.files:
                bad.append(f"{stem}:{key}: only in the {'tracked' if key in old.files else 'regenerated'} file")
                continue
            a, b = old[key], new[key]
            same = a.shape == b.shape and a.dtype == b.dtype and (np.array_equal(a, b, equal_nan=True) if a.dtype.kind in "fc"
                                                                    else np.array_equal(a, b))
            if same:
                continue
            if (stem, key) in EXPECTED:
                print(f"  (expected) {stem}:{key} differs")
                continue
            if a.shape == b.shape and a.dtype.kind == "f":
                d = np.abs(a.astype(np.float64) - b.astype(np.float64))
                bad.append(f"{stem}:{key}: max |diff| {d.max():.3g} (max |value| {np.abs(a).max():.3g})")
            else:
                bad.append(f"{stem}:{key}: {a.shape} {a.dtype} vs {b.shape} {b.dtype}")
    return bad, seen


def main():
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--keep", default=None, help="capture into this directory and leave it behind")
    args = ap.parse_args()
    tmp = args.keep or tempfile.mkdtemp(prefix="golden_")
    os.makedirs(tmp, exist_ok=True)
    for sc in SCRIPTS:
        print(f"== {sc}", flush=True)
        try:
            r = subprocess.run([sys.executable, os.path.join(HERE, sc), "--out", tmp], capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            raise SystemExit(f"{sc} did not finish in 300 s (each capture takes seconds: something it imports is waiting)")
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-4000:])
            raise SystemExit(f"{sc} failed with code {r.returncode}")
    bad, seen = compare(tmp)
    if not args.keep:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    print(f"{seen} arrays compared, {len(bad)} unexpected difference(s)")
    for line in bad:
        print("  " + line)
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
