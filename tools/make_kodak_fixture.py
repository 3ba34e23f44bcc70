#!/usr/bin/env python3
"""tests/golden/kodak_crops.npz: sixteen 256 x 256 crops of the reference's own Kodak images (DATA of the reference tree:
/root/reference/task-oriented-PTQ/datasets/kodak24/kodim*.png) as uint8 -- calibration inputs with natural-image statistics for the
parity tests (missing 3 / next 3).  The reference calibrates on random 256 x 256 crops turned into [0, 1] floats
(datasets/dataset.py:8-12: RandomCrop(patchsize) + ToTensor; main2.py:53 --patchsize 256); the tests do the same division by 255.

Runs in the build container only (the reference tree does not exist on the GPU box); the output is data: pixel values, the source
file names and the crop offsets.

usage: python tools/make_kodak_fixture.py [--out tests/golden/kodak_crops.npz]"""
import argparse
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/task-oriented-PTQ/datasets/kodak24"
N, P = 16, 256


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "kodak_crops.npz"))
    a = ap.parse_args()
    rs = np.random.RandomState(1005)
    files = sorted(f for f in os.listdir(SRC) if f.endswith(".png"))
    pick = [files[i] for i in sorted(rs.choice(len(files), N, replace=False))]
    crops, offs = [], []
    for f in pick:
        im = np.asarray(Image.open(os.path.join(SRC, f)).convert("RGB"))
        h, w, _ = im.shape
        y, x = int(rs.randint(0, h - P + 1)), int(rs.randint(0, w - P + 1))
        crops.append(im[y:y + P, x:x + P].copy())
        offs.append((y, x))
    crops = np.stack(crops).astype(np.uint8)
    np.savez_compressed(a.out, crops=crops, files=np.array(pick), offsets=np.array(offs, dtype=np.int32))
    print(f"{a.out}: {crops.shape} uint8, {os.path.getsize(a.out) / 1e6:.2f} MB; mean {crops.mean():.1f}, "
          f"per-crop std {[round(float(c.std()), 1) for c in crops]}")


if __name__ == "__main__":
    main()
