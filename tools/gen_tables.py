#!/usr/bin/env python3
"""Extract the numeric DATA tables the hot path consumes into little-endian float32 blobs.

Run in the build container (needs /root/reference); the blobs under pathtracer_amd/data/ are
committed because the reference tree does not travel to the GPU box.

  hw_datasets.bin      11*2*10*6*9 f32   Hosek-Wilkie sky-dome coefficients   hosek_wilkie.rs:387-2214
  hw_datasets_rad.bin  11*2*10*6   f32   Hosek-Wilkie mean radiances          hosek_wilkie.rs:2216-2503
  hw_solar.bin         11*10*45*4  f32   solar-disc piecewise cubics          hosek_wilkie.rs:2505-7698
  hw_limb.bin          11*6        f32   limb-darkening quintics              hosek_wilkie.rs:7700-7712
  srgb2refl.bin        471*3       f32   sRGB -> reflectance basis (360..830 nm)   src/data/RGB-Components-CIE-1931-1nm.csv via build.rs:36-41
  d65.bin              531         f32   CIE D65 (300..830 nm), column 2      src/data/Illuminantd65.csv via build.rs:43-50

These are published measurement/fit data (Hosek & Wilkie 2012/2013 data set, BSD-3 — notice kept in
pathtracer_amd/data/NOTICE; CIE D65; Burns' sRGB reflectance basis), not program text.  Only the
numbers are taken; every literal is rounded decimal -> binary32 exactly (round-half-even on the
exact rational), which is what rustc and the csv crate's f32 parser do.
"""
import re
import struct
import sys
from fractions import Fraction
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent.parent / "pathtracer_amd" / "data"

NUM = re.compile(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?")


def to_f32_exact(lit: str) -> np.float32:
    """Correctly rounded decimal literal -> binary32."""
    exact = Fraction(lit)
    guess = np.float32(float(lit))
    if not np.isfinite(guess):
        raise ValueError(lit)
    best, best_err = None, None
    for cand in (np.nextafter(guess, np.float32(-np.inf)), guess, np.nextafter(guess, np.float32(np.inf))):
        err = abs(Fraction(float(cand)) - exact)
        even = (int(np.float32(cand).view(np.uint32)) & 1) == 0
        key = (err, 0 if even else 1)
        if best is None or key < best_err:
            best, best_err = cand, key
    return np.float32(best)


def rust_table(src: str, name: str, count: int) -> np.ndarray:
    m = re.search(r"const\s+" + name + r"\s*:[^=]*=\s*\[", src)
    if not m:
        raise SystemExit(f"table {name} not found")
    start = m.end()
    end = src.index("];", start)
    lits = NUM.findall(src[start:end])
    if len(lits) != count:
        raise SystemExit(f"{name}: expected {count} numbers, found {len(lits)}")
    return np.array([to_f32_exact(x) for x in lits], dtype=np.float32)


def main() -> None:
    OUT.mkdir(parents=True, exist_ok=True)
    src = (REF / "src" / "hosek_wilkie.rs").read_text()
    # strip comments so a number inside one can never be picked up
    src = re.sub(r"//[^\n]*", "", src)
    tabs = {
        "hw_datasets.bin": rust_table(src, "DATASETS", 11 * 2 * 10 * 6 * 9),
        "hw_datasets_rad.bin": rust_table(src, "DATASETS_RAD", 11 * 2 * 10 * 6),
        "hw_solar.bin": rust_table(src, "SOLAR_DATASETS", 11 * 10 * 45 * 4),
        "hw_limb.bin": rust_table(src, "LIMB_DARKENING_DATASETS", 11 * 6),
    }
    rows = [l for l in (REF / "src/data/RGB-Components-CIE-1931-1nm.csv").read_text().splitlines() if l.strip()]
    refl = [to_f32_exact(x.strip()) for r in rows for x in r.split(",")]
    assert len(refl) == 471 * 3, len(refl)
    tabs["srgb2refl.bin"] = np.array(refl, dtype=np.float32)
    rows = [l for l in (REF / "src/data/Illuminantd65.csv").read_text().splitlines() if l.strip()]
    assert int(rows[0].split(",")[0]) == 300 and len(rows) == 531, (rows[0], len(rows))
    tabs["d65.bin"] = np.array([to_f32_exact(r.split(",")[1].strip()) for r in rows], dtype=np.float32)
    for fn, arr in tabs.items():
        (OUT / fn).write_bytes(arr.astype("<f4").tobytes())
        print(f"{fn}: {arr.size} f32, first={arr[0]!r} last={arr[-1]!r}")


if __name__ == "__main__":
    main()
