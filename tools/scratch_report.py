#!/usr/bin/env python
"""Static spill report of a HIP source: per kernel, the scratch (spill) instructions and how many of them sit inside a loop that also
holds MFMAs (the per-tile / k loops of the conv kernels).  CPU only — compiles to device assembly with hipcc.

    python tools/scratch_report.py imagen-pytorch_amd/csrc/igemm.hip [-DIGEMM_EPI_REMAT ...]

A spill reload in such a loop is a scratch_load, a VMEM operation that retires through the in-order vmcnt counter: the wait in front of
its first use also waits for every older global load (weight ring, epilogue operands) — DESIGN.md 9.1."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(asm_path):
    txt = open(asm_path).read().split("\n")
    starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
    rows = []
    for start in starts:
        name = txt[start].split(":")[0]
        end = next((i for i in range(start, len(txt)) if txt[i].strip().startswith(".Lfunc_end")), len(txt))
        body = txt[start:end]
        labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        scr = [i for i, l in enumerate(body) if "scratch_" in l]
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        in_loop = sum(1 for i in scr if any(a <= i <= b and any(a <= j <= b for j in mf) for a, b in loops))
        drains = sum(1 for i in scr if "scratch_load" in body[i] and any("vmcnt(0)" in body[j] for j in range(i + 1, min(i + 4, len(body)))))
        rows.append((name, len(body), len(scr), in_loop, drains))
    return rows


def device_code_hash(asm_path):
    """sha256 of the device assembly without what changes with the source TEXT but not with the code: the compilation-unit id symbol,
    .file / .ident lines, comments.  Equal hash = the same instructions, registers and metadata as the compared build."""
    import hashlib
    text = re.sub(r"__hip_cuid_[0-9a-f]+", "__hip_cuid_X", open(asm_path).read())
    lines = [l for l in text.split("\n") if not l.strip().startswith((".file", ".ident", ";")) and ".loc" not in l]
    return hashlib.sha256("\n".join(lines).encode()).hexdigest()


def main():
    src = sys.argv[1]
    flags = sys.argv[2:]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "imagen-pytorch_amd", "csrc"), "--cuda-device-only", "-S", src, "-o", out, *flags]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        rows = report(out)
        digest = device_code_hash(out)
    print(f"{'kernel':100s} {'lines':>6s} {'scratch':>8s} {'in MFMA loops':>14s} {'reload+vmcnt(0)':>16s}")
    for name, n, s, l, d in rows:
        if s:
            print(f"{name[:100]:100s} {n:6d} {s:8d} {l:14d} {d:16d}")
    print(f"{len(rows)} kernels, {sum(1 for r in rows if r[2])} with scratch, {sum(r[3] for r in rows)} scratch instructions inside MFMA loops")
    print(f"device code sha256 {digest}  (profiles/r02_device_code_hashes.txt holds the versions that met hardware)")


if __name__ == "__main__":
    main()
